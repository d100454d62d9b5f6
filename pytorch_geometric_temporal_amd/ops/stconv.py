"""STConv: TemporalConv and the node-wise batch norm.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""


import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import KERNEL_TIMER, _timed, gemm, gemm_tn_acc


# --------------------------------------------------------------------------------------------- ST-Conv block: dense halves

class TemporalConvFunction(torch.autograd.Function):
    """TemporalConv.forward (stgcn.py:27-44): H = relu(conv_1(X) * sigmoid(conv_2(X)) + conv_3(X)), three Conv2d(Cin -> Cout,
    (1, k)) over time, on X [B, T, N, Cin] -> [B, T - k + 1, N, Cout] without the reference's permutes: ONE launch forward
    (pgt_tconv_glu_f32: the three convolutions as one row-shifted K-segmented product on the matrix cores, the gate on the
    accumulators).  Backward: the gate's adjoint as one streaming pass (pgt_tconv_glu_bwd_f32) that lays dP | dQ | dR out as
    the operand of ONE weight-gradient product (pgt_gemm_tn_acc_f32, X's taps as K segments) and ONE input-gradient
    product (pgt_gemm_f32, dZ's taps as K segments against the taps in reverse order)."""

    @staticmethod
    def forward(ctx, X, W1, W2, W3, b1, b2, b3):
        lib = _lib.get_lib()
        for t, n in ((X, "X"), (W1, "conv_1.weight"), (W2, "conv_2.weight"), (W3, "conv_3.weight")):
            check_tensor(lib, t, n)
        if X.dim() != 4:
            raise ValueError(f"TemporalConv: X must be [batch, time, nodes, channels], got {tuple(X.shape)}")
        B, T, N, Cin = X.shape
        Cout, k = W1.size(0), W1.size(3)
        if W1.shape != (Cout, Cin, 1, k) or W2.shape != W1.shape or W3.shape != W1.shape:
            raise ValueError("TemporalConv: the three convolutions must be Conv2d(in, out, (1, k)) of one shape")
        if T < k:
            raise RuntimeError(f"TemporalConv: kernel size {k} can't be greater than the {T} time steps")
        Xc = X.contiguous()
        dev = X.device
        # Wp[dt * Cin + ci, g * Cout + c] = conv_{g+1}.weight[c, ci, 0, dt]
        Wp = torch.stack((W1, W2, W3), 0)[:, :, :, 0, :].permute(3, 2, 0, 1).reshape(k * Cin, 3 * Cout).contiguous()
        has_bias = b1 is not None
        bias3 = torch.cat((b1, b2, b3)).contiguous() if has_bias else None
        Tp = T - k + 1
        H = torch.empty(B, Tp, N, Cout, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad)
        P = torch.empty_like(H) if need else None
        S = torch.empty_like(H) if need else None
        work = 4.0 * (B * T * N * Cin + (3 if need else 1) * H.numel() + Wp.numel()) if KERNEL_TIMER else 0
        _timed("tconv", work, lambda: lib.call(
            "pgt_tconv_glu_f32", ptr(Xc), Cin, B, T, N, Cin, Cout, k, ptr(Wp), ptr(bias3), ptr(H), ptr(P), ptr(S),
            stream_of(lib, H)), tag=(B * Tp * N, Cin, Cout, k))
        if need:
            ctx.save_for_backward(Xc, Wp, H, P, S)
        ctx.dims = (B, T, N, Cin, Cout, k)
        ctx.has_bias = has_bias
        return H

    @staticmethod
    def backward(ctx, dH):
        lib = _lib.get_lib()
        Xc, Wp, H, P, S = ctx.saved_tensors
        B, T, N, Cin, Cout, k = ctx.dims
        dev = dH.device
        dHc = dH.contiguous()
        pad, M = (k - 1) * N, B * T * N
        C3 = 3 * Cout
        dZ = torch.empty(pad + M, C3, dtype=F32, device=dev)
        _timed("tconv_bwd", 4.0 * (4 * H.numel() + dZ.numel()) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_tconv_glu_bwd_f32", ptr(dHc), ptr(H), ptr(P), ptr(S), B, T, N, Cout, k, ptr(dZ), stream_of(lib, dZ)))
        # weight / bias gradients: dWp[dt Cin + ci, :] = sum_m X[m + dt N, ci] dZ[m, :] over the rows that have outputs
        Mw = (B * T - (k - 1)) * N
        dWp = torch.zeros(k * Cin, C3, dtype=F32, device=dev)
        db3 = torch.zeros(C3, dtype=F32, device=dev) if ctx.has_bias else None
        G = dZ[pad:]
        gemm_tn_acc(Xc.view(M, Cin), Cin, N * Cin, k, Cin, G, C3, dWp, C3, db3, Mw, C3)
        dW = dWp.view(k, Cin, 3, Cout).permute(2, 3, 1, 0).unsqueeze(3)                    # [3, Cout, Cin, 1, k]
        dX = None
        if ctx.needs_input_grad[0]:
            # dX[m] = sum_j dZpad[m + j N] Wp[(k - 1 - j) Cin : (k - j) Cin, :]^T   (dZpad = dZ behind (k - 1) N zero rows)
            Wb = Wp.view(k, Cin, C3).flip(0).permute(0, 2, 1).reshape(k * C3, Cin).contiguous()
            dX = torch.empty(B, T, N, Cin, dtype=F32, device=dev)
            gemm(dZ, C3, N * C3, k, C3, Wb, Cin, 1, dX, Cin, 0, Cin, None, M, Cin)
        db = (db3[:Cout], db3[Cout:2 * Cout], db3[2 * Cout:]) if ctx.has_bias else (None, None, None)
        return (dX, dW[0].contiguous(), dW[1].contiguous(), dW[2].contiguous()) + db


def temporal_conv(X, conv_1, conv_2, conv_3):
    """The gate of a TemporalConv module from its three nn.Conv2d parameter holders."""
    return TemporalConvFunction.apply(X, conv_1.weight, conv_2.weight, conv_3.weight, conv_1.bias, conv_2.bias, conv_3.bias)


class BatchNormNodesFunction(torch.autograd.Function):
    """BatchNorm2d(num_nodes) of STConv (stgcn.py:129, :156-159) on [B, T', N, C] in place of permute -> BatchNorm2d ->
    permute: per-node statistics over (batch, time, channel), running statistics updated by the same launch
    (pgt_batchnorm_nodes_f32 / pgt_batchnorm_nodes_bwd_f32: one workgroup per node, deterministic)."""

    @staticmethod
    def forward(ctx, X, gamma, beta, running_mean, running_var, momentum, eps, training):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        B, Tp, N, C = X.shape
        Xc = X.contiguous()
        dev = X.device
        Y = torch.empty_like(Xc)
        stats = torch.empty(N, 2, dtype=F32, device=dev)
        use_batch = bool(training) or running_mean is None
        work = 8.0 * Xc.numel() if KERNEL_TIMER else 0
        _timed("batchnorm", work, lambda: lib.call(
            "pgt_batchnorm_nodes_f32", ptr(Xc), B * Tp, N, C, ptr(gamma), ptr(beta),
            ptr(running_mean) if training else ptr(running_mean if not use_batch else None),
            ptr(running_var) if training else ptr(running_var if not use_batch else None),
            float(momentum), float(eps), int(use_batch), ptr(Y), ptr(stats), stream_of(lib, Y)))
        ctx.save_for_backward(Xc, stats, gamma)
        ctx.use_batch = use_batch
        ctx.mark_non_differentiable(*(t for t in (running_mean, running_var) if t is not None))
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        Xc, stats, gamma = ctx.saved_tensors
        B, Tp, N, C = Xc.shape
        dev = dY.device
        dYc = dY.contiguous()
        dX = torch.empty_like(Xc) if ctx.needs_input_grad[0] else None
        dg = torch.empty(N, dtype=F32, device=dev) if gamma is not None and ctx.needs_input_grad[1] else None
        dbt = torch.empty(N, dtype=F32, device=dev) if ctx.needs_input_grad[2] else None
        _timed("batchnorm_bwd", 12.0 * Xc.numel() if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_batchnorm_nodes_bwd_f32", ptr(dYc), ptr(Xc), ptr(stats), ptr(gamma), B * Tp, N, C, int(ctx.use_batch),
            ptr(dX), ptr(dg), ptr(dbt), stream_of(lib, dYc)))
        return dX, dg, dbt, None, None, None, None, None


def batch_norm_nodes(X, bn, training):
    """X [B, T', N, C] through the parameters / buffers of an nn.BatchNorm2d(num_nodes) holder, torch's bookkeeping
    (num_batches_tracked, momentum = None -> cumulative average) on the host."""
    if X.size(2) != bn.num_features:
        raise ValueError(f"STConv: the input has {X.size(2)} nodes, the batch norm was built for {bn.num_features}")
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    if training and X.size(0) * X.size(1) * X.size(3) <= 1:
        raise ValueError("Expected more than 1 value per channel when training")
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return BatchNormNodesFunction.apply(X, bn.weight, bn.bias, rm, rv, momentum, bn.eps, training)
