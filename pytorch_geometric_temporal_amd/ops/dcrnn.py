"""DConv / DCRNN / BatchedDCRNN: diffusion stacks, cell Functions, the whole-sequence Functions (per step, seq_small, seq64).
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""
import ctypes

import os

import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import (FOLD_STATE_GRADIENT, FUSE_GATE_EPILOGUES, KERNEL_TIMER, ONE_FEATURE_GRADIENT, ONE_FEATURE_GRADIENT_MIN_ROWS, RowMap, SKIP_INPUT_COLUMNS_WHEN_UNUSED, SPLIT_FEATURE_GRADIENT, _gru_h, _gru_h_bwd, _gru_zr, _gru_zr_bwd, _timed, add2d, axpby2d, copy2d, gemm, gemm_gru_h, gemm_gru_zr, gemm_tn_acc, spmm)
# Weight-gradient GEMMs of step t on a side stream while the main stream runs the BPTT chain of step t-1 (same
# arithmetic, fp32 atomics into dW either way).  Measured on MI355X at METR-LA shape, B = 1024: 23.81 ms per step with
# the overlap vs 23.79 ms with one whole-sequence weight-gradient GEMM at the end -> off by default.
OVERLAP_WEIGHT_GRADIENTS = False
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


# --------------------------------------------------------------------------------------------- diffusion stack

def _stack_fwd(g, TS, t, K, Nn):
    """T_0 = TS[0,t] given; fill T_k^{o,i} (dcrnn.py:85-106): T_1 = P T_0, T_k = 2 P T_{k-1} - T_0 (Tx_0 is never
    advanced in the reference, dcrnn.py:106 — reproduced).  Segment order: [T0, T1o, T1i, T2o, T2i, ...]."""
    T0 = TS[0, t].view(Nn, -1)
    for k in range(1, K):
        for d, csr in enumerate((g.fwd_o, g.fwd_i)):
            src = T0 if k == 1 else TS[2 * (k - 1) - 1 + d, t].view(Nn, -1)
            dst = TS[2 * k - 1 + d, t].view(Nn, -1)
            if k == 1:
                spmm(csr, src, dst)
            else:
                spmm(csr, src, dst, T=T0, alpha=2.0, beta=-1.0)


def fold_backward_weight(Wst, K, C):
    """For K <= 3 the "- Tx_0" terms of the recursion only touch the LAST hop, so their adjoint
    (G_0 -= G_k^o + G_k^i) is linear in dPRE and folds into the segment-0 rows of the weight used by the
    feature-gradient GEMM: G_0 = dPRE (W_0 - sum_{k>=2} W_k^o + W_k^i)^T.  Removes 2(K-2) streaming passes over
    [M, C] per stack.  (K >= 4 keeps the explicit form: there G_k is updated before it is subtracted.)"""
    if K < 3 or K > 3:
        return Wst, False
    Wb = Wst.clone()
    for k in range(2, K):
        for d in range(2):
            j = 2 * k - 1 + d
            Wb[0:C] -= Wst[j * C:(j + 1) * C]
    return Wb, True


def _stack_bwd(g, G, K, Nn, folded=False):
    """Adjoint of _stack_fwd on G [S][M][C] (in place); on exit G[0] holds d/dT_0.  `folded`: the G_0 -= G_k terms
    were already applied through fold_backward_weight."""
    G0 = G[0].view(Nn, -1)
    for k in range(K - 1, 1, -1):
        for d, csr in enumerate((g.bwd_o, g.bwd_i)):
            Gk = G[2 * k - 1 + d].view(Nn, -1)
            Gp = G[2 * (k - 1) - 1 + d].view(Nn, -1)
            spmm(csr, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
            if not folded:
                axpby2d(G0, Gk, -1.0, G0, 1.0)
    if K > 1:
        for d, csr in enumerate((g.bwd_o, g.bwd_i)):
            spmm(csr, G[1 + d].view(Nn, -1), G0, T=G0, alpha=1.0, beta=1.0)


def slab_fits(g, C, K):
    """True when the LDS-resident one-launch diffusion stack (pgt_dconv_stack_slab_f32) covers this shape."""
    if K < 2:
        return False
    lib = _lib.get_lib()
    return bool(lib._pgt_dconv_stack_slab_fits(g.N, int(C), int(K), g.E, g.E))


def slab_plan(g, C, K, n_samples=1 << 20):
    """(column windows per sample, workgroups per CU, threads, tasks per thread) of the stack launch for this shape and batch
    (pgt_dconv_stack_slab_plan); (1, 1, 1024, 0) = the whole-sample kernels, zeros = not supported."""
    lib = _lib.get_lib()
    out = (ctypes.c_int32 * 4)()
    lib.call("pgt_dconv_stack_slab_plan", g.N, int(n_samples), int(C), int(K), g.E, g.E, out)
    return tuple(out)


def _slab_fwd(g, TS0, seg_stride, n_samples, C, K):
    """TS0: the [n_samples*N, C] block of segment 0 (batch-major rows); the other segments follow at seg_stride."""
    lib = _lib.get_lib()
    so, si = g.fwd_o.struct(), g.fwd_i.struct()
    work = (5 if K >= 3 else 3) * 4 * TS0.numel() if KERNEL_TIMER else 0
    _timed("stack", work, lambda: lib.call(
        "pgt_dconv_stack_slab_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, g.N, n_samples, C, K, ptr(TS0),
        seg_stride, stream_of(lib, TS0)))


def _slab_bwd(g, G0, seg_stride, n_samples, C, K, folded):
    lib = _lib.get_lib()
    so, si = g.bwd_o.struct(), g.bwd_i.struct()
    work = (6 if K >= 3 else 4) * 4 * G0.numel() if KERNEL_TIMER else 0
    _timed("stack", work, lambda: lib.call(
        "pgt_dconv_stack_slab_bwd_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, g.N, n_samples, C, K, ptr(G0),
        seg_stride, int(bool(folded)), stream_of(lib, G0)))


class _StackWeight(torch.autograd.Function):
    """DConv weight [2,K,C,O] -> [(2K-1)*C, O] with three copies forward and three backward (torch's slice / cat graph
    of the same rearrangement costs ~30 tiny launches per weight and backward pass)."""

    @staticmethod
    def forward(ctx, weight):
        _, K, C, O = weight.shape
        out = torch.empty((2 * K - 1) * C, O, dtype=weight.dtype, device=weight.device)
        torch.add(weight[0, 0], weight[1, 0], out=out[:C])
        if K > 1:
            out[C:].view(K - 1, 2, C, O).copy_(weight[:, 1:].permute(1, 0, 2, 3))
        ctx.shape = weight.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        _, K, C, O = ctx.shape
        dw = torch.empty(ctx.shape, dtype=dout.dtype, device=dout.device)
        dw[0, 0].copy_(dout[:C])
        dw[1, 0].copy_(dout[:C])
        if K > 1:
            dw[:, 1:].copy_(dout[C:].reshape(K - 1, 2, C, O).permute(1, 0, 2, 3))
        return dw


def stack_weight(weight):
    """DConv weight [2,K,C,O] -> stacked [(2K-1)*C, O] matching the segment order of _stack_fwd.
    Segment 0 carries W[0,0] + W[1,0] (the reference computes X@W[0,0] + X@W[1,0], dcrnn.py:81-83)."""
    return _StackWeight.apply(weight)


class CellWeightsFunction(torch.autograd.Function):
    """conv_x_{z,r,h}.weight [2, K, C, O] (+ the z / r biases) -> (Wzr [(2K-1) C, 2O], bzr [2O] | None, Wh [(2K-1) C, O]):
    the stacked operands of the two gate products, one launch forward and one backward (pgt_dcrnn_pack_weights_f32)."""

    @staticmethod
    def forward(ctx, Wz, Wr, Wh, bz, br):
        lib = _lib.get_lib()
        for t, n in ((Wz, "conv_x_z.weight"), (Wr, "conv_x_r.weight"), (Wh, "conv_x_h.weight")):
            check_tensor(lib, t, n)
        _, K, C, O = Wz.shape
        if Wr.shape != Wz.shape or Wh.shape != Wz.shape or (bz is None) != (br is None):
            raise ValueError("CellWeightsFunction: the three convolutions must have one shape and agree on bias")
        dev = Wz.device
        S = 2 * K - 1
        Wzr = torch.empty(S * C, 2 * O, dtype=F32, device=dev)
        Whs = torch.empty(S * C, O, dtype=F32, device=dev)
        bzr = torch.empty(2 * O, dtype=F32, device=dev) if bz is not None else None
        CellWeightsFunction.repack((Wz, Wr, Wh, bz, br), (Wzr, bzr, Whs))
        ctx.dims = (K, C, O)
        ctx.has_bias = bz is not None
        if bzr is None:
            return Wzr, None, Whs
        return Wzr, bzr, Whs

    @staticmethod
    def repack(params, packed):
        """The pack launch alone, into operands that already exist (nn/_states.py packed_once refreshes cached operands with it)."""
        lib = _lib.get_lib()
        Wz, Wr, Wh, bz, br = params
        Wzr, bzr, Whs = packed
        _, K, C, O = Wz.shape
        Wzc, Wrc, Whc = Wz.contiguous(), Wr.contiguous(), Wh.contiguous()
        lib.call("pgt_dcrnn_pack_weights_f32", ptr(Wzc), ptr(Wrc), ptr(Whc), ptr(bz.contiguous() if bz is not None else None),
                 ptr(br.contiguous() if br is not None else None), K, C, O, ptr(Wzr), ptr(bzr), ptr(Whs), stream_of(lib, Wzr))

    @staticmethod
    def backward(ctx, dWzr, dbzr, dWhs):
        lib = _lib.get_lib()
        K, C, O = ctx.dims
        ref = dWzr if dWzr is not None else dWhs
        if ref is None:
            return None, None, None, None, None
        dev = ref.device
        dWz, dWr, dWh = (torch.empty(2, K, C, O, dtype=F32, device=dev) for _ in range(3))
        dbz = dbr = None
        if ctx.has_bias:
            dbz, dbr = torch.empty(O, dtype=F32, device=dev), torch.empty(O, dtype=F32, device=dev)
            if dbzr is None:
                dbzr = torch.zeros(2 * O, dtype=F32, device=dev)
        lib.call("pgt_dcrnn_unpack_weight_grads_f32", ptr(dWzr.contiguous() if dWzr is not None else None),
                 ptr(dbzr.contiguous() if (ctx.has_bias and dbzr is not None) else None),
                 ptr(dWhs.contiguous() if dWhs is not None else None), K, C, O, ptr(dWz), ptr(dWr), ptr(dWh), ptr(dbz), ptr(dbr),
                 stream_of(lib, dWz))
        return dWz, dWr, dWh, dbz, dbr


def cell_k1_fits(N, Fin, O):
    """Whether the one-launch K = 1 cell (csrc/small_cell.hip) takes this shape (else: the general path)."""
    return bool(_lib.get_lib()._pgt_dcrnn_cell_k1_fits(int(N), int(Fin), int(O)))


class DCRNNCellK1Function(torch.autograd.Function):
    """DCRNN(in, out, K = 1) cell step, one launch forward and one backward (pgt_dcrnn_cell_k1_f32; dcrnn.py:79-82 +
    172-192): X [N, in], H [N, out] | None, the three convolutions' parameters as they are ([2, 1, in + out, out], [out])."""

    @staticmethod
    def forward(ctx, X, H, Wz, Wr, Wh, bz, br, bh):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        N, Fin = X.shape
        O = Wz.shape[3]
        if Wz.shape != (2, 1, Fin + O, O) or Wr.shape != Wz.shape or Wh.shape != Wz.shape:
            raise ValueError(f"DCRNN cell (K = 1): weights must be [2, 1, {Fin + O}, {O}], got {tuple(Wz.shape)}")
        if X.stride(1) != 1:
            X = X.contiguous()
        if H is not None:
            check_tensor(lib, H, "H")
            if H.shape != (N, O):
                raise ValueError(f"H must be [{N}, {O}], got {tuple(H.shape)}")
            if H.stride(1) != 1:
                H = H.contiguous()
        Wz, Wr, Wh = Wz.contiguous(), Wr.contiguous(), Wh.contiguous()
        out = torch.empty(N, O, dtype=F32, device=X.device)
        saved = torch.empty(N, 3 * O, dtype=F32, device=X.device)
        lib.call("pgt_dcrnn_cell_k1_f32", ptr(X), X.stride(0) if N > 1 else Fin, ptr(H),
                 (H.stride(0) if N > 1 else O) if H is not None else 0, ptr(Wz), ptr(Wr), ptr(Wh), ptr(bz), ptr(br), ptr(bh),
                 ptr(out), O, ptr(saved), N, Fin, O, stream_of(lib, X))
        ctx.save_for_backward(X, H, Wz, Wr, Wh, saved)
        ctx.has_bias = (bz is not None, br is not None, bh is not None)
        return out

    @staticmethod
    def backward(ctx, G):
        lib = _lib.get_lib()
        X, H, Wz, Wr, Wh, saved = ctx.saved_tensors
        N, Fin = X.shape
        O = Wz.shape[3]
        C = Fin + O
        if G.stride(1) != 1 or (N > 1 and G.stride(0) < O):
            G = G.contiguous()
        need = ctx.needs_input_grad
        dev = X.device
        # one allocation: the three weight gradients, the three bias gradients, the kernel's scratch
        nW = 2 * C * O
        buf = torch.empty(3 * nW + 3 * O + N * 3 * O, dtype=F32, device=dev)
        dWz, dWr, dWh = (buf[i * nW:(i + 1) * nW].view(2, 1, C, O) for i in range(3))
        dbz, dbr, dbh = (buf[3 * nW + i * O:3 * nW + (i + 1) * O] if ctx.has_bias[i] else None for i in range(3))
        dP = buf[3 * nW + 3 * O:]
        dX = torch.empty(N, Fin, dtype=F32, device=dev) if need[0] else None
        dH = torch.empty(N, O, dtype=F32, device=dev) if (H is not None and need[1]) else None
        lib.call("pgt_dcrnn_cell_k1_bwd_f32", ptr(G), G.stride(0) if N > 1 else O, ptr(X), X.stride(0) if N > 1 else Fin,
                 ptr(H), (H.stride(0) if N > 1 else O) if H is not None else 0, ptr(Wz), ptr(Wr), ptr(Wh), ptr(saved),
                 ptr(dX), Fin, ptr(dH), O, ptr(dWz), ptr(dWr), ptr(dWh), ptr(dbz), ptr(dbr), ptr(dbh), ptr(dP), N, Fin, O,
                 stream_of(lib, G))
        return dX, dH, dWz, dWr, dWh, dbz, dbr, dbh


class DConvFunction(torch.autograd.Function):
    """H = DConv(X) for node-major X [N*B, C]: diffusion stack (SpMM) + one segmented MFMA GEMM."""

    @staticmethod
    def forward(ctx, X, Wst, bias, g, K, B):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        M, C = X.shape
        Nn = g.N
        if M != Nn * B:
            raise ValueError(f"X has {M} rows, expected num_nodes*B = {Nn * B}")
        S = 2 * K - 1
        O = Wst.size(1)
        TS = torch.empty(S, 1, M, C, dtype=F32, device=X.device)
        copy2d(TS[0, 0], X)
        slab = B == 1 and slab_fits(g, C, K)      # one sample: batch-major == node-major
        if slab:
            _slab_fwd(g, TS[0, 0], M * C, 1, C, K)
        else:
            _stack_fwd(g, TS, 0, K, Nn)
        Wc = Wst.contiguous()
        out = torch.empty(M, O, dtype=F32, device=X.device)
        gemm(TS, C, M * C, S, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K, ctx.B, ctx.slab = g, K, B, slab
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc)
        return out

    @staticmethod
    def backward(ctx, dH):
        TS, Wc = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        S, _, M, C = TS.shape
        O = Wc.size(1)
        dH = dH.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros_like(Wc)
            db = torch.zeros(O, dtype=F32, device=dH.device) if ctx.has_bias else None
            gemm_tn_acc(TS, C, M * C, S, C, dH, O, dW, O, db, M, O)
        if ctx.needs_input_grad[0]:
            G = torch.empty(S, M, C, dtype=F32, device=dH.device)
            Wb, folded = fold_backward_weight(Wc, K, C)
            gemm(dH, O, 0, 1, O, Wb, 1, O, G, C, M * C, C, None, M, S * C)
            if ctx.slab:
                _slab_bwd(g, G[0], M * C, 1, C, K, folded)
            else:
                _stack_bwd(g, G, K, g.N, folded)
            dX = G[0]
        return dX, dW, db, None, None, None


class _StateLayout:
    """The [M, O] slice of time step t inside a contiguous [B, T, N, O] tensor, as RowMaps: batch-major rows
    m = b*N + n -> period N, stride_hi T*N*O, ld O; node-major rows m = n*B + b -> period B, stride_hi O, ld T*N*O."""

    def __init__(self, tensor, T, N, B, O, batch_major):
        if tensor.shape != (B, T, N, O) or not tensor.is_contiguous():
            raise ValueError(f"expected a contiguous [B, T, N, O] = {(B, T, N, O)} tensor, got {tuple(tensor.shape)}")
        self.flat, self.step_stride, self.O = tensor.view(-1), N * O, O
        self.args = (O, N, T * N * O) if batch_major else (T * N * O, B, O)      # (ld, period, stride_hi)

    def step(self, t):
        ld, period, hi = self.args
        return RowMap(self.flat[t * self.step_stride:], ld, period, hi, self.O)
SEQ_SMALL_MAX_O = int(os.environ.get("PGT_SEQ_SMALL_MAX_O", "8"))
USE_SEQ_SMALL = os.environ.get("PGT_SEQ_SMALL", "1") != "0"


def seq_small_fits(g, Fin, O, K):
    """Whether pgt_dcrnn_seq_small_f32 takes this graph / width (per-sample state within a workgroup's LDS)."""
    return bool(_lib.get_lib()._pgt_dcrnn_seq_small_fits(g.N, g.E, g.E, int(Fin), int(O), int(K)))


class DCRNNSeqSmallFunction(torch.autograd.Function):
    """BatchedDCRNN.forward / a DCRNN cell step for small graphs and narrow states: the whole sequence of a sample in one
    workgroup, one launch forward and one backward (csrc/seq_small.hip).  X [B, T, N, Fin], H0 [B, N, O] | None ->
    [B, T, N, O] (the reference's layout); Wzr / bzr / Wh / bh = the stacked operands of CellWeightsFunction."""

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        if X.dim() != 4:
            raise ValueError(f"DCRNNSeqSmallFunction: X must be [B, T, N, in], got {tuple(X.shape)}")
        B, T, N, Fin = X.shape
        O = Wh.size(1)
        S, C = 2 * K - 1, Fin + O
        # the C entry point only null-checks its pointers: a wrong width here would be an out-of-bounds device read
        if N != g.N:
            raise ValueError(f"X has {N} nodes, the graph {g.N}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O):
            raise ValueError(f"DCRNNSeqSmallFunction: inconsistent operand shapes: X has {Fin} input channels, the stacked weights "
                             f"{tuple(Wzr.shape)} / {tuple(Wh.shape)} expect in + out = {Wzr.size(0) // S if S else 0} with out = {O} "
                             f"(K = {K})")
        if (bzr is not None and bzr.shape != (2 * O,)) or (bh is not None and bh.shape != (O,)):
            raise ValueError("DCRNNSeqSmallFunction: inconsistent bias shapes")
        Xc = X.contiguous()
        H0c = None
        if H0 is not None:
            check_tensor(lib, H0, "H")
            if H0.shape != (B, N, O):
                raise ValueError(f"H must be {(B, N, O) if B > 1 else (N, O)}, got {tuple(H0.shape[1:] if B == 1 and H0.dim() == 3 else H0.shape)}")
            H0c = H0.contiguous()
        dev = X.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        out = torch.empty(B, T, N, O, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad)
        save = None
        if need:
            per = int(lib._pgt_dcrnn_seq_small_save_floats(N, Fin, O, K))
            save = torch.empty(B * T * per, dtype=F32, device=dev)
        so, si = g.fwd_o.struct(), g.fwd_i.struct()
        _timed("seq_small", 4.0 * (Xc.numel() + out.numel() + (save.numel() if need else 0)) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_dcrnn_seq_small_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, N, ptr(Xc), T * N * Fin, N * Fin, ptr(H0c),
            ptr(Wzr_c), ptr(bzr), ptr(Wh_c), ptr(bh), B, T, Fin, O, K, ptr(out), T * N * O, N * O, ptr(save), stream_of(lib, out)))
        ctx.g, ctx.K = g, K
        ctx.has = (H0 is not None, bzr is not None, bh is not None)
        if need:
            ctx.save_for_backward(out, H0c, save, Wzr_c, Wh_c)
        ctx.dims = (B, T, N, Fin, O)
        return out

    @staticmethod
    def backward(ctx, dOut):
        lib = _lib.get_lib()
        out, H0c, save, Wzr_c, Wh_c = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        B, T, N, Fin, O = ctx.dims
        C, S = Fin + O, 2 * K - 1
        dev = out.device
        dOc = dOut.contiguous()
        need = ctx.needs_input_grad
        dX = torch.empty(B, T, N, Fin, dtype=F32, device=dev) if need[0] and Fin > 0 else None
        dH0 = torch.empty(B, N, O, dtype=F32, device=dev) if (ctx.has[0] and need[1]) else None
        nW = S * C * 3 * O + 3 * O
        part = torch.zeros(B, nW, dtype=F32, device=dev)
        to, ti = g.bwd_o.struct(), g.bwd_i.struct()
        _timed("seq_small", 4.0 * (dOc.numel() + out.numel() + save.numel()) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_dcrnn_seq_small_bwd_f32", ctypes.byref(to), ctypes.byref(ti), g.E, g.E, N, ptr(dOc), T * N * O, N * O, ptr(out),
            T * N * O, N * O, ptr(H0c), ptr(save), ptr(Wzr_c), ptr(Wh_c), B, T, Fin, O, K, ptr(dX), T * N * Fin, N * Fin, ptr(dH0),
            ptr(part), stream_of(lib, out)))
        dW = part.sum(dim=0) if B > 1 else part[0]            # the samples in index order: deterministic
        n1, n2 = S * C * 2 * O, S * C * 3 * O
        dWzr, dWh = dW[:n1].view(S * C, 2 * O), dW[n1:n2].view(S * C, O)
        dbzr = dW[n2:n2 + 2 * O] if ctx.has[1] else None
        dbh = dW[n2 + 2 * O:] if ctx.has[2] else None
        return dX, dH0, dWzr, dbzr, dWh, dbh, None, None


class DCRNNSeqFunction(torch.autograd.Function):
    """T steps of the DCRNN GRU cell (dcrnn.py:172-219 / :406-475) on node-major rows m = n*B + b.

    X [T, M, F_in], H0 [M, O] -> Hall [T, M, O].  Per step: [X_t, H] -> diffusion stack -> one MFMA GEMM for the
    update and reset gates together (the reference aggregates [X,H] twice) -> sigmoid / H*R -> stack -> GEMM ->
    tanh / blend.  Backward is hand-written BPTT on the transposed operators; the weight gradients of all T steps
    are one split-K GEMM over the saved stacks.
    """

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K, B, batch_major=False, btno=False):
        """batch_major: rows m = b*N + n and the diffusion stacks run as ONE LDS-resident launch per conv
        (pgt_dconv_stack_slab_f32; requires slab_fits); otherwise rows m = n*B + b and one launch per hop.
        btno: the hidden states are returned as the reference returns them, a contiguous [B, T, N, O] tensor
        (torch.stack(outputs, dim=1), dcrnn.py:463-475): the candidate-gate epilogue of step t stores H_t straight into
        out[:, t] through a two-level row map (pgt_rowmap) and the backward pass reads the incoming gradient and the
        previous states in place from that layout — no transposition pass either way."""
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, H0, "H0")
        X = X.contiguous()
        T, M, Fin = X.shape
        O = Wh.size(1)
        C = Fin + O
        S = 2 * K - 1
        Nn = g.N
        if M != Nn * B:
            raise ValueError(f"X has {M} rows per step, expected num_nodes*B = {Nn * B}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O) or H0.shape != (M, O):
            raise ValueError("DCRNNSeqFunction: inconsistent operand shapes")
        dev = X.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        TSzr = torch.empty(S, T, M, C, dtype=F32, device=dev)
        TSh = torch.empty(S, T, M, C, dtype=F32, device=dev)
        ZR = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(T, M, O, dtype=F32, device=dev)
        Hout = torch.empty((B, T, Nn, O) if btno else (T, M, O), dtype=F32, device=dev)
        H0c = H0.contiguous()
        state = _StateLayout(Hout, T, Nn, B, O, batch_major) if btno else None
        seg = T * M * C
        slab = bool(batch_major) or (B == 1 and slab_fits(g, C, K))
        if slab and K > 1 and not slab_fits(g, C, K):
            raise ValueError("DCRNNSeqFunction: batch-major rows need the LDS-resident stack (slab_fits)")

        def stack(TSx, t):
            if K < 2:
                return
            if slab:
                _slab_fwd(g, TSx[0, t], seg, B, C, K)
            else:
                _stack_fwd(g, TSx, t, K, Nn)

        # the input columns of segment 0 of both stacks (all T steps) and H0 into step 0 of the gate stack: one launch
        lib.call("pgt_dcrnn_stage_f32", ptr(X), ptr(H0c), T, M, Fin, O, ptr(TSzr[0]), ptr(TSh[0]), stream_of(lib, X))
        fuse = FUSE_GATE_EPILOGUES and O % 4 == 0
        for t in range(T):
            # H_{t-1}: the plain [M, O] state, or (btno: the states live in the [B, T, N, O] result) the hidden columns of
            # this step's stack segment 0, which the previous step's blend wrote
            Hp = H0c if t == 0 else (TSzr[0, t][:, Fin:] if btno else Hout[t - 1])
            Hnext = TSzr[0, t + 1][:, Fin:] if t + 1 < T else None
            Ht = state.step(t) if btno else Hout[t]
            stack(TSzr, t)
            if fuse:
                gemm_gru_zr(TSzr[0, t], C, seg, S, C, Wzr_c, 2 * O, 1, bzr, ZR[t], Hp, TSh[0, t], Fin)
                stack(TSh, t)
                gemm_gru_h(TSh[0, t], C, seg, S, C, Wh_c, O, 1, bh, HT[t], ZR[t], Hp, Ht, Hnext)
            else:
                gemm(TSzr[0, t], C, seg, S, C, Wzr_c, 2 * O, 1, ZR[t], 2 * O, 0, 2 * O, bzr, M, 2 * O)
                _gru_zr(ZR[t], Hp, TSh[0, t], Fin)
                stack(TSh, t)
                gemm(TSh[0, t], C, seg, S, C, Wh_c, O, 1, HT[t], O, 0, O, bh, M, O)
                _gru_h(HT[t], ZR[t], Hp, Ht, Hnext)
        ctx.g, ctx.K, ctx.B, ctx.Fin, ctx.slab = g, K, B, Fin, slab
        ctx.btno, ctx.batch_major = btno, bool(batch_major)
        ctx.has_bias = (bzr is not None, bh is not None)
        ctx.save_for_backward(TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dOut):
        if _seq64_adjoint_applies(ctx):
            # hidden 64, the input is data: all T steps of the adjoint in ONE launch (csrc/seq64.hip) — faster than the six
            # launches per step at every batch size (B = 64: 1.32 -> 1.11 ms, B = 1024: 8.8 -> 6.9 ms)
            return (None,) + _seq64_adjoint(ctx, dOut) + (None,) * 5
        TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c = ctx.saved_tensors
        g, K, Fin = ctx.g, ctx.K, ctx.Fin
        S, T, M, C = TSzr.shape
        O = HT.size(2)
        Nn = g.N
        dev = dOut.device
        dOut = dOut.contiguous()
        if ctx.btno:      # gradient and states in the reference's [B, T, N, O] layout, read in place step by step
            grad_in = _StateLayout(dOut, T, Nn, ctx.B, O, ctx.batch_major)
            states = _StateLayout(Hout, T, Nn, ctx.B, O, ctx.batch_major)
        need_x = ctx.needs_input_grad[0]
        dX = torch.zeros(T, M, Fin, dtype=F32, device=dev) if need_x else None
        dH = torch.zeros(M, O, dtype=F32, device=dev)      # running d/dH_t
        dPzr = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        dPh = torch.empty(T, M, O, dtype=F32, device=dev)
        Wh_b, folded = fold_backward_weight(Wh_c, K, C)
        Wzr_b, _ = fold_backward_weight(Wzr_c, K, C)
        # When the input needs no gradient (the usual case: X is data), only the hidden-state columns of the stack
        # gradient are ever read, and the adjoint of the stack acts on every column independently.  The whole backward
        # stack then runs on those O columns alone: the feature-gradient GEMMs use the weight rows of the hidden
        # columns (S*O = 320 output columns instead of S*C = 330, i.e. 2.5 instead of 3 128-wide column tiles) and
        # write [S][M][O] segments -- 256-byte rows, float4 stores, no 8-byte holes where the input columns would be
        # -- and the stack adjoint reads and writes 64- instead of 66-wide rows.
        skip_x = (SKIP_INPUT_COLUMNS_WHEN_UNUSED and not need_x and Fin > 0 and O % 4 == 0 and S > 1)
        Cb, Fb = (O, 0) if skip_x else (C, Fin)                      # width of the stack gradient, its first H column
        G = torch.empty(S, M, Cb, dtype=F32, device=dev)
        if skip_x:
            WhH = Wh_b.view(S, C, O)[:, Fin:, :].reshape(S * O, O).contiguous()
            WzrH = Wzr_b.view(S, C, 2 * O)[:, Fin:, :].reshape(S * O, 2 * O).contiguous()
            NH = S * O
            n1 = (NH // 128) * 128 if (SPLIT_FEATURE_GRADIENT and NH % 128 != 0 and NH > 128 and ((NH // 128) * 128) % O == 0) else NH
            # tall batches: ONE product over all S*O <= 320 columns on the symmetric split-bf16 kernel (dP is read once;
            # column blocks 8 and 9 ride along as second blocks) instead of 256 columns + a 64-column remainder
            # (from 8 192 rows, the split-bf16 kernels' own floor: at B = 64, M = 13 248, the one product is 4.5 % of the step
            # faster than 256 + 64 since the round-3 kernels, 2.55 -> 2.44 ms)
            if ONE_FEATURE_GRADIENT and M >= ONE_FEATURE_GRADIENT_MIN_ROWS and 128 < NH <= 320 and NH % 32 == 0 and 2 * O <= 128 and O % 32 == 0:
                n1 = NH

        def feature_grad(dP, Wfull, WH, Kd):
            """G[s] = dP W_s^T for every stack segment (the stack adjoint consumes G in place)."""
            if not skip_x:
                gemm(dP, Kd, 0, 1, Kd, Wfull, 1, Kd, G, C, M * C, C, None, M, S * C)
                return
            gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, O, M * O, O, None, M, n1)
            if n1 < NH:                                              # the narrow remainder: 64-wide tiles, no padding
                gemm(dP, Kd, 0, 1, Kd, WH[n1:], 1, Kd, G[n1 // O], O, M * O, O, None, M, NH - n1)
        B = ctx.B
        seg = T * M * C
        need_wzr = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        need_wh = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dWzr = torch.zeros_like(Wzr_c) if need_wzr else None
        dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if (need_wzr and ctx.has_bias[0]) else None
        dWh = torch.zeros_like(Wh_c) if need_wh else None
        dbh = torch.zeros(O, dtype=F32, device=dev) if (need_wh and ctx.has_bias[1]) else None
        overlap = OVERLAP_WEIGHT_GRADIENTS and dev.type == "cuda" and KERNEL_TIMER is None and (need_wzr or need_wh)
        if overlap:
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)          # dW / db zero-fills and the saved stacks are ready

        def weight_grads(t):
            """dW += stack_t^T dPRE_t for one time step (pgt_gemm_tn_acc_f32 accumulates)."""
            if need_wzr:
                gemm_tn_acc(TSzr[0, t], C, seg, S, C, dPzr[t], 2 * O, dWzr, 2 * O, dbzr, M, 2 * O)
            if need_wh:
                gemm_tn_acc(TSh[0, t], C, seg, S, C, dPh[t], O, dWh, O, dbh, M, O)

        def stack_bwd():
            if K < 2:
                return
            if ctx.slab:
                _slab_bwd(g, G[0], M * Cb, B, Cb, K, folded)
            else:
                _stack_bwd(g, G, K, Nn, folded)

        for t in range(T - 1, -1, -1):
            Hp = H0c if t == 0 else (states.step(t - 1) if ctx.btno else Hout[t - 1])
            # d/dH_t = dOut[t] + running state gradient, summed inside the gate-backward kernel
            # (+ the later step's gate-stack gradient of H, still sitting in G[0]: no accumulation pass of its own)
            _gru_h_bwd(grad_in.step(t) if ctx.btno else dOut[t], ZR[t], Hp, HT[t], dPh[t], dPzr[t], dH, accumulate=False,
                       dHn2=dH, dHn3=None if (t == T - 1 or not FOLD_STATE_GRADIENT) else G[0][:, Fb:])
            # candidate conv: dT = dPh Wh^T ; adjoint of the stack
            feature_grad(dPh[t], Wh_b, WhH if skip_x else None, O)
            stack_bwd()
            _gru_zr_bwd(G[0], Fb, ZR[t], Hp, dPzr[t], dH)
            if overlap:                     # dPh[t], dPzr[t] are final: their weight gradients go to the side stream
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    weight_grads(t)
            if need_x:
                copy2d(dX[t], G[0][:, :Fin])
            # gate convs
            feature_grad(dPzr[t], Wzr_b, WzrH if skip_x else None, 2 * O)
            stack_bwd()
            if not FOLD_STATE_GRADIENT:
                add2d(dH, G[0][:, Fb:])
            if need_x:
                add2d(dX[t], G[0][:, :Fin])
        if ctx.needs_input_grad[1] and FOLD_STATE_GRADIENT:
            add2d(dH, G[0][:, Fb:])              # d/dH0: the first step's gate-stack gradient joins here
        if overlap:
            main.wait_stream(side)
        else:
            if need_wzr:
                gemm_tn_acc(TSzr, C, seg, S, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, T * M, 2 * O)
            if need_wh:
                gemm_tn_acc(TSh, C, seg, S, C, dPh, O, dWh, O, dbh, T * M, O)
        dH0 = dH if ctx.needs_input_grad[1] else None
        return dX, dH0, dWzr, dbzr, dWh, dbh, None, None, None, None, None


# hidden width 64 on a graph whose block fits a CU's LDS: the whole T-step forward of every sample in ONE launch (csrc/seq64.hip);
# PGT_SEQ64=0 = the per-step launches of DCRNNSeqFunction (A/B)
USE_SEQ64 = os.environ.get("PGT_SEQ64", "1") != "0"
# smallest batch that takes it: a sample occupies ONE CU for the whole sequence, so below ~100 samples the per-step launches —
# which spread every step over all 256 CUs — are faster (B = 64: 0.82 ms against 0.98 ms forward; B = 256: 1.68 against 1.11)
SEQ64_MIN_BATCH = int(os.environ.get("PGT_SEQ64_MIN_B", "96"))
USE_SEQ64_BWD = os.environ.get("PGT_SEQ64_BWD", "1") != "0"      # 0: the per-step adjoint launches behind the one-launch forward (A/B)


def seq64_fits(g, Fin, O, K):
    """Whether pgt_dcrnn_seq64_f32 takes this graph / width: hidden 64, two input channels, K = 2 | 3, the sample's block + both
    operators + the weight ring within a CU's LDS — and finite operator coefficients (a node without incoming edges makes
    DConv's 1 / deg infinite, dcrnn.py:71-77: inf / nan placement is the general path's speciality, csrc/gemm_bx.hip)."""
    return bool(getattr(g, "finite", False)) and bool(_lib.get_lib()._pgt_dcrnn_seq64_fits(g.N, g.E, g.E, int(Fin), int(O), int(K)))


def _seq64_adjoint(ctx, dOut):
    """(dH0, dWzr, dbzr, dWh, dbh) of a hidden-64 sequence whose input is data: the whole BPTT in ONE launch (csrc/seq64.hip) on
    what either forward path saved (both stacks, Z | R, the candidates, the states in the reference's [B, T, N, O] layout,
    batch-major rows) + the two weight-gradient products over all T steps."""
    lib = _lib.get_lib()
    TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c = ctx.saved_tensors
    g, K = ctx.g, ctx.K
    S, T, M, C = TSzr.shape
    O = HT.size(2)
    N = g.N
    B, Fin = M // N, C - O
    dev = dOut.device
    dOc = dOut.contiguous()
    need = ctx.needs_input_grad
    dPzr = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
    dPh = torch.empty(T, M, O, dtype=F32, device=dev)
    dH0 = torch.empty(M, O, dtype=F32, device=dev) if need[1] else None
    Wp = torch.empty(int(lib._pgt_dcrnn_seq64_pack_floats(K)), dtype=F32, device=dev)
    lib.call("pgt_dcrnn_seq64_pack_bwd_f32", ptr(Wzr_c), ptr(Wh_c), Fin, K, ptr(Wp), stream_of(lib, Wp))
    nws = int(lib._pgt_dcrnn_seq64_bwd_ws_floats(N, B))
    ws = torch.empty(nws, dtype=F32, device=dev)
    to, ti = g.bwd_o.struct(), g.bwd_i.struct()
    work = 4.0 * (dOc.numel() + Hout.numel() + ZR.numel() + HT.numel() + dPzr.numel() + dPh.numel()) if KERNEL_TIMER else 0
    _timed("seq64", work, lambda: lib.call(
        "pgt_dcrnn_seq64_bwd_f32", ctypes.byref(to), ctypes.byref(ti), g.E, g.E, N, ptr(dOc), T * N * O, N * O, ptr(Hout), T * N * O,
        N * O, ptr(H0c), ptr(ZR), ptr(HT), ptr(Wp), B, T, Fin, K, ptr(dPzr), ptr(dPh), ptr(dH0), ptr(ws), nws,
        stream_of(lib, dPh)), tag=("bwd", B, T, N))
    seg = T * M * C
    dWzr = dbzr = dWh = dbh = None
    if need[2] or need[3]:
        dWzr = torch.zeros_like(Wzr_c)
        dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if ctx.has_bias[0] else None
        gemm_tn_acc(TSzr, C, seg, S, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, T * M, 2 * O)
    if need[4] or need[5]:
        dWh = torch.zeros_like(Wh_c)
        dbh = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias[1] else None
        gemm_tn_acc(TSh, C, seg, S, C, dPh, O, dWh, O, dbh, T * M, O)
    return dH0, dWzr, dbzr, dWh, dbh


def _seq64_adjoint_applies(ctx):
    """The one-launch adjoint takes a DCRNNSeqFunction / DCRNNSeq64Function context: hidden 64 with two input channels on a graph
    pgt_dcrnn_seq64_fits covers, batch-major rows with the states in the reference's layout, the input being data."""
    if not (USE_SEQ64 and USE_SEQ64_BWD) or ctx.needs_input_grad[0] or not (ctx.btno and ctx.batch_major and ctx.slab):
        return False
    TSzr, _, _, HT = ctx.saved_tensors[:4]
    return seq64_fits(ctx.g, TSzr.size(3) - HT.size(2), HT.size(2), ctx.K)


class DCRNNSeq64Function(torch.autograd.Function):
    """BatchedDCRNN.forward at hidden width 64 (dcrnn.py:429-475): X [B, T, N, Fin], H0 [B * N, O] | None -> [B, T, N, O] in ONE
    launch for all T steps (csrc/seq64.hip: one workgroup per sample, diffusion terms and products never leave the CU); the
    launch leaves behind exactly what DCRNNSeqFunction.forward saves (both stacks, Z | R, the candidates, the states in the
    reference's layout), so the hand-written BPTT of DCRNNSeqFunction.backward runs on it unchanged."""

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        if X.dim() != 4:
            raise ValueError(f"DCRNNSeq64Function: X must be [B, T, N, in], got {tuple(X.shape)}")
        B, T, N, Fin = X.shape
        O = Wh.size(1)
        S, C = 2 * K - 1, Fin + O
        M = B * N
        if N != g.N:
            raise ValueError(f"X has {N} nodes, the graph {g.N}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O):
            raise ValueError(f"DCRNNSeq64Function: inconsistent operand shapes: X has {Fin} input channels, the stacked weights "
                             f"{tuple(Wzr.shape)} / {tuple(Wh.shape)} (K = {K})")
        if (bzr is not None and bzr.shape != (2 * O,)) or (bh is not None and bh.shape != (O,)):
            raise ValueError("DCRNNSeq64Function: inconsistent bias shapes")
        if not lib._pgt_dcrnn_seq64_fits(N, g.E, g.E, Fin, O, K):
            raise ValueError("DCRNNSeq64Function: shape not covered (seq64_fits)")
        dev = X.device
        Xc = X.contiguous()
        H0c = None
        if H0 is not None:
            check_tensor(lib, H0, "H0")
            if H0.shape != (M, O):
                raise ValueError(f"H0 must be {(M, O)}, got {tuple(H0.shape)}")
            H0c = H0.contiguous()
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        Wp = torch.empty(int(lib._pgt_dcrnn_seq64_pack_floats(K)), dtype=F32, device=dev)
        lib.call("pgt_dcrnn_seq64_pack_f32", ptr(Wzr_c), ptr(Wh_c), Fin, K, ptr(Wp), stream_of(lib, Wp))
        TSzr = torch.empty(S, T, M, C, dtype=F32, device=dev)
        TSh = torch.empty(S, T, M, C, dtype=F32, device=dev)
        ZR = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(T, M, O, dtype=F32, device=dev)
        Hout = torch.empty(B, T, N, O, dtype=F32, device=dev)
        so, si = g.fwd_o.struct(), g.fwd_i.struct()
        work = 4.0 * (Xc.numel() + Hout.numel() + TSzr.numel() + TSh.numel() + ZR.numel() + HT.numel()) if KERNEL_TIMER else 0
        _timed("seq64", work, lambda: lib.call(
            "pgt_dcrnn_seq64_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, N, ptr(Xc), T * N * Fin, N * Fin, ptr(H0c), ptr(Wp),
            ptr(Wzr_c), ptr(bzr), ptr(Wh_c), ptr(bh), B, T, Fin, K, ptr(Hout), T * N * O, N * O, ptr(TSzr), ptr(TSh), T * M * C, M * C,
            ptr(ZR), ptr(HT), stream_of(lib, Hout)), tag=("fwd", B, T, N))
        if any(ctx.needs_input_grad):
            if H0c is None:
                H0c = torch.zeros(M, O, dtype=F32, device=dev)
            ctx.g, ctx.K, ctx.B, ctx.Fin, ctx.slab = g, K, B, Fin, True
            ctx.btno, ctx.batch_major = True, True
            ctx.has_bias = (bzr is not None, bh is not None)
            ctx.dims = (B, T, N, Fin)
            ctx.save_for_backward(TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dOut):
        if not _seq64_adjoint_applies(ctx):
            # the input wants a gradient (or the A/B switch is off): the per-step adjoint launches on what the forward saved
            dX, dH0, dWzr, dbzr, dWh, dbh = DCRNNSeqFunction.backward(ctx, dOut)[:6]
            if dX is not None:                                 # [T, B N, Fin] (time-major steps of batch-major rows) -> X's layout
                B, T, N, Fin = ctx.dims
                dX = dX.view(T, B, N, Fin).permute(1, 0, 2, 3).contiguous()
            return dX, dH0, dWzr, dbzr, dWh, dbh, None, None
        return (None,) + _seq64_adjoint(ctx, dOut) + (None, None)
