"""Building blocks shared by the model families: aggregation, linear and read-out Functions.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""


import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import KERNEL_TIMER, _det_workspace, _rows, _timed, gemm, gemm_tn_acc, linear_fwd, spmm


# --------------------------------------------------------------------------------------------- generic building blocks

class SpmmFunction(torch.autograd.Function):
    """Y = alpha * A @ X on [n_rows, F] (propagate with aggr="add"); the gradient runs on the transposed operator."""

    @staticmethod
    def forward(ctx, X, fwd, bwd, alpha):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        Y = torch.empty_like(Xc)
        spmm(fwd, Xc, Y, alpha=alpha)
        ctx.bwd, ctx.alpha = bwd, alpha
        return Y

    @staticmethod
    def backward(ctx, dY):
        dYc = dY.contiguous()
        dX = torch.empty_like(dYc)
        spmm(ctx.bwd, dYc, dX, alpha=ctx.alpha)
        return dX, None, None, None


def propagate(g, X2d, alpha=1.0):
    """g: SymGraph (fwd/bwd).  X2d [N, F] -> A @ X2d with autograd."""
    return SpmmFunction.apply(X2d, g.fwd, g.bwd, float(alpha))


class LinearFunction(torch.autograd.Function):
    """Y[M,N] = X[M,K] @ W_kn[K,N] (+ bias) on the fp32 MFMA GEMM; W_kn may be any strided 2-D view (e.g. weight.t())."""

    @staticmethod
    def forward(ctx, X, W_kn, bias):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        Y = linear_fwd(Xc, W_kn, bias)
        ctx.save_for_backward(Xc, W_kn)
        ctx.has_bias = bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        Xc, W_kn = ctx.saved_tensors
        M, K = Xc.shape
        N = W_kn.size(1)
        dYc = dY.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            dX = torch.empty(M, K, dtype=F32, device=dYc.device)
            # dX = dY @ W_kn^T : B element (k' = n, n' = k) is W_kn[k, n]
            gemm(dYc, N, 0, 1, N, W_kn, W_kn.stride(1), W_kn.stride(0), dX, K, 0, K, None, M, K)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros(K, N, dtype=F32, device=dYc.device)
            db = torch.zeros(N, dtype=F32, device=dYc.device) if ctx.has_bias else None
            gemm_tn_acc(Xc, K, 0, 1, K, dYc, N, dW, N, db, M, N)
        return dX, dW, db


def linear(X2d, W_kn, bias=None):
    return LinearFunction.apply(X2d, W_kn, bias)


class ReadoutFunction(torch.autograd.Function):
    """The read-out the reference's models apply to the states of a recurrent layer — `linear(relu(h))` or `linear(h)` with a
    torch.nn.Linear of 1 .. 4 outputs (examples/indexBatching/tgcn/metr_la_main.py:43-45, examples/recurrent/dcrnn_example.py:27-31)
    — as one streaming pass each way (csrc/readout.hip): X [M, K] rows (the PRE-relu states), weight [N, K] as torch.nn.Linear
    holds it, bias [N] | None -> Y [M, N]; the adjoint reads X once and writes dX once (relu's mask applied in the same pass),
    weight / bias gradients from per-workgroup partial sums added in index order (deterministic)."""

    @staticmethod
    def forward(ctx, X, weight, bias, relu):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, weight, "weight")
        M, K = X.shape
        N = weight.size(0)
        if weight.shape != (N, K) or (bias is not None and bias.shape != (N,)):
            raise ValueError(f"read-out: weight must be [out, {K}] and bias [out], got {tuple(weight.shape)}")
        xp, ldx = _rows(X, "X")
        W = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        Y = torch.empty(M, N, dtype=F32, device=X.device)
        _timed("readout", 4.0 * M * (K + N) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_relu_linear_f32", xp, ldx, ptr(W), ptr(b), M, K, N, int(bool(relu)), ptr(Y), N, stream_of(lib, Y)), tag=("fwd", M, K, N))
        ctx.save_for_backward(X, W)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        X, W = ctx.saved_tensors
        M, K = X.shape
        N = W.size(0)
        dYc = dY.contiguous()
        xp, ldx = _rows(X, "X")
        need = ctx.needs_input_grad
        dX = torch.empty(M, K, dtype=F32, device=X.device) if need[0] else None
        dW = torch.empty(N, K, dtype=F32, device=X.device) if need[1] else None
        db = torch.empty(N, dtype=F32, device=X.device) if (ctx.has_bias and need[2]) else None
        nws = int(lib._pgt_relu_linear_bwd_ws_floats(K, N))
        ws = _det_workspace(X.device, nws)
        _timed("readout", 4.0 * M * (2 * K + N) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_relu_linear_bwd_f32", xp, ldx, ptr(dYc), N, ptr(W), M, K, N, int(ctx.relu), ptr(dX), K, ptr(dW), ptr(db), ptr(ws), nws,
            stream_of(lib, X)), tag=("bwd", M, K, N))
        return dX, dW, db, None


def readout_fits(x, weight, bias):
    """Whether the streaming read-out kernels take this call: fp32 rows of 4 .. 64 (a multiple of 4) floats that are 16-byte
    addressable, 1 .. 4 outputs."""
    if x.dim() < 2 or weight.dim() != 2 or x.size(-1) != weight.size(1):
        return False
    K, N = weight.size(1), weight.size(0)
    return bool(_lib.get_lib()._pgt_relu_linear_fits(int(K), int(N)))


def readout(x, weight, bias, relu):
    """linear(relu(x)) / linear(x) over the last dimension of x through ReadoutFunction (rows in memory order, no copy when x is
    contiguous)."""
    lead = x.shape[:-1]
    K = x.size(-1)
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    return ReadoutFunction.apply(x2, weight, bias, relu).view(*lead, weight.size(0))
