"""EvolveGCN-H / -O: the weight evolution Function.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""


import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32
I32 = torch.int32

from ._graphs import small_edges


def gcn_small_fits(N, E, Fi, Fo):
    return bool(_lib.get_lib()._pgt_gcn_small_fits(int(N), int(E), int(Fi), int(Fo)))


class GcnSmallFunction(torch.autograd.Function):
    """out = A_hat (x W) from the raw edge list in one launch each way (pgt_gcn_small_f32; evolvegcno.py:76-101)."""

    @staticmethod
    def forward(ctx, x, W, edges, improved, add_self_loops, normalize):
        lib = _lib.get_lib()
        check_tensor(lib, x, "x")
        check_tensor(lib, W, "W")
        N, Fi = x.shape
        if W.dim() != 2 or W.size(0) != Fi:
            raise ValueError(f"W must be [{Fi}, out], got {tuple(W.shape)}")
        if N != edges.N:
            raise ValueError(f"x has {N} rows, the graph {edges.N} nodes")
        Fo = W.size(1)
        if x.stride(1) != 1 or (N > 1 and x.stride(0) < Fi):
            x = x.contiguous()
        W = W.contiguous()
        out = torch.empty(N, Fo, dtype=F32, device=x.device)
        coef = torch.empty(edges.E + N, dtype=F32, device=x.device)
        lib.call("pgt_gcn_small_f32", ptr(edges.ei), ptr(edges.ew), edges.E, N, int(bool(improved)), int(bool(add_self_loops)),
                 int(bool(normalize)), ptr(x), x.stride(0) if N > 1 else Fi, ptr(W), Fi, Fo, ptr(out), ptr(coef),
                 ptr(edges.info), stream_of(lib, x))
        ctx.save_for_backward(x, W, coef)
        ctx.edges = edges
        ctx.flags = (int(bool(add_self_loops)), int(bool(normalize)))
        return out

    @staticmethod
    def backward(ctx, G):
        lib = _lib.get_lib()
        x, W, coef = ctx.saved_tensors
        edges = ctx.edges
        N, Fi = x.shape
        Fo = W.size(1)
        if G.stride(1) != 1 or (N > 1 and G.stride(0) < Fo):
            G = G.contiguous()
        dW = torch.empty(Fi, Fo, dtype=F32, device=x.device)
        dX = torch.empty(N, Fi, dtype=F32, device=x.device) if ctx.needs_input_grad[0] else None
        lib.call("pgt_gcn_small_bwd_f32", ptr(edges.ei), ptr(coef), edges.E, N, ctx.flags[0], ctx.flags[1], ptr(G),
                 G.stride(0) if N > 1 else Fo, ptr(x), x.stride(0) if N > 1 else Fi, ptr(W), Fi, Fo, ptr(dW), ptr(dX), Fi,
                 stream_of(lib, G))
        return dX, dW, None, None, None, None


def gcn_small(x, W, edge_index, edge_weight, improved=False, add_self_loops=True, normalize=True):
    return GcnSmallFunction.apply(x, W, small_edges(edge_index, edge_weight, x.size(0)), improved, add_self_loops, normalize)


# --------------------------------------------------------------------------------------------- EvolveGCN weight evolution

class EvolveWeightFunction(torch.autograd.Function):
    """W_t = GRU(summary(X_t), W_{t-1}) (EvolveGCN-H, evolvegcnh.py:93-100) or GRU(W_{t-1}, W_{t-1}) (EvolveGCN-O,
    evolvegcno.py:185-187) in ONE launch forward and ONE backward (pgt_evolve_weight(_bwd)_f32): top-k scoring / selection,
    the GRU cell on the 8 x 8 state and every gradient.  X may be None (O variant)."""

    @staticmethod
    def forward(ctx, X, p, Wih, Whh, bih, bhh, Wprev, k):
        lib = _lib.get_lib()
        pool = X is not None
        for t, n in ((Wih, "weight_ih"), (Whh, "weight_hh"), (Wprev, "weight")) + (((X, "X"), (p, "select.weight")) if pool else ()):
            check_tensor(lib, t, n)
        F_ = Wprev.size(-1)
        k = int(k)
        Wp = Wprev.reshape(-1, F_).contiguous()
        if Wp.size(0) != k or Wih.shape != (3 * F_, F_) or Whh.shape != (3 * F_, F_):
            raise ValueError("EvolveWeightFunction: the GRU's batch must be the k pooled rows (k == in_channels in the reference)")
        dev = Wp.device
        Xc = X.contiguous() if pool else None
        pc = p.reshape(-1).contiguous() if pool else None
        Wihc, Whhc = Wih.contiguous(), Whh.contiguous()
        Wnew = torch.empty(k, F_, dtype=F32, device=dev)
        perm = torch.empty(k, dtype=I32, device=dev)
        score = torch.empty(k, dtype=F32, device=dev)
        gates = torch.empty(4, k, F_, dtype=F32, device=dev)
        xt = torch.empty(k, F_, dtype=F32, device=dev)
        lib.call("pgt_evolve_weight_f32", ptr(Xc), Xc.stride(0) if pool else 0, Xc.size(0) if pool else 0, ptr(pc), ptr(Wihc),
                 ptr(Whhc), ptr(bih.contiguous() if bih is not None else None), ptr(bhh.contiguous() if bhh is not None else None),
                 ptr(Wp), F_, k, int(pool), ptr(Wnew), ptr(perm), ptr(score), ptr(gates), ptr(xt), stream_of(lib, Wp))
        ctx.save_for_backward(Xc, pc, Wihc, Whhc, Wp, perm, score, gates, xt)
        ctx.pool, ctx.has_bias, ctx.prev_shape, ctx.p_shape = pool, bih is not None, Wprev.shape, (p.shape if pool else None)
        return Wnew

    @staticmethod
    def backward(ctx, dWnew):
        lib = _lib.get_lib()
        Xc, pc, Wihc, Whhc, Wp, perm, score, gates, xt = ctx.saved_tensors
        k, F_ = Wp.shape
        dev = dWnew.device
        dWih, dWhh = torch.empty_like(Wihc), torch.empty_like(Whhc)
        dbih = torch.empty(3 * F_, dtype=F32, device=dev) if ctx.has_bias else None
        dbhh = torch.empty(3 * F_, dtype=F32, device=dev) if ctx.has_bias else None
        dWprev = torch.empty(k, F_, dtype=F32, device=dev)
        dX = torch.zeros_like(Xc) if ctx.pool else None
        dp = torch.empty(F_, dtype=F32, device=dev) if ctx.pool else None
        lib.call("pgt_evolve_weight_bwd_f32", ptr(dWnew.contiguous()), ptr(Xc), Xc.stride(0) if ctx.pool else 0,
                 Xc.size(0) if ctx.pool else 0, ptr(pc), ptr(Wihc), ptr(Whhc), ptr(Wp), ptr(perm), ptr(score), ptr(gates), ptr(xt),
                 F_, k, int(ctx.pool), ptr(dX), dX.stride(0) if ctx.pool else 0, ptr(dp), ptr(dWih), ptr(dWhh), ptr(dbih),
                 ptr(dbhh), ptr(dWprev), stream_of(lib, dWprev))
        return (dX, dp.view(ctx.p_shape) if ctx.pool else None, dWih, dWhh, dbih, dbhh, dWprev.view(ctx.prev_shape), None)
