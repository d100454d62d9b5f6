"""ChebConv and the Chebyshev recurrent cells (GConvGRU / GConvLSTM / GCLSTM): stacks, cell Functions, LSTM gates.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""


import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import (FUSE_GATE_EPILOGUES, _gru_h, _gru_h_bwd, _gru_zr, _gru_zr_bwd, add2d, axpby2d, copy2d, gemm, gemm_gru_h, gemm_gru_zr, gemm_tn_acc, spmm)


# --------------------------------------------------------------------------------------------- Chebyshev convolution

def _cheb_stack_fwd(g, TS, K, N):
    """Tx_0 = TS[0] given; Tx_1 = L Tx_0, Tx_k = 2 L Tx_{k-1} - Tx_{k-2} (PyG ChebConv.forward) on node-major rows."""
    for k in range(1, K):
        src, dst = TS[k - 1].view(N, -1), TS[k].view(N, -1)
        if k == 1:
            spmm(g.fwd, src, dst)
        else:
            spmm(g.fwd, src, dst, T=TS[k - 2].view(N, -1), alpha=2.0, beta=-1.0)


def _cheb_stack_bwd(g, G, K, N):
    """Adjoint of _cheb_stack_fwd on G [K][M][C] in place, highest order first: G_{k-1} += 2 L^T G_k ; G_{k-2} -= G_k;
    on exit G[0] holds d/dTx_0."""
    for k in range(K - 1, 1, -1):
        Gk = G[k].view(N, -1)
        Gp = G[k - 1].view(N, -1)
        spmm(g.bwd, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
        axpby2d(G[k - 2], G[k], -1.0, G[k - 2], 1.0)
    if K > 1:
        G0 = G[0].view(N, -1)
        spmm(g.bwd, G[1].view(N, -1), G0, T=G0, alpha=1.0, beta=1.0)


class ChebGRUCellFunction(torch.autograd.Function):
    """One GConvGRU cell step (gconv_gru.py:119-170) on rows [N, .]: Chebyshev stack of [X, H] -> ONE MFMA GEMM for the
    update and reset gates with sigmoid / H*R in its epilogue -> Chebyshev stack of [X, H*R] -> GEMM with tanh / blend
    in its epilogue (the same gate-fused entry points DCRNN uses: pgt_gemm_gru_zr/h_f32).  Backward: gate kernels,
    feature-gradient GEMMs, the stack adjoint on the transposed operator, one weight-gradient GEMM per stack.
    Wzr [K*C, 2*O] / Wh [K*C, O] stack lins[k].weight^T of the x- and h-convolutions (C = in + out)."""

    @staticmethod
    def forward(ctx, X, H, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, H, "H")
        Xc, Hc = X.contiguous(), H.contiguous()
        M, Fin = Xc.shape
        O = Wh.size(1)
        C = Fin + O
        N = g.N
        if M != N or Hc.shape != (M, O) or Wzr.shape != (K * C, 2 * O) or Wh.shape != (K * C, O):
            raise ValueError("ChebGRUCellFunction: inconsistent operand shapes")
        dev = Xc.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        TSzr = torch.empty(K, M, C, dtype=F32, device=dev)
        TSh = torch.empty(K, M, C, dtype=F32, device=dev)
        copy2d(TSzr[0][:, :Fin], Xc)
        copy2d(TSzr[0][:, Fin:], Hc)
        copy2d(TSh[0][:, :Fin], Xc)
        _cheb_stack_fwd(g, TSzr, K, N)
        ZR = torch.empty(M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(M, O, dtype=F32, device=dev)
        Hout = torch.empty(M, O, dtype=F32, device=dev)
        if FUSE_GATE_EPILOGUES and O % 4 == 0:
            gemm_gru_zr(TSzr, C, M * C, K, C, Wzr_c, 2 * O, 1, bzr, ZR, Hc, TSh[0], Fin)
            _cheb_stack_fwd(g, TSh, K, N)
            gemm_gru_h(TSh, C, M * C, K, C, Wh_c, O, 1, bh, HT, ZR, Hc, Hout, None)
        else:
            gemm(TSzr, C, M * C, K, C, Wzr_c, 2 * O, 1, ZR, 2 * O, 0, 2 * O, bzr, M, 2 * O)
            _gru_zr(ZR, Hc, TSh[0], Fin)
            _cheb_stack_fwd(g, TSh, K, N)
            gemm(TSh, C, M * C, K, C, Wh_c, O, 1, HT, O, 0, O, bh, M, O)
            _gru_h(HT, ZR, Hc, Hout, None)
        ctx.g, ctx.K, ctx.Fin = g, K, Fin
        ctx.has_bias = (bzr is not None, bh is not None)
        ctx.save_for_backward(TSzr, TSh, ZR, HT, Hc, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dHout):
        TSzr, TSh, ZR, HT, Hc, Wzr_c, Wh_c = ctx.saved_tensors
        g, K, Fin = ctx.g, ctx.K, ctx.Fin
        _, M, C = TSzr.shape
        O = HT.size(1)
        N = g.N
        dev = dHout.device
        dHout = dHout.contiguous()
        dPzr = torch.empty(M, 2 * O, dtype=F32, device=dev)
        dPh = torch.empty(M, O, dtype=F32, device=dev)
        dH = torch.zeros(M, O, dtype=F32, device=dev)
        _gru_h_bwd(dHout, ZR, Hc, HT, dPh, dPzr, dH, accumulate=False)          # dH = dH' * Z ; dPh ; the Z half of dPzr
        G = torch.empty(K, M, C, dtype=F32, device=dev)
        gemm(dPh, O, 0, 1, O, Wh_c, 1, O, G, C, M * C, C, None, M, K * C)       # candidate conv: G_k = dPh W_k^T
        _cheb_stack_bwd(g, G, K, N)
        _gru_zr_bwd(G[0], Fin, ZR, Hc, dPzr, dH)                                # the R half of dPzr ; dH += dXHR_H * R
        need_x = ctx.needs_input_grad[0]
        dX = G[0][:, :Fin].clone() if need_x else None
        gemm(dPzr, 2 * O, 0, 1, 2 * O, Wzr_c, 1, 2 * O, G, C, M * C, C, None, M, K * C)
        _cheb_stack_bwd(g, G, K, N)
        add2d(dH, G[0][:, Fin:])
        if need_x:
            add2d(dX, G[0][:, :Fin])
        dWzr = dbzr = dWh = dbh = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dWzr = torch.zeros_like(Wzr_c)
            dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if ctx.has_bias[0] else None
            gemm_tn_acc(TSzr, C, M * C, K, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, M, 2 * O)
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            dWh = torch.zeros_like(Wh_c)
            dbh = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias[1] else None
            gemm_tn_acc(TSh, C, M * C, K, C, dPh, O, dWh, O, dbh, M, O)
        return dX, (dH if ctx.needs_input_grad[1] else None), dWzr, dbzr, dWh, dbh, None, None


class ChebConvFunction(torch.autograd.Function):
    """PyG ChebConv.forward on node-major rows [N*Bt, C] (any number of independent graphs-in-batch folded into the
    feature dimension): Tx_0 = X, Tx_1 = L X, Tx_k = 2 L Tx_{k-1} - Tx_{k-2}; out = sum_k Tx_k @ W_k^T + bias.
    Wst [K*C, O] stacks lins[k].weight^T; g is the scaled-Laplacian SymGraph (pgt_cheb_prep, variant 0)."""

    @staticmethod
    def forward(ctx, X, Wst, bias, g, K, Bt):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        M, C = Xc.shape
        N = g.N
        if M != N * Bt:
            raise ValueError(f"ChebConv: X has {M} rows, expected num_nodes*B = {N * Bt}")
        O = Wst.size(1)
        TS = torch.empty(K, M, C, dtype=F32, device=Xc.device)
        copy2d(TS[0], Xc)
        _cheb_stack_fwd(g, TS, K, N)
        Wc = Wst.contiguous()
        out = torch.empty(M, O, dtype=F32, device=Xc.device)
        gemm(TS, C, M * C, K, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K = g, K
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc)
        return out

    @staticmethod
    def backward(ctx, dOut):
        TS, Wc = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        _, M, C = TS.shape
        O = Wc.size(1)
        N = g.N
        dOut = dOut.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros_like(Wc)
            db = torch.zeros(O, dtype=F32, device=dOut.device) if ctx.has_bias else None
            gemm_tn_acc(TS, C, M * C, K, C, dOut, O, dW, O, db, M, O)
        if ctx.needs_input_grad[0]:
            G = torch.empty(K, M, C, dtype=F32, device=dOut.device)
            gemm(dOut, O, 0, 1, O, Wc, 1, O, G, C, M * C, C, None, M, K * C)   # G_k = dOut @ W_k^T
            _cheb_stack_bwd(g, G, K, N)
            dX = G[0]
        return dX, dW, db, None, None, None


# --------------------------------------------------------------------------------------------- LSTM gates

class LSTMGatesFunction(torch.autograd.Function):
    """(H', C') = peephole-LSTM gates of GConvLSTM / GCLSTM from the gate pre-activations P [M, 4*O] = i | f | c | o
    (pgt_lstm_gates_f32 / pgt_lstm_gates_bwd_f32).  w_ci / w_cf / w_co are [1, O] peephole weights or None."""

    @staticmethod
    def forward(ctx, P, C, w_ci, w_cf, w_co):
        lib = _lib.get_lib()
        check_tensor(lib, P, "P")
        check_tensor(lib, C, "C")
        M, O4 = P.shape
        O = O4 // 4
        gates = P.contiguous().clone()          # activated in place by the kernel; P itself stays untouched
        Cc = C.contiguous()
        ws = [None if w is None else w.contiguous().view(-1) for w in (w_ci, w_cf, w_co)]
        Hn = torch.empty(M, O, dtype=F32, device=P.device)
        Cn = torch.empty(M, O, dtype=F32, device=P.device)
        lib.call("pgt_lstm_gates_f32", ptr(gates), ptr(Cc), O, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(Hn), O, ptr(Cn),
                 O, M, O, stream_of(lib, P))
        ctx.save_for_backward(gates, Cc, Cn, *[w for w in ws if w is not None])
        ctx.has_w = tuple(w is not None for w in ws)
        ctx.w_shapes = tuple(None if w is None else tuple(w.shape) for w in (w_ci, w_cf, w_co))
        return Hn, Cn

    @staticmethod
    def backward(ctx, dH, dCn):
        lib = _lib.get_lib()
        saved = list(ctx.saved_tensors)
        gates, Cc, Cn = saved[:3]
        rest = saved[3:]
        ws = []
        for has in ctx.has_w:
            ws.append(rest.pop(0) if has else None)
        M, O = Cc.shape
        dev = gates.device
        dHc = (dH if dH is not None else torch.zeros_like(Cn)).contiguous()
        dCc = None if dCn is None else dCn.contiguous()
        dP = torch.empty_like(gates)
        dC = torch.empty(M, O, dtype=F32, device=dev)
        dw = torch.zeros(3, O, dtype=F32, device=dev) if any(ctx.has_w) else None
        lib.call("pgt_lstm_gates_bwd_f32", ptr(gates), ptr(Cc), O, ptr(Cn), O, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(dHc),
                 O, ptr(dCc), O, ptr(dP), ptr(dC), O, ptr(dw), M, O, stream_of(lib, gates))
        grads = [dw[i].view(ctx.w_shapes[i]) if ctx.has_w[i] else None for i in range(3)]
        return dP, dC, grads[0], grads[1], grads[2]
