"""ASTGCN / MSTGCN: attention scores, small batched products, time convolution + residual + norm, ChebConvAttention.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""


import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import axpby2d, gemm, gemm_tn_acc, spmm, swap01


# --------------------------------------------------------------------------------------------- dense attention scores

class AttentionScoresFunction(torch.autograd.Function):
    """S = softmax_dim1( V . sigmoid( L R + bias ) ) for a batch of score matrices (ASTGCN's SpatialAttention,
    astgcn.py:226-262, and TemporalAttention, :291-328): L [B, n, m], R [B, m, n], bias [1, n, n] or [n, n], V [n, n]
    -> S [B, n, n].  Three launches forward (fused L R + bias + sigmoid; ONE MFMA GEMM for the whole batch on the
    [i][b][j] layout; softmax over dim 1) instead of five torch ops with [B, n, n] temporaries; hand-written backward
    (softmax and sigmoid adjoints, two GEMMs); the two small products with L and R that close the chain run on
    pgt_bmm_f32."""

    @staticmethod
    def forward(ctx, L, R, bias, V):
        lib = _lib.get_lib()
        for t, nm in ((L, "L"), (R, "R"), (bias, "bias"), (V, "V")):
            check_tensor(lib, t, nm)
        Lc, Rc, Vc = L.contiguous(), R.contiguous(), V.contiguous()
        bc = bias.contiguous()
        B, n, m = Lc.shape
        if Rc.shape != (B, m, n) or Vc.shape != (n, n) or bc.numel() != n * n:
            raise ValueError("AttentionScoresFunction: inconsistent operand shapes")
        dev, st = Lc.device, stream_of(lib, Lc)
        sig = torch.empty(n, B, n, dtype=F32, device=dev)
        lib.call("pgt_att_sigmoid_scores_f32", ptr(Lc), ptr(Rc), ptr(bc), B, n, m, ptr(sig), st)
        C = torch.empty(n, B * n, dtype=F32, device=dev)
        gemm(Vc, n, 0, 1, n, sig, B * n, 1, C, B * n, 0, B * n, None, n, B * n)
        S = torch.empty(B, n, n, dtype=F32, device=dev)
        lib.call("pgt_att_softmax_rows_f32", ptr(C), B, n, ptr(S), st)
        ctx.save_for_backward(Lc, Rc, Vc, sig, S)
        ctx.bias_shape = bias.shape
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.get_lib()
        Lc, Rc, Vc, sig, S = ctx.saved_tensors
        B, n, m = Lc.shape
        dev, st = dS.device, stream_of(lib, dS)
        dS = dS.contiguous()
        dC = torch.empty(n, B * n, dtype=F32, device=dev)
        lib.call("pgt_att_softmax_rows_bwd_f32", ptr(S), ptr(dS), B, n, ptr(dC), st)
        dV = None
        if ctx.needs_input_grad[3]:
            dV = torch.empty(n, n, dtype=F32, device=dev)
            gemm(dC, B * n, 0, 1, B * n, sig, 1, B * n, dV, n, 0, n, None, n, n)          # dV = dC sig^T
        dL = dR = dbias = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            Vt = Vc.t().contiguous()
            dsig = torch.empty(n, B * n, dtype=F32, device=dev)
            gemm(Vt, n, 0, 1, n, dC, B * n, 1, dsig, B * n, 0, B * n, None, n, B * n)    # dsig = V^T dC
            dP = torch.empty(B, n, n, dtype=F32, device=dev)
            db = torch.empty(n, n, dtype=F32, device=dev)
            lib.call("pgt_att_sigmoid_bwd_f32", ptr(sig), ptr(dsig), B, n, ptr(dP), ptr(db), st)
            if ctx.needs_input_grad[2]:
                dbias = db.view(ctx.bias_shape)
            if ctx.needs_input_grad[0]:
                dL = _bmm_raw(dP, Rc.transpose(1, 2), torch.empty(B, n, m, dtype=F32, device=dev))     # [B, n, m]
            if ctx.needs_input_grad[1]:
                dR = _bmm_raw(Lc.transpose(1, 2), dP, torch.empty(B, m, n, dtype=F32, device=dev))     # [B, m, n]
        return dL, dR, dbias, dV


# --------------------------------------------------------------------------------------------- small batched products

def _bmm_raw(A, B, C, accumulate=False):
    """pgt_bmm_f32 on 3-D views [nb, M, K] x [nb, K, N] -> [nb, M, N]; strides are taken as they are (an expanded
    dimension has stride 0, a transposed view swapped strides): no copies."""
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (B, "B"), (C, "C")):
        check_tensor(lib, t, n)
    nb, M, K = A.shape
    N = B.size(2)
    if B.shape != (nb, K, N) or C.shape != (nb, M, N):
        raise ValueError(f"bmm: shapes {tuple(A.shape)} x {tuple(B.shape)} -> {tuple(C.shape)}")
    lib.call("pgt_bmm_f32", ptr(A), *A.stride(), ptr(B), *B.stride(), ptr(C), *C.stride(), nb, M, N, K,
             int(bool(accumulate)), stream_of(lib, C))
    return C


class BmmFunction(torch.autograd.Function):
    """C[b] = A[b] B[b] for small matrices on pgt_bmm_f32 (the embeddings around ASTGCN's attention: astgcn.py:252-256,
    :318-322, :437).  Operands are 3-D views with any strides; gradients come back in the operands' (possibly expanded)
    shapes, so a matrix shared by the batch gets its sum over the batch from `expand`'s own adjoint."""

    @staticmethod
    def forward(ctx, A, B):
        C = torch.empty(A.size(0), A.size(1), B.size(2), dtype=F32, device=A.device)
        _bmm_raw(A, B, C)
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty(A.shape, dtype=F32, device=dC.device)
            _bmm_raw(dC, B.transpose(1, 2), dA)
        if ctx.needs_input_grad[1]:
            dB = torch.empty(B.shape, dtype=F32, device=dC.device)
            _bmm_raw(A.transpose(1, 2), dC, dB)
        return dA, dB


def bmm(A, B):
    """[nb, M, K] x [nb, K, N] (2-D operands are shared by the batch) -> [nb, M, N]."""
    nb = A.size(0) if A.dim() == 3 else B.size(0)
    if A.dim() == 2:
        A = A.unsqueeze(0).expand(nb, -1, -1)
    if B.dim() == 2:
        B = B.unsqueeze(0).expand(nb, -1, -1)
    return BmmFunction.apply(A, B)


class TimeConvResidualNormFunction(torch.autograd.Function):
    """The tail of an ASTGCN block (astgcn.py:463-478): time convolution Conv2d(O -> Ft, (1, 3), stride (1, s), padding
    (0, 1)) of the graph convolution's output + residual Conv2d(Fin -> Ft, (1, 1), stride (1, s)) of the block input ->
    relu -> LayerNorm(Ft), on channels-last rows (b, n, t).

    Xh [B, N, T, O] (already relu'd), Xcl [B, N, T, Fin] -> [B, N, T_out, Ft].  The three taps of the time convolution are
    ONE product on pgt_gemm_f32: the rows live in a buffer with a zero row before and after every (b, n) series, and the
    K-segmented operand takes segment j = the same buffer shifted by j rows (segment stride = one row), so no im2col copy
    exists; the residual convolution accumulates into the same output; relu + LayerNorm pick the strided rows and skip the
    padding rows (pgt_relu_layernorm_f32).  Backward: LayerNorm / relu adjoint, two weight-gradient products on
    pgt_gemm_tn_acc_f32 with the same shifted segments, three accumulating products for the taps' input gradient."""

    @staticmethod
    def forward(ctx, Xh, Xcl, Wt, bt, Wr, br, gamma, beta, stride, eps):
        lib = _lib.get_lib()
        for t, n in ((Xh, "Xh"), (Xcl, "Xcl"), (Wt, "Wt"), (Wr, "Wr"), (gamma, "gamma"), (beta, "beta")):
            check_tensor(lib, t, n)
        B, N, T, O = Xh.shape
        Fin, Ft = Xcl.size(3), Wt.size(0)
        if Xcl.shape != (B, N, T, Fin) or Wt.shape != (Ft, O, 1, 3) or Wr.shape != (Ft, Fin, 1, 1):
            raise ValueError("TimeConvResidualNormFunction: inconsistent operand shapes")
        dev = Xh.device
        BN, Tp, s = B * N, T + 2, int(stride)
        M = BN * Tp
        P = torch.zeros(M + 2, O, dtype=F32, device=dev)          # row (bn, tp) = Xh at t = tp - 1; zero rows at tp = 0, T + 1
        P[:M].view(BN, Tp, O)[:, 1:T + 1].copy_(Xh.reshape(BN, T, O))
        Q = torch.zeros(M, Fin, dtype=F32, device=dev)            # row (bn, tp) = X at t = tp
        Q.view(BN, Tp, Fin)[:, :T].copy_(Xcl.reshape(BN, T, Fin))
        W3 = Wt[:, :, 0, :].permute(2, 1, 0).reshape(3 * O, Ft).contiguous()        # W3[dt * O + o, c] = Wt[c, o, 0, dt]
        WrT = Wr[:, :, 0, 0].t().contiguous()                                        # [Fin, Ft]
        bias = None
        if bt is not None or br is not None:
            bias = (bt if bt is not None else 0) + (br if br is not None else 0)
            bias = bias.contiguous()
        Z = torch.empty(M, Ft, dtype=F32, device=dev)
        gemm(P, O, O, 3, O, W3, Ft, 1, Z, Ft, 0, Ft, bias, M, Ft)                     # segment j = the buffer shifted by j rows
        gemm(Q, Fin, 0, 1, Fin, WrT, Ft, 1, Z, Ft, 0, Ft, None, M, Ft, accumulate=True)
        T_out = (T - 1) // s + 1
        rows = BN * T_out
        Y = torch.empty(rows, Ft, dtype=F32, device=dev)
        stats = torch.empty(rows, 2, dtype=F32, device=dev)
        gc, bc = gamma.contiguous(), beta.contiguous()
        lib.call("pgt_relu_layernorm_f32", ptr(Z), T_out, Tp, s, ptr(gc), ptr(bc), float(eps), rows, Ft, ptr(Y), ptr(stats),
                 stream_of(lib, Z))
        ctx.save_for_backward(P, Q, W3, WrT, Z, stats, gc)
        ctx.dims = (B, N, T, O, Fin, Ft, s, T_out)
        ctx.has_bias = (bt is not None, br is not None)
        return Y.view(B, N, T_out, Ft)

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        P, Q, W3, WrT, Z, stats, gc = ctx.saved_tensors
        B, N, T, O, Fin, Ft, s, T_out = ctx.dims
        dev = dY.device
        BN, Tp = B * N, T + 2
        M, rows = BN * Tp, BN * T_out
        dYc = dY.contiguous().view(rows, Ft)
        dZ = torch.zeros(M, Ft, dtype=F32, device=dev)            # padding / skipped rows carry no gradient
        dgamma, dbeta = torch.zeros(Ft, dtype=F32, device=dev), torch.zeros(Ft, dtype=F32, device=dev)
        lib.call("pgt_relu_layernorm_bwd_f32", ptr(Z), T_out, Tp, s, ptr(gc), ptr(stats), ptr(dYc), rows, Ft, ptr(dZ),
                 ptr(dgamma), ptr(dbeta), stream_of(lib, dZ))
        dW3 = torch.zeros(3 * O, Ft, dtype=F32, device=dev)
        db = torch.zeros(Ft, dtype=F32, device=dev)
        gemm_tn_acc(P, O, O, 3, O, dZ, Ft, dW3, Ft, db, M, Ft)
        dWt = dW3.view(3, O, Ft).permute(2, 1, 0).unsqueeze(2).contiguous()          # [Ft, O, 1, 3]
        dWrT = torch.zeros(Fin, Ft, dtype=F32, device=dev)
        gemm_tn_acc(Q, Fin, 0, 1, Fin, dZ, Ft, dWrT, Ft, None, M, Ft)
        dWr = dWrT.t().contiguous().view(Ft, Fin, 1, 1)
        dXh = dXcl = None
        if ctx.needs_input_grad[0]:
            dP = torch.zeros(M + 2, O, dtype=F32, device=dev)
            for dt in range(3):          # dP[m + dt] += dZ[m] W3[dt]^T : B(k = c, n = o) = W3[dt * O + o, c]
                gemm(dZ, Ft, 0, 1, Ft, W3[dt * O:(dt + 1) * O], 1, Ft, dP[dt:], O, 0, O, None, M, O, accumulate=True)
            dXh = dP[:M].view(BN, Tp, O)[:, 1:T + 1].reshape(B, N, T, O)
        if ctx.needs_input_grad[1]:
            dQ = torch.empty(M, Fin, dtype=F32, device=dev)
            gemm(dZ, Ft, 0, 1, Ft, WrT, 1, Ft, dQ, Fin, 0, Fin, None, M, Fin)          # B(k = c, n = f) = WrT[f, c]
            dXcl = dQ.view(BN, Tp, Fin)[:, :T].reshape(B, N, T, Fin)
        return (dXh, dXcl, dWt, db if ctx.has_bias[0] else None, dWr, db if ctx.has_bias[1] else None, dgamma, dbeta,
                None, None)


# --------------------------------------------------------------------------------------------- attention Chebyshev conv

def spmm_att(csr, S, X3, transpose_s=False):
    """Y[i,b,:] = sum_q val[q] * S[b,i,col[q]] * X3[col[q],b,:]   (S[b,col[q],i] when transpose_s) on [N,B,C]."""
    lib = _lib.get_lib()
    check_tensor(lib, S, "S")
    check_tensor(lib, X3, "X")
    N, B, C = X3.shape
    Y = torch.empty_like(X3)
    lib.call("pgt_spmm_csr_att_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), ptr(S), N, B, C, ptr(X3), ptr(Y),
             int(bool(transpose_s)), stream_of(lib, X3))
    return Y


def sddmm_att(csr, G3, X3, dS):
    """dS[b,i,col[q]] += val[q] * <G3[i,b,:], X3[col[q],b,:]>  (pgt_sddmm_att_f32)."""
    lib = _lib.get_lib()
    N, B, C = X3.shape
    lib.call("pgt_sddmm_att_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), N, B, C, ptr(G3), ptr(X3), ptr(dS),
             stream_of(lib, X3))
    return dS


class ChebConvAttentionFunction(torch.autograd.Function):
    """ChebConvAttention.forward (astgcn.py:112-183) on x [B,N,Fin] — or [B,N,Tt,Fin]: Tt independent calls that share
    the attention (ASTGCNBlock calls the layer once per time step with the same S, astgcn.py:442-452), folded into
    one call — S [B,N,N], W [K,Fin,Fout]:

        T_0 = diag(S[b]) x[b]                         (the reference builds it through a dense eye(N)*S bmm, :159-164)
        T_1 = sum_e norm_e S[b,row_e,col_e] T_0[col_e]  (propagate on the transposed list with Att_norm, :157,169-171)
        T_k = 2 L T_{k-1} - T_{k-2}, k >= 2            (plain norm, :173-177)
        out = sum_k T_k W[k] + bias
    g = SymGraph from pgt_cheb_prep variant 1 (the in-tree __norm__, :82-110)."""

    @staticmethod
    def forward(ctx, x, S, W, bias, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, x, "x")
        check_tensor(lib, S, "spatial_attention")
        squeeze = x.dim() == 3
        if squeeze:
            x = x.unsqueeze(2)
        B, N, Tt, C = x.shape
        if S.shape != (B, N, N) or g.N != N:
            raise ValueError(f"ChebConvAttention: x {tuple(x.shape)}, spatial_attention {tuple(S.shape)}, graph N={g.N}")
        O = W.size(2)
        CC = Tt * C                                                # channels seen by the aggregation
        M = N * B * Tt                                             # rows seen by the feature transform
        Sc = S.contiguous()
        Xnm = swap01(x.contiguous().view(B, N, CC), B, N, CC)      # [N, B, Tt*C]
        d = torch.diagonal(Sc, dim1=1, dim2=2).t().contiguous()    # [N, B]   S[b, i, i]
        TS = torch.empty(K, N, B, CC, dtype=F32, device=x.device)
        torch.mul(Xnm, d.unsqueeze(-1), out=TS[0])
        if K > 1:
            TS[1].copy_(spmm_att(g.fwd, Sc, TS[0]))
        for k in range(2, K):
            spmm(g.fwd, TS[k - 1].view(N, B * CC), TS[k].view(N, B * CC), T=TS[k - 2].view(N, B * CC), alpha=2.0, beta=-1.0)
        Wc = W.contiguous().view(K * C, O)
        out = torch.empty(M, O, dtype=F32, device=x.device)
        gemm(TS, C, M * C, K, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K, ctx.dims, ctx.squeeze = g, K, (B, N, Tt, C, O), squeeze
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc, Sc, Xnm, d)
        res = swap01(out.view(N, B, Tt * O), N, B, Tt * O).view(B, N, Tt, O)
        return res[:, :, 0] if squeeze else res

    @staticmethod
    def backward(ctx, dOut):
        TS, Wc, Sc, Xnm, d = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        B, N, Tt, C, O = ctx.dims
        CC, M = Tt * C, N * B * Tt
        dev = dOut.device
        if ctx.squeeze:
            dOut = dOut.unsqueeze(2)
        dO = swap01(dOut.contiguous().view(B, N, Tt * O), B, N, Tt * O).view(M, O)   # node-major rows (n, b, t)
        dW = torch.zeros_like(Wc)
        db = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias else None
        gemm_tn_acc(TS, C, M * C, K, C, dO, O, dW, O, db, M, O)
        G = torch.empty(K, N, B, CC, dtype=F32, device=dev)
        gemm(dO, O, 0, 1, O, Wc, 1, O, G, C, M * C, C, None, M, K * C)
        for k in range(K - 1, 1, -1):                               # adjoint of T_k = 2 L T_{k-1} - T_{k-2}
            Gk, Gp = G[k].view(N, B * CC), G[k - 1].view(N, B * CC)
            spmm(g.bwd, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
            axpby2d(G[k - 2].view(N * B, CC), G[k].view(N * B, CC), -1.0, G[k - 2].view(N * B, CC), 1.0)
        dS = torch.zeros(B, N, N, dtype=F32, device=dev)
        if K > 1:
            sddmm_att(g.fwd, G[1], TS[0], dS)                       # d/dS of the attention-weighted hop
            G[0].add_(spmm_att(g.bwd, Sc, G[1], transpose_s=True))  # d/dT_0 through the same hop
        # T_0 = d * X : d/dX and the diagonal of d/dS
        dXnm = G[0] * d.unsqueeze(-1)
        dd = (G[0] * Xnm).sum(dim=-1)                               # [N, B]
        torch.diagonal(dS, dim1=1, dim2=2).add_(dd.t())
        dx = swap01(dXnm.contiguous(), N, B, CC).view(B, N, Tt, C)
        if ctx.squeeze:
            dx = dx[:, :, 0]
        return dx, dS, dW.view(K, C, O), db, None, None
