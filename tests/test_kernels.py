"""Per-kernel parity through the C ABI.  `backend` = "emu" (same kernel source on the CPU test double, runs in the
`-m "not gpu"` suite) or "hip" (`-m gpu`, the product library on an MI355X).  fp32 tolerance: 1e-5 (north_star)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import assert_close_with_nonfinite
from oracle import functional as F
from oracle import pyg_restated as P
from pytorch_geometric_temporal_amd import _lib, ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn

ATOL, RTOL = 1e-5, 1e-5


def csr_to_dense(csr, n, live_only=True):
    rp, col, val = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), csr.val.cpu().numpy()
    A = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        for q in range(rp[i], rp[i + 1]):
            A[i, col[q]] += val[q]
    return torch.from_numpy(A)


def dense_from_edges(ei, w, n, to_row=1):
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((ei[to_row], ei[1 - to_row]), w.double(), accumulate=True)
    return A


def random_csr(n, deg_lo, deg_hi, seed, device):
    rng = np.random.default_rng(seed)
    degs = rng.integers(deg_lo, deg_hi + 1, size=n)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    col = rng.integers(0, n, size=rowptr[-1]).astype(np.int32)
    val = rng.standard_normal(rowptr[-1]).astype(np.float32)
    csr = ops.Csr.__new__(ops.Csr)
    csr.n_rows = n
    csr.rowptr = torch.from_numpy(rowptr).to(device)
    csr.col = torch.from_numpy(col).to(device)
    csr.val = torch.from_numpy(val).to(device)
    return csr


def spmm_reference(csr, X, T, alpha, beta):
    rp, col, val = csr.rowptr.cpu().long(), csr.col.cpu().long(), csr.val.cpu()
    rows = torch.repeat_interleave(torch.arange(csr.n_rows), rp[1:] - rp[:-1])
    Xc = X.cpu().double()
    out = torch.zeros_like(Xc).index_add_(0, rows, val.double().view(-1, 1) * Xc[col[: rows.numel()]])
    out = alpha * out
    if T is not None:
        out = out + beta * T.cpu().double()
    return out


# ------------------------------------------------------------------------------------------------ SpMM

@pytest.mark.parametrize("F_", [1, 2, 3, 4, 6, 8, 16, 33, 64, 66, 130, 256, 260, 1000])
def test_spmm_feature_widths(backend, F_):
    n = 150 if backend.name == "emu" else 3000
    csr = random_csr(n, 0, 12, seed=F_, device=backend.device)
    g = torch.Generator().manual_seed(F_)
    X = torch.randn(n, F_, generator=g).to(backend.device)
    T = torch.randn(n, F_, generator=g).to(backend.device)
    Y = torch.full((n, F_), float("nan"), device=backend.device)
    ops.spmm(csr, X, Y)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), ATOL, RTOL, f"F={F_}")
    ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), ATOL, RTOL, f"F={F_} epilogue")
    # in-place accumulate form used by the backward recursion (Y aliases T)
    Tc = T.clone()
    ops.spmm(csr, X, Tc, T=Tc, alpha=2.0, beta=1.0)
    assert_close_with_nonfinite(Tc, spmm_reference(csr, X, T, 2.0, 1.0), ATOL, RTOL, f"F={F_} aliased")


def test_spmm_strided_views_and_heavy_rows(backend):
    n = 130 if backend.name == "emu" else 2000
    csr = random_csr(n, 0, 6, seed=7, device=backend.device)
    # one very heavy tile (> LDS staging capacity) and an empty tail
    rp = csr.rowptr.cpu().numpy().copy()
    extra = 2000
    rng = np.random.default_rng(1)
    col = np.concatenate([csr.col.cpu().numpy()[: rp[5]], rng.integers(0, n, extra).astype(np.int32),
                          csr.col.cpu().numpy()[rp[5]:]])
    val = np.concatenate([csr.val.cpu().numpy()[: rp[5]], rng.standard_normal(extra).astype(np.float32),
                          csr.val.cpu().numpy()[rp[5]:]])
    rp[6:] += extra
    csr.rowptr, csr.col, csr.val = (torch.from_numpy(a).to(backend.device) for a in (rp, col, val))
    big = torch.randn(n, 80, generator=torch.Generator().manual_seed(3)).to(backend.device)
    X = big[:, 8:72]            # 64 columns inside a wider buffer, 32-byte aligned start
    out = torch.zeros(n, 100, device=backend.device)
    Y = out[:, 4:68]
    ops.spmm(csr, X, Y)
    # row 5 sums 2 000 products in fp32 (sequential, like index_add_): its rounding error is ~1e-4, not 1e-5
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 3e-4, 1e-5, "strided")
    assert float(out[:, :4].abs().max()) == 0.0 and float(out[:, 68:].abs().max()) == 0.0
    X1 = big[:, 3:70]           # odd offset -> scalar path
    Y1 = torch.zeros(n, 67, device=backend.device)
    ops.spmm(csr, X1, Y1)
    assert_close_with_nonfinite(Y1, spmm_reference(csr, X1, None, 1.0, 0.0), 5e-5, 1e-5, "unaligned")


def test_spmm_empty_and_errors(backend):
    csr = random_csr(10, 1, 3, seed=0, device=backend.device)
    X = torch.randn(10, 4).to(backend.device)
    with pytest.raises(_lib.PgtError):
        ops.spmm(csr, X, X)                       # Y must not alias X
    empty = random_csr(0, 0, 0, seed=0, device=backend.device)
    ops.spmm(empty, X[:0], X[:0].clone())         # zero rows: no-op
    lib = _lib.get_lib()
    with pytest.raises(_lib.PgtError, match="negative"):
        lib.call("pgt_spmm_csr_f32", None, None, None, -1, None, 0, None, 0, None, 0, 1.0, 0.0, 4, None)


def test_spmm_propagates_inf_and_nan_like_index_add(backend):
    csr = random_csr(20, 2, 4, seed=3, device=backend.device)
    csr.val[0] = float("inf")
    X = torch.randn(20, 8).to(backend.device)
    X[int(csr.col[1]), 2] = 0.0
    Y = torch.empty(20, 8, device=backend.device)
    ops.spmm(csr, X, Y)
    ref = spmm_reference(csr, X, None, 1.0, 0.0)
    assert_close_with_nonfinite(Y, ref, ATOL, RTOL, "inf/nan")


# ------------------------------------------------------------------------------------------------ GEMM

@pytest.mark.parametrize("M,segs,segk,N", [(70, 1, 5, 3), (64, 1, 32, 64), (129, 5, 66, 128), (200, 3, 7, 65),
                                           (33, 2, 1, 2), (257, 1, 100, 40), (131, 5, 64, 64)])
def test_gemm_segmented(backend, M, segs, segk, N):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(segs, M, segk, generator=g)
    W = torch.randn(segs * segk, N, generator=g)
    b = torch.randn(N, generator=g)
    ref = (torch.cat([A[j] for j in range(segs)], dim=1).double() @ W.double()) + b.double()
    Ad, Wd, bd = A.to(backend.device), W.to(backend.device), b.to(backend.device)
    C = torch.full((M, N), float("nan"), device=backend.device)
    ops.gemm(Ad, segk, M * segk, segs, segk, Wd, N, 1, C, N, 0, N, bd, M, N)
    assert_close_with_nonfinite(C, ref, 1e-4, 1e-5, "gemm NN")
    ops.gemm(Ad, segk, M * segk, segs, segk, Wd, N, 1, C, N, 0, N, None, M, N, accumulate=True)
    assert_close_with_nonfinite(C, 2 * ref - b.double(), 2e-4, 1e-5, "gemm accumulate")
    # NT + segmented output: G[j] = dC @ W_j^T
    dC = torch.randn(M, N, generator=g)
    G = torch.full((segs, M, segk), float("nan"), device=backend.device)
    ops.gemm(dC.to(backend.device), N, 0, 1, N, Wd, 1, N, G, segk, M * segk, segk, None, M, segs * segk)
    refG = (dC.double() @ W.double().t()).view(M, segs, segk).permute(1, 0, 2)
    assert_close_with_nonfinite(G, refG, 1e-4, 1e-5, "gemm NT segmented")


@pytest.mark.parametrize("M,segs,segk,N", [(100, 1, 5, 3), (1000, 5, 66, 128), (333, 2, 33, 70), (64, 1, 64, 64)])
def test_gemm_tn_weight_and_bias_gradient(backend, M, segs, segk, N):
    if backend.name == "emu" and M > 500:
        M = 300
    g = torch.Generator().manual_seed(M + N + 1)
    A = torch.randn(segs, M, segk, generator=g)
    G = torch.randn(M, N, generator=g)
    dW0 = torch.randn(segs * segk, N, generator=g)
    db0 = torch.randn(N, generator=g)
    refW = dW0.double() + torch.cat([A[j] for j in range(segs)], dim=1).double().t() @ G.double()
    refb = db0.double() + G.double().sum(0)
    dW, db = dW0.clone().to(backend.device), db0.clone().to(backend.device)
    ops.gemm_tn_acc(A.to(backend.device), segk, M * segk, segs, segk, G.to(backend.device), N, dW, N, db, M, N)
    assert_close_with_nonfinite(dW, refW, 2e-4, 1e-5, "dW")
    assert_close_with_nonfinite(db, refb, 2e-4, 1e-5, "db")


def test_gemm_is_exact_fp32_fma_chain(backend):
    # v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain: integers up to 2^24 must be exact
    A = torch.randint(-8, 9, (64, 64)).float()
    W = torch.randint(-8, 9, (64, 64)).float()
    C = torch.empty(64, 64, device=backend.device)
    ops.gemm(A.to(backend.device), 64, 0, 1, 64, W.to(backend.device), 64, 1, C, 64, 0, 64, None, 64, 64)
    assert torch.equal(C.cpu(), A @ W)
    # asymmetric-B identity check (cdna_hip_programming.md §3: catches a transposed C write)
    I = torch.eye(64)
    B = torch.arange(64 * 64, dtype=torch.float32).view(64, 64)
    ops.gemm(I.to(backend.device), 64, 0, 1, 64, B.to(backend.device), 64, 1, C, 64, 0, 64, None, 64, 64)
    assert torch.equal(C.cpu(), B)


# ------------------------------------------------------------------------------------------------ graph prep

def _graphs():
    yield "sensor_asym", syn.sensor_graph(45, 300, seed=1, symmetric=False)
    yield "sensor_sym", syn.sensor_graph(45, 301, seed=2, symmetric=True)
    e = syn.watts_strogatz_directed(40, 6, 0.5, seed=2)
    yield "ws_directed", (e, (np.random.default_rng(0).random(e.shape[1]) + 0.05).astype(np.float32))


@pytest.mark.parametrize("which", ["sensor_asym", "sensor_sym", "ws_directed"])
def test_dconv_prep_matches_oracle_operators(backend, which):
    ei_np, ew_np = dict(_graphs())[which]
    n = int(ei_np.max()) + 1
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    norm_out, norm_in, rev = F.dconv_norms_scatter(ei, ew, n)
    Po = dense_from_edges(ei, norm_out, n)
    Pi = dense_from_edges(rev, norm_in, n)
    for name, csr, ref in (("fwd_o", g.fwd_o, Po), ("fwd_i", g.fwd_i, Pi), ("bwd_o", g.bwd_o, Po.t()),
                           ("bwd_i", g.bwd_i, Pi.t())):
        assert_close_with_nonfinite(csr_to_dense(csr, n), ref, 1e-6, 1e-6, f"{which}:{name}")
    deg_out = torch.zeros(n).scatter_add_(0, ei[0], ew)
    deg_in = torch.zeros(n).scatter_add_(0, ei[1], ew)
    assert torch.equal(g.deg_out.cpu()[:n], deg_out), "deg_out must be bit-identical to scatter_add_ (edge order)"
    assert torch.equal(g.deg_in.cpu()[:n], deg_in)
    assert g.info.tolist()[:3] == [0, 0, 0]
    # slot order inside a row = reference edge order (so the sequential row sum equals index_add_'s order)
    rp, col = g.fwd_o.rowptr.cpu().numpy(), g.fwd_o.col.cpu().numpy()
    for i in range(n):
        expect = ei_np[0][ei_np[1] == i]
        assert np.array_equal(col[rp[i]:rp[i + 1]], expect)


def test_dconv_prep_flags_duplicates_zero_weights_and_bad_indices(backend):
    ei = torch.tensor([[0, 0, 1, 2], [1, 1, 2, 0]])
    ew = torch.tensor([1.0, 2.0, 0.0, 1.0])
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), 3, validate=False)
    assert g.info.tolist()[:3] == [1, 1, 0]
    with pytest.raises(RuntimeError, match="duplicate"):
        ops.DConvGraph(backend.t(ei), backend.t(ew), 3, strict_dense=True)
    with pytest.raises(IndexError):
        ops.DConvGraph(backend.t(torch.tensor([[0, 5], [1, 0]])), None, 3)
    empty = ops.DConvGraph(backend.t(torch.zeros(2, 0, dtype=torch.long)), None, 4)
    assert empty.fwd_o.rowptr.tolist() == [0] * 5


@pytest.mark.parametrize("improved,loops", [(False, True), (True, True), (False, False)])
def test_gcn_prep_matches_gcn_norm(backend, improved, loops):
    ei_np, ew_np = syn.sensor_graph(40, 260, seed=5, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    n = 40
    g = ops.SymGraph("gcn", backend.t(ei), backend.t(ew), n, improved=improved, add_self_loops=loops)
    ei2, w2 = P.gcn_norm(ei, ew, n, improved, loops)
    A = dense_from_edges(ei2, w2, n)
    assert_close_with_nonfinite(csr_to_dense(g.fwd, n), A, 1e-6, 1e-5, "gcn fwd")
    assert_close_with_nonfinite(csr_to_dense(g.bwd, n), A.t(), 1e-6, 1e-5, "gcn bwd")
    # unweighted input
    g = ops.SymGraph("gcn", backend.t(ei), None, n, improved=False, add_self_loops=loops)
    ei2, w2 = P.gcn_norm(ei, None, n, False, loops, dtype=torch.float32)
    assert_close_with_nonfinite(csr_to_dense(g.fwd, n), dense_from_edges(ei2, w2, n), 1e-6, 1e-5, "gcn unweighted")


@pytest.mark.parametrize("norm,lam", [("sym", None), ("rw", None), (None, None), ("rw", 2.3), (None, 3.1)])
@pytest.mark.parametrize("variant", [0, 1])
def test_cheb_prep_matches_scaled_laplacian(backend, norm, lam, variant):
    ei_np, ew_np = syn.sensor_graph(36, 240, seed=6, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    n = 36
    g = ops.SymGraph("cheb", backend.t(ei), backend.t(ew), n, normalization=norm, lambda_max=lam, variant=variant)
    if variant == 0:
        ei2, w2 = F.cheb_norm(ei, ew, n, norm, None if lam is None else torch.tensor(lam), torch.float32)
        L = dense_from_edges(ei2, w2, n)                 # out[col] += norm * x[row]
    else:
        if lam is None and norm != "sym":
            pytest.skip("ChebConvAttention requires lambda_max for non-sym normalisation (astgcn.py:135-139)")
        ei1, w1 = P.remove_self_loops(ei, ew)
        ei1, w1 = P.get_laplacian(ei1, w1, norm, torch.float32, n)
        w1 = (2.0 * w1) / (2.0 if lam is None else lam)
        w1.masked_fill_(w1 == float("inf"), 0)
        ei1, w1 = P.add_self_loops(ei1, w1, fill_value=-1.0, num_nodes=n)
        L = dense_from_edges(ei1, w1, n, to_row=0)       # transposed list: out[row] += norm * x[col]
    assert_close_with_nonfinite(csr_to_dense(g.fwd, n), L, 2e-6, 1e-5, "cheb fwd")
    assert_close_with_nonfinite(csr_to_dense(g.bwd, n), L.t(), 2e-6, 1e-5, "cheb bwd")


# ------------------------------------------------------------------------------------------------ movers / gates

def test_movers_and_swap(backend):
    x = torch.randn(6, 5, 4)
    y = ops.swap01(backend.t(x), 6, 5, 4)
    assert torch.equal(y.cpu(), x.permute(1, 0, 2).contiguous())
    a, b = torch.randn(7, 9), torch.randn(7, 9)
    dst = torch.zeros(7, 12, device=backend.device)
    ops.copy2d(dst[:, 2:11], backend.t(a))
    assert torch.equal(dst[:, 2:11].cpu(), a) and float(dst[:, :2].abs().sum()) == 0
    ops.add2d(dst[:, 2:11], backend.t(b))
    assert torch.allclose(dst[:, 2:11].cpu(), a + b)
    ops.axpby2d(dst[:, 2:11], backend.t(a), -1.0, dst[:, 2:11], 1.0)
    assert torch.allclose(dst[:, 2:11].cpu(), b, atol=1e-6)


def test_gru_gate_kernels_against_autograd(backend):
    M, O, Fin = 37, 6, 3
    g = torch.Generator().manual_seed(0)
    pre_zr = torch.randn(M, 2 * O, generator=g)
    pre_h = torch.randn(M, O, generator=g)
    H = torch.randn(M, O, generator=g)
    # forward
    zr = pre_zr.clone().to(backend.device)
    xhr = torch.zeros(M, Fin + O, device=backend.device)
    ops._gru_zr(zr, backend.t(H), xhr, Fin)
    Z, R = torch.sigmoid(pre_zr[:, :O]), torch.sigmoid(pre_zr[:, O:])
    assert torch.allclose(zr.cpu(), torch.cat([Z, R], 1), atol=1e-6)
    assert torch.allclose(xhr[:, Fin:].cpu(), H * R, atol=1e-6) and float(xhr[:, :Fin].abs().sum()) == 0
    ht = pre_h.clone().to(backend.device)
    Hn = torch.empty(M, O, device=backend.device)
    ops._gru_h(ht, zr, backend.t(H), Hn)
    assert torch.allclose(Hn.cpu(), Z * H + (1 - Z) * torch.tanh(pre_h), atol=1e-6)
    # backward vs autograd
    pz, ph, Hh = pre_zr.clone().requires_grad_(), pre_h.clone().requires_grad_(), H.clone().requires_grad_()
    Z_, R_ = torch.sigmoid(pz[:, :O]), torch.sigmoid(pz[:, O:])
    HR = Hh * R_
    Hn_ = Z_ * Hh + (1 - Z_) * torch.tanh(ph)
    dHn, dHR = torch.randn(M, O, generator=g), torch.randn(M, O, generator=g)
    (Hn_ * dHn).sum().backward(retain_graph=True)
    gz_h, gph, gH_h = pz.grad.clone(), ph.grad.clone(), Hh.grad.clone()
    pz.grad = None; Hh.grad = None
    (HR * dHR).sum().backward()
    d_pre_h = torch.empty(M, O, device=backend.device)
    d_pre_zr = torch.zeros(M, 2 * O, device=backend.device)
    dH = torch.empty(M, O, device=backend.device)
    half = backend.t(dHn * 0.25)
    dH.copy_(backend.t(dHn * 0.75))      # dH doubles as the second gradient input (aliased, as in the BPTT loop)
    ops._gru_h_bwd(half, zr, backend.t(H), ht, d_pre_h, d_pre_zr, dH, accumulate=False, dHn2=dH)
    assert torch.allclose(d_pre_h.cpu(), gph, atol=1e-5)
    assert torch.allclose(d_pre_zr[:, :O].cpu(), gz_h[:, :O], atol=1e-5)
    assert torch.allclose(dH.cpu(), gH_h, atol=1e-5)
    dxhr = torch.zeros(M, Fin + O)
    dxhr[:, Fin:] = dHR
    ops._gru_zr_bwd(backend.t(dxhr), Fin, zr, backend.t(H), d_pre_zr, dH)
    assert torch.allclose(d_pre_zr[:, O:].cpu(), pz.grad[:, O:], atol=1e-5)
    assert torch.allclose(dH.cpu(), gH_h + Hh.grad, atol=1e-5)


@pytest.mark.parametrize("F_", [4096 + 64, 16 * 256, 17 * 256 + 8, 1030])
def test_spmm_wide_rows_xcd_chunk_mapping(backend, F_):
    # node-major batches: F = B*C is thousands of floats; >= 16 chunks switches on the XCD-slab block mapping
    n = 21 if backend.name == "emu" else 207
    csr = random_csr(n, 0, 9, seed=F_, device=backend.device)
    g = torch.Generator().manual_seed(F_)
    X = torch.randn(n, F_, generator=g).to(backend.device)
    T = torch.randn(n, F_, generator=g).to(backend.device)
    Y = torch.full((n, F_), float("nan"), device=backend.device)
    ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), ATOL, RTOL, f"wide F={F_}")


@pytest.mark.parametrize("M,segs,segk,N", [(200, 5, 66, 128), (300, 5, 66, 64), (150, 1, 128, 330), (260, 3, 7, 65), (210, 5, 64, 128),
                                           (129, 2, 33, 40), (260, 1, 64, 48), (140, 2, 18, 72), (70, 1, 2, 4)])
@pytest.mark.parametrize("pipelined", [1, 2, 0])
def test_gemm_large_tile_variants(backend, M, segs, segk, N, pipelined):
    """The 128-wide tiles (used for M >= 2048) forced onto small problems so the CPU test double covers them too:
    NN with float2 / scalar A loads, NT (k-major B staging), segmented output, and the TN weight-gradient kernel;
    with and without the two-stage pipelined kernel (gemm_db_kernel; it takes the float2-loadable shapes;
    pipelined = 2 selects its two-wavefront 128 x 64 tile instead of the four-wavefront one for N <= 64)."""
    lib = _lib.get_lib()
    lib.tune("gemm_small_tiles", 2)
    lib.tune("gemm_db", 2 if pipelined else 0)
    lib.tune("gemm_db64", 0 if pipelined == 2 else 1)
    try:
        g = torch.Generator().manual_seed(M * 7 + N)
        A = torch.randn(segs, M, segk, generator=g)
        W = torch.randn(segs * segk, N, generator=g)
        b = torch.randn(N, generator=g)
        Acat = torch.cat([A[j] for j in range(segs)], dim=1).double()
        ref = Acat @ W.double() + b.double()
        Ad, Wd, bd = A.to(backend.device), W.to(backend.device), b.to(backend.device)
        C = torch.full((M, N), float("nan"), device=backend.device)
        ops.gemm(Ad, segk, M * segk, segs, segk, Wd, N, 1, C, N, 0, N, bd, M, N)
        assert_close_with_nonfinite(C, ref, 1e-4, 1e-5, "big NN")
        ops.gemm(Ad, segk, M * segk, segs, segk, Wd, N, 1, C, N, 0, N, None, M, N, accumulate=True)
        assert_close_with_nonfinite(C, 2 * ref - b.double(), 2e-4, 1e-5, "big NN accumulate")
        dC = torch.randn(M, N, generator=g)
        G = torch.full((segs, M, segk), float("nan"), device=backend.device)
        ops.gemm(dC.to(backend.device), N, 0, 1, N, Wd, 1, N, G, segk, M * segk, segk, None, M, segs * segk)
        refG = (dC.double() @ W.double().t()).view(M, segs, segk).permute(1, 0, 2)
        assert_close_with_nonfinite(G, refG, 1e-4, 1e-5, "big NT")
        dW = torch.zeros(segs * segk, N, device=backend.device)
        db = torch.zeros(N, device=backend.device)
        ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, dC.to(backend.device), N, dW, N, db, M, N)
        assert_close_with_nonfinite(dW, Acat.t() @ dC.double(), 2e-4, 1e-5, "big TN")
        assert_close_with_nonfinite(db, dC.double().sum(0), 2e-4, 1e-5, "big TN bias")
    finally:
        lib.tune("gemm_small_tiles", 0)
        lib.tune("gemm_db", 1)
        lib.tune("gemm_db64", 1)


def test_spmm_tuning_variants_agree(backend):
    lib = _lib.get_lib()
    n = 300 if backend.name == "emu" else 5000
    csr = random_csr(n, 0, 20, seed=11, device=backend.device)
    X = torch.randn(n, 64).to(backend.device)
    ref = spmm_reference(csr, X, None, 1.0, 0.0)
    try:
        for rows in (32, 64, 128):
            for unroll in (4, 8):
                for xcd in (0, 1):
                    lib.tune("spmm_tile_rows", rows); lib.tune("spmm_unroll", unroll); lib.tune("spmm_tile_xcd", xcd)
                    Y = torch.full((n, 64), float("nan"), device=backend.device)
                    ops.spmm(csr, X, Y)
                    assert_close_with_nonfinite(Y, ref, ATOL, RTOL, f"rows={rows} unroll={unroll} xcd={xcd}")
        with pytest.raises(_lib.PgtError, match="unknown key"):
            lib.tune("no_such_knob", 1)
    finally:
        lib.tune("spmm_tile_rows", 32); lib.tune("spmm_unroll", 8); lib.tune("spmm_tile_xcd", 1)


# ------------------------------------------------------------------------------------------------ ELLW (LDS-window) SpMM

def banded_csr(n, deg_lo, deg_hi, window, seed, device, far_frac=0.0, heavy_row=None, source_scaled=False):
    """rows with sources inside +-window (wrapping modulo n at both ends, like synthetic.local_graph) plus a fraction
    of far-away sources; optionally one row with hundreds of slots; `source_scaled`: val[q] = scale[col[q]] (DConv's P_o)."""
    rng = np.random.default_rng(seed)
    degs = rng.integers(deg_lo, deg_hi + 1, size=n)
    if heavy_row is not None:
        degs[heavy_row] = 300
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    rows = np.repeat(np.arange(n), degs)
    col = (rows + rng.integers(-window, window + 1, size=rows.size)) % n
    far = rng.random(rows.size) < far_frac
    col[far] = rng.integers(0, n, size=int(far.sum()))
    val = rng.standard_normal(rows.size).astype(np.float32)
    if source_scaled:
        val = (0.25 + rng.random(n)).astype(np.float32)[col]
    csr = ops.Csr.__new__(ops.Csr)
    csr.n_rows, csr.halo, csr.max_len, csr.nnz, csr.ellw, csr.long_rows = n, 0, -1, -1, None, None
    csr.rowptr = torch.from_numpy(rowptr).to(device)
    csr.col = torch.from_numpy(col.astype(np.int32)).to(device)
    csr.val = torch.from_numpy(val).to(device)
    return csr


def source_scaled_reference(csr, X):
    """fp32, the roundings of the reference's propagate: message = norm * x_j (rounded), then a sequential add per
    destination in slot order (scatter-add on the CPU) — what the source-scale mode of the ELLW kernel reproduces."""
    rp = csr.rowptr.cpu().long()
    n = csr.n_rows
    lens = rp[1:] - rp[:-1]
    col, val, Xc = csr.col.cpu().long(), csr.val.cpu(), X.cpu()
    acc = torch.zeros(n, X.size(1))
    for j in range(int(lens.max()) if n else 0):
        live = torch.nonzero(lens > j).flatten()
        q = rp[live] + j
        acc[live] = acc[live] + val[q, None] * Xc[col[q]]
    return acc


@pytest.mark.parametrize("halo,window,n,cap,cus", [(32, 32, 333, 0, 0), (32, 40, 1000, 48, 3), (96, 96, 700, 0, 0),
                                                   (32, 5, 17, 0, 0), (32, 30, 900, 100, 2), (96, 20, 64, 16, 0)])
def test_spmm_ellw_matches_csr_kernels_and_reference(backend, halo, window, n, cap, cus):
    """pgt_spmm_ellw_f32 (ELLW layout, X window in LDS) with per-slot coefficients: the same fmaf chain in slot order as
    the CSR row tiles, bit for bit, with sources outside the window (wrap-around, 5 % long-range edges), empty rows,
    tile heights forced small (`spmm_ellw_rows`) so that several ragged tiles occur, the epilogue, an aliased T."""
    lib = _lib.get_lib()
    if backend.name == "hip":
        n *= 37
    elif n > 600:
        n = 600                     # the CPU test double is a fiber emulator: keep its share of the suite to seconds
    lib.tune("spmm_ellw_rows", cap)
    lib.tune("spmm_ellw_cus", cus)
    try:
        csr = banded_csr(n, 0, 20, window, seed=n, device=backend.device, far_frac=0.05)
        g = torch.Generator().manual_seed(n)
        X = torch.randn(n, 64, generator=g).to(backend.device)
        T = torch.randn(n, 64, generator=g).to(backend.device)
        e = ops._force_ellw(csr, halo)
        assert e is not None and e.vals is not None and e.scale is None and e.width == 24
        assert e.n_tiles * e.tile_rows >= n and (cap == 0 or e.tile_rows <= cap)
        assert e.far > 0 or n < 200     # wrap-around / long-range sources outside the window
        ref = spmm_reference(csr, X, None, 1.0, 0.0)
        Yb = torch.full((n, 64), float("nan"), device=backend.device)
        ops.spmm(csr, X, Yb)
        assert_close_with_nonfinite(Yb, ref, 5e-5, 1e-5, "ellw")
        Yp = torch.empty_like(Yb)
        ops.spmm(csr, X, Yp, ellw=False)
        assert torch.equal(Yb, Yp)          # same slot-order fma chain in both kernels: bit-identical
        ops.spmm(csr, X, Yb, T=T, alpha=2.0, beta=-1.0)
        ops.spmm(csr, X, Yp, T=T, alpha=2.0, beta=-1.0, ellw=False)
        assert torch.equal(Yb, Yp)
        assert_close_with_nonfinite(Yb, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "ellw epilogue")
        Tc = T.clone()
        ops.spmm(csr, X, Tc, T=Tc, alpha=2.0, beta=1.0)
        assert_close_with_nonfinite(Tc, spmm_reference(csr, X, T, 2.0, 1.0), 5e-5, 1e-5, "ellw aliased")
    finally:
        lib.tune("spmm_ellw_rows", 0)
        lib.tune("spmm_ellw_cus", 0)


@pytest.mark.parametrize("n,cap,deg_hi", [(400, 40, 8), (523, 0, 30), (90, 12, 5)])
def test_spmm_ellw_source_scaled_operator_drops_the_coefficient_stream(backend, n, cap, deg_hi):
    """val[q] == scale[col[q]] for every slot (DConv's P_o, dcrnn.py:70-73): the layout keeps the per-source table only and
    the kernel reproduces the reference's roundings exactly (rounded product, sequential rounded adds)."""
    lib = _lib.get_lib()
    if backend.name == "hip":
        n *= 53
    lib.tune("spmm_ellw_rows", cap)
    try:
        csr = banded_csr(n, 0, deg_hi, 30, seed=n + 1, device=backend.device, far_frac=0.03, source_scaled=True)
        e = ops._force_ellw(csr, 32)
        assert e is not None and e.scale is not None and e.vals is None
        X = torch.randn(n, 64, generator=torch.Generator().manual_seed(n)).to(backend.device)
        Y = torch.full((n, 64), float("nan"), device=backend.device)
        ops.spmm(csr, X, Y)
        assert torch.equal(Y.cpu(), source_scaled_reference(csr, X))
        assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "vs fp64")
        T = torch.randn(n, 64).to(backend.device)
        ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0)
        assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "epilogue")
    finally:
        lib.tune("spmm_ellw_rows", 0)


@pytest.mark.parametrize("source_scaled", [True, False])
def test_spmm_ellw_out_of_window_sources_get_lds_rows(backend, source_scaled):
    """Sources outside a tile's window (long-range edges): up to far_rows of them per tile are prefetched into LDS rows
    behind the zero row (their slots point there), the rest keep 0xFFFF and come through the CSR — same sums either way."""
    lib = _lib.get_lib()
    n = 500 if backend.name == "emu" else 40_000
    lib.tune("spmm_ellw_rows", 64)
    try:
        for far_frac, all_fit in ((0.01, True), (0.5, False)):
            csr = banded_csr(n, 2, 8, 20, seed=17, device=backend.device, far_frac=far_frac, source_scaled=source_scaled)
            e = ops._force_ellw(csr, 32)
            assert e is not None and (e.scale is not None) == source_scaled and e.width == 8 and e.config == 1
            assert e.far_rows == (128 if source_scaled else 32) and e.far > 0 and e.far_col is not None
            table = e.far_col.view(e.n_tiles, e.far_rows).cpu()
            used = (table >= 0).sum().item()
            # overflow (slots left to the CSR path) only once a tile's table is full
            assert e.far_csr == 0 if all_fit else (0 <= e.far_csr < e.far and (e.far_csr == 0 or int((table >= 0).sum(1).max()) == e.far_rows))
            if not (source_scaled and backend.name == "emu"):
                assert all_fit or e.far_csr > 0
            assert 0 < used <= e.far - e.far_csr and int(table.max()) < n              # a hash set of DISTINCT sources per tile
            for row in table[:50]:
                live = row[row >= 0]
                assert live.numel() == live.unique().numel()
            X = torch.randn(n, 64, generator=torch.Generator().manual_seed(1)).to(backend.device)
            T = torch.randn(n, 64, generator=torch.Generator().manual_seed(2)).to(backend.device)
            Y = torch.full((n, 64), float("nan"), device=backend.device)
            ops.spmm(csr, X, Y, T=T, alpha=0.5, beta=2.0)
            if source_scaled:
                assert torch.equal(Y.cpu(), (0.5 * source_scaled_reference(csr, X) + 2.0 * T.cpu()))
            else:
                Yp = torch.empty_like(Y)
                ops.spmm(csr, X, Yp, T=T, alpha=0.5, beta=2.0, ellw=False)
                assert torch.equal(Y, Yp)
            assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 0.5, 2.0), 5e-5, 1e-5, "far rows")
    finally:
        lib.tune("spmm_ellw_rows", 0)


def test_spmm_ellw_compact_tiles_of_a_mesh_on_a_space_filling_curve(backend):
    """A 2-D mesh numbered along a Hilbert curve is not a band (under 95 % of the slots within +-96 rows), but its tiles are
    compact patches whose ring (~85 distinct outside rows per 392-row tile, named by ~230 slots) fits the per-tile table:
    ops.ellw_of keeps the layout when no slot is left to the CSR path.  A uniform-random graph has no numbering with
    compact tiles: rejected (also by the renumbering attempt), it runs the CSR row tiles.  Source-scale mode: bit for bit
    the reference's roundings."""
    side = 64 if backend.name == "emu" else 200
    n = side * side
    X = torch.randn(n, 64, generator=torch.Generator().manual_seed(3)).to(backend.device)
    # (a row-major or shuffled mesh is laid out in a numbering of the library's own: test_spmm_ellw_renumbered_layout_*)
    for order, takes in (("hilbert", True), ("uniform", False)):
        ei, ew = syn.uniform_graph(n, 8, seed=1) if order == "uniform" else syn.grid2d_graph(side, order, seed=1)
        G = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
        csr = G.fwd_o
        assert csr.halo == 0, order
        Y = torch.full((n, 64), float("nan"), device=backend.device)
        ops.spmm(csr, X, Y)
        e = csr.ellw
        if takes:
            assert e and e.scale is not None and e.far_csr == 0 and e.far > 0
            assert torch.equal(Y.cpu(), source_scaled_reference(csr, X))
        else:
            assert e is False
        Yc = torch.empty_like(Y)
        ops.spmm(csr, X, Yc, ellw=False)
        assert_close_with_nonfinite(Y, Yc, 1e-5, 1e-5, order)
    # the same mesh with a junction of 60 more in-edges: the compact-tile layout leaves that row out like a band's would
    ei, ew = syn.grid2d_graph(side, "hilbert", seed=1)
    rng = np.random.default_rng(4)
    src = rng.choice(n, 60, replace=False)
    e2 = np.concatenate([ei, np.stack([src, np.full(60, 1234)])], axis=1)
    w2 = np.concatenate([ew, (0.5 + rng.random(60)).astype(np.float32)])
    key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    G = ops.DConvGraph(backend.t(e2[:, key]), backend.t(w2[key]), n)
    csr = G.fwd_o
    assert csr.halo == 0 and csr.left_rows is not None and csr.left_rows.tolist() == [1234] and csr.long_rows is None
    Y = torch.full((n, 64), float("nan"), device=backend.device)
    ops.spmm(csr, X, Y)
    e = csr.ellw
    assert e and e.left_out == 1 and e.order is None and e.far_csr <= ops.ELLW_COMPACT_MAX_CSR_FRACTION * csr.nnz
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "compact tiles + a junction")
    ordinary = torch.ones(n, dtype=torch.bool)
    ordinary[1234] = False
    assert torch.equal(Y.cpu()[ordinary], source_scaled_reference(csr, X)[ordinary])


def test_spmm_ellw_strided_nonfinite_and_fallback_shapes(backend):
    n = 300 if backend.name == "emu" else 9000
    csr = banded_csr(n, 1, 9, 30, seed=5, device=backend.device)
    assert ops._force_ellw(csr, 32) is not None
    big = torch.randn(n, 80).to(backend.device)
    big[7, 10] = float("inf")
    big[n - 1, 12] = float("nan")
    X = big[:, 8:72]
    out = torch.zeros(n, 100, device=backend.device)
    Y = out[:, 4:68]
    ops.spmm(csr, X, Y)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "ellw strided")
    assert float(out[:, :4].abs().max()) == 0.0 and float(out[:, 68:].abs().max()) == 0.0
    # multiples of 64 run one column chunk per 64 floats; other widths and operands that are not 16-byte aligned run
    # the CSR kernels under the same entry point
    for F_ in (32, 66, 128, 320):
        Xf = torch.randn(n, F_).to(backend.device)
        Yf = torch.empty_like(Xf)
        ops.spmm(csr, Xf, Yf)
        assert_close_with_nonfinite(Yf, spmm_reference(csr, Xf, None, 1.0, 0.0), 5e-5, 1e-5, f"fallback F={F_}")
    Xo = torch.randn(n, 65).to(backend.device)[:, 1:]
    Yo = torch.empty(n, 64, device=backend.device)
    ops.spmm(csr, Xo, Yo)
    assert_close_with_nonfinite(Yo, spmm_reference(csr, Xo, None, 1.0, 0.0), 5e-5, 1e-5, "unaligned X")
    # rows longer than 32 slots: the layout does not apply, the operator stays on the CSR kernels
    heavy = banded_csr(n, 0, 6, 30, seed=6, device=backend.device, heavy_row=40)
    assert ops._force_ellw(heavy, 32) is None and heavy.ellw is None
    Xh, Yh = torch.randn(n, 64).to(backend.device), torch.empty(n, 64, device=backend.device)
    ops.spmm(heavy, Xh, Yh, ellw=True)
    assert_close_with_nonfinite(Yh, spmm_reference(heavy, Xh, None, 1.0, 0.0), 5e-5, 1e-5, "heavy row")


def test_spmm_ellw_wide_rows_run_column_chunks(backend):
    """F = 64 k (node-major batches [N][B C]): k column chunks per tile, bit-identical to the CSR kernels in per-slot mode,
    with the epilogue and strided operands."""
    n = 260 if backend.name == "emu" else 30_000
    csr = banded_csr(n, 0, 12, 28, seed=3, device=backend.device, far_frac=0.02)
    assert ops._force_ellw(csr, 32) is not None
    big = torch.randn(n, 200).to(backend.device)
    X, T = big[:, 4:196], torch.randn(n, 192).to(backend.device)
    Ya, Yb = torch.full((n, 192), float("nan"), device=backend.device), torch.empty(n, 192, device=backend.device)
    ops.spmm(csr, X, Ya, T=T, alpha=2.0, beta=-1.0)
    ops.spmm(csr, X, Yb, T=T, alpha=2.0, beta=-1.0, ellw=False)
    assert torch.equal(Ya, Yb)
    assert_close_with_nonfinite(Ya, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "wide rows")


def test_ellw_plan_fills_whole_rounds_of_the_cus(backend):
    """pgt_ellw_plan: N = 200 000, halo 32 on 256 CUs -> 511 tiles of 392 rows (two rounds); the caps are honoured."""
    import ctypes
    lib = _lib.get_lib()

    def plan(n, halo, max_len, source_scaled=1):
        tr, w, cfg, nt, fr = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
        lib.call("pgt_ellw_plan", n, halo, max_len, source_scaled, ctypes.byref(tr), ctypes.byref(w), ctypes.byref(cfg),
                 ctypes.byref(nt), ctypes.byref(fr))
        assert cfg.value == (1 if (w.value == 8 or halo > 40 or (w.value == 16 and source_scaled)) else 2)
        # out-of-window table: what the LDS budget of the launch shape leaves next to window, slots (and coefficients)
        assert fr.value == {(1, 1): 128, (1, 0): 32, (2, 1): 48, (2, 0): 12}[(cfg.value, source_scaled)]
        return tr.value, w.value, nt.value

    lib.tune("spmm_ellw_cus", 256)
    try:
        assert plan(200_000, 32, 8) == (392, 8, 511)
        assert plan(200_000, 32, 8, 0) == (392, 8, 511)
        assert plan(200_000, 96, 8) == (264, 8, 758)          # window of 456 rows: 264 + 2 * 96
        assert plan(200_000, 32, 17) == (100, 24, 2000)       # two workgroups per CU, 176 * 16 staged slots: <= 116 rows of 24
        assert plan(200_000, 32, 16) == (392, 16, 511)        # source-scale mode: one workgroup per CU up to 16 slots (26.4 vs 27.3 us)
        assert plan(200_000, 32, 16, 0) == (132, 16, 1516)    # per-slot mode: 176 window-limited rows -> three rounds of 512
        assert plan(200_000, 96, 16) == (264, 16, 758)        # wide halo: one workgroup per CU
        tr, w, nt = plan(50_000, 32, 8)
        assert (tr, w, nt) == (196, 8, 256)                   # one round
        tr, w, nt = plan(1_000_000, 32, 3)
        assert w == 8 and tr <= 392 and nt == -(-1_000_000 // tr) and nt <= 11 * 256
        with pytest.raises(_lib.PgtError, match="exceed"):
            plan(1000, 32, 33)
        with pytest.raises(_lib.PgtError, match="halo"):
            plan(1000, 300, 8)
    finally:
        lib.tune("spmm_ellw_cus", 0)


def test_locality_hint_is_measured_per_operator(backend):
    n = 5000
    ei, ew = syn.local_graph(n, 4, window=64, seed=0)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    assert g.fwd_o.halo == 32 and g.bwd_o.halo == 32 and g.fwd_i.halo == 32
    ei, ew = syn.uniform_graph(n, 4, seed=0)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    assert g.fwd_o.halo == 0
    ei, ew = syn.local_graph(n, 4, window=150, seed=0)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    assert g.fwd_o.halo == 96
    small = ops.DConvGraph(*[backend.t(a) for a in syn.sensor_graph(207, 1515, seed=0)], 207)
    assert small.fwd_o.halo == 0          # tiny graphs keep the plain schedule


# ------------------------------------------------------------------------------------------------ LDS-resident diffusion stack

@pytest.mark.parametrize("n,C,K,B", [(20, 36, 2, 3), (40, 6, 3, 5), (33, 7, 3, 2), (207, 66, 3, 2), (25, 10, 3, 4)])
@pytest.mark.parametrize("pairs", [2, 1])
def test_slab_stack_equals_per_hop_launches(backend, n, C, K, B, pairs, request):
    """pgt_dconv_stack_slab(_bwd)_f32 (batch-major rows, one launch) against the per-hop pgt_spmm_csr_f32 path
    (node-major rows), forward and adjoint, folded and unfolded; with one and with two column pairs per lane (the
    two-pair kernels take even C >= 8: C = 66 and C = 10 end a row on a half-filled lane)."""
    lib = _lib.get_lib()
    lib.tune("slab_pairs", pairs)
    request.addfinalizer(lambda: lib.tune("slab_pairs", 2))
    if backend.name == "emu" and n > 100:
        B = 1
    ei, ew = syn.sensor_graph(n, 6 * n, seed=n, symmetric=False)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    assert ops.slab_fits(g, C, K)
    S = 2 * K - 1
    gen = torch.Generator().manual_seed(C)
    X = torch.randn(B, n, C, generator=gen)
    # node-major reference: rows m = n*B + b
    TSn = torch.zeros(S, 1, n * B, C)
    TSn[0, 0] = X.permute(1, 0, 2).reshape(n * B, C)
    TSn = backend.t(TSn)
    ops._stack_fwd(g, TSn, 0, K, n)
    # batch-major slab
    TSb = torch.zeros(S, 1, B * n, C)
    TSb[0, 0] = X.reshape(B * n, C)
    TSb = backend.t(TSb)
    ops._slab_fwd(g, TSb[0, 0], B * n * C, B, C, K)
    ref = TSn.cpu().view(S, n, B, C).permute(0, 2, 1, 3).reshape(S, B * n, C)
    assert_close_with_nonfinite(TSb.cpu().view(S, B * n, C), ref, 1e-6, 1e-6, "forward stack")
    for folded in (False, True):
        Gsrc = torch.randn(S, B, n, C, generator=gen)
        Gn = backend.t(Gsrc.permute(0, 2, 1, 3).reshape(S, n * B, C).contiguous())
        ops._stack_bwd(g, Gn, K, n, folded)
        Gb = backend.t(Gsrc.reshape(S, B * n, C).contiguous())
        ops._slab_bwd(g, Gb[0], B * n * C, B, C, K, folded)
        refg = Gn[0].cpu().view(n, B, C).permute(1, 0, 2).reshape(B * n, C)
        assert_close_with_nonfinite(Gb[0], refg, 2e-6, 2e-6, f"adjoint folded={folded}")
        assert torch.equal(Gb[1:].cpu(), Gsrc.reshape(S, B * n, C)[1:])      # other segments untouched


def test_slab_stack_rejects_what_does_not_fit_lds(backend):
    lib = _lib.get_lib()
    ei, ew = syn.sensor_graph(600, 3000, seed=0, symmetric=False)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), 600)
    assert not ops.slab_fits(g, 4, 4)         # K > 3
    assert ops.slab_fits(g, 4, 3)
    assert ops.slab_fits(g, 66, 3)            # 2 x 600 x 66 x 4 B > 160 KiB as a whole, but three column windows fit
    lib.tune("slab_split", 0)
    try:
        assert not ops.slab_fits(g, 66, 3)    # ... which only the column-split kernels can use
        TS = backend.t(torch.zeros(5, 600, 66))
        with pytest.raises(_lib.PgtError, match="not supported"):
            ops._slab_fwd(g, TS[0], 600 * 66, 1, 66, 3)
    finally:
        lib.tune("slab_split", 1)
    ei, ew = syn.sensor_graph(2500, 12000, seed=0, symmetric=False)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), 2500)
    assert not ops.slab_fits(g, 66, 3)        # the operators alone (2 x 12 000 slots x 8 B) exceed the LDS
    with pytest.raises(_lib.PgtError, match="not supported"):
        ops._slab_fwd(g, backend.t(torch.zeros(5, 2500, 66))[0], 2500 * 66, 1, 66, 3)


@pytest.mark.parametrize("n,C,K,B", [(207, 66, 3, 11), (207, 64, 3, 9), (325, 66, 3, 3), (40, 24, 2, 19), (60, 10, 3, 17),
                                     (600, 66, 3, 2)])
def test_slab_stack_column_split_is_bit_identical(backend, n, C, K, B):
    """The column-split kernels (work item = sample x column window; two or three workgroups per CU) against the
    whole-sample kernels and the per-hop path: the recursion is column-independent and every element keeps its fmaf chain,
    so the results are bit-identical — for every window count, ragged batches (B not a multiple of 8: the item order packs
    eight samples per group), forced workgroup sizes, forward and both adjoint forms.  N = 325 and 600 at C = 66 only fit
    column by column.  C = 64 / 66 also covers the quad-layout whole-sample kernels against the pair-layout ones."""
    lib = _lib.get_lib()
    if backend.name == "emu":
        B = min(B, 3) if n > 100 else B
    ei, ew = syn.sensor_graph(n, 6 * n, seed=n, symmetric=False)
    g = ops.DConvGraph(backend.t(ei), backend.t(ew), n)
    S = 2 * K - 1
    gen = torch.Generator().manual_seed(C + n)
    X = torch.randn(B, n, C, generator=gen)
    Gsrc = torch.randn(S, B * n, C, generator=gen)

    def run(split, threads=0, wpc=0, quad=1):
        lib.tune("slab_split", split)
        lib.tune("slab_threads", threads)
        lib.tune("slab_wpc", wpc)
        lib.tune("slab_quad", quad)
        try:
            if not ops.slab_fits(g, C, K):
                return None
            TS = torch.zeros(S, 1, B * n, C)
            TS[0, 0] = X.reshape(B * n, C)
            TS = backend.t(TS)
            ops._slab_fwd(g, TS[0, 0], B * n * C, B, C, K)
            outs = [TS.cpu()]
            for folded in (False, True):
                Gb = backend.t(Gsrc)
                ops._slab_bwd(g, Gb[0], B * n * C, B, C, K, folded)
                outs.append(Gb.cpu())
            return outs
        finally:
            lib.tune("slab_split", 1)
            lib.tune("slab_threads", 0)
            lib.tune("slab_wpc", 0)
            lib.tune("slab_quad", 1)

    # reference: the per-hop launches on node-major rows
    TSn = torch.zeros(S, 1, n * B, C)
    TSn[0, 0] = X.permute(1, 0, 2).reshape(n * B, C)
    TSn = backend.t(TSn)
    ops._stack_fwd(g, TSn, 0, K, n)
    ref_fwd = TSn.cpu().view(S, n, B, C).permute(0, 2, 1, 3).reshape(S, 1, B * n, C)
    whole = run(0)
    base = run(1)
    assert base is not None
    assert torch.equal(base[0], ref_fwd)
    if whole is not None:
        for a, b in zip(whole, base):
            assert torch.equal(a, b)
        # whole-sample kernels in the pair layout against the quad layout (C = 64 / 66: one conflict-free ds_read_b128 per slot)
        for a, b in zip(run(0, quad=0), whole):
            assert torch.equal(a, b)
    for split, threads, wpc in ((2, 0, 0), (3, 0, 0), (4, 512, 0), (5, 0, 1), (8, 0, 2), (2, 1024, 1), (3, 640, 0)):
        got = run(split, threads, wpc)
        if got is None:
            continue
        for a, b in zip(got, base):
            assert torch.equal(a, b), (split, threads, wpc)


def test_batched_dcrnn_node_major_fallback_for_larger_graphs(backend):
    """N too large for the LDS-resident stack: BatchedDCRNN keeps node-major rows and one launch per hop."""
    from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN
    n = 1300 if backend.name == "emu" else 2600     # (8 column windows of 2 x n x 8 floats + both operators > 160 KB)
    ei_np, ew_np = syn.sensor_graph(n, 4 * n, seed=5, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    torch.manual_seed(0)
    m = BatchedDCRNN(2, 64, K=2)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = ops.dconv_graph(backend.t(ei), backend.t(ew), n)
    assert not ops.slab_fits(g, 66, 2)
    X = torch.randn(1, 2, n, 2)
    with torch.no_grad():
        out = m.to(backend.device)(backend.t(X), backend.t(ei), backend.t(ew))
    ref = F.batched_dcrnn(X, ei, ew, params)
    assert_close_with_nonfinite(out, ref, 1e-5, 1e-5, "node-major fallback")


@pytest.mark.parametrize("M,segs,segk,N", [(300, 5, 66, 128), (257, 5, 66, 64), (130, 3, 33, 100), (90, 1, 384, 20),
                                           (64, 2, 5, 192)])
def test_gemm_tn_whole_k_schedule(backend, M, segs, segk, N):
    """The whole-K weight-gradient kernel (one workgroup spans every k-tile; used for M >= 16384) forced onto small
    problems so the CPU test double covers it; on the GPU leg also at its natural size."""
    lib = _lib.get_lib()
    if backend.name == "hip":
        M = M * 97
    lib.tune("gemm_tn_fullk", 2)
    lib.tune("gemm_tn_pipe", 0)
    try:
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(segs, M, segk, generator=g)
        G = torch.randn(M, N, generator=g)
        dW0 = torch.randn(segs * segk, N, generator=g)
        db0 = torch.randn(N, generator=g)
        refW = dW0.double() + torch.cat([A[j] for j in range(segs)], dim=1).double().t() @ G.double()
        refb = db0.double() + G.double().sum(0)
        dW, db = dW0.clone().to(backend.device), db0.clone().to(backend.device)
        ops.gemm_tn_acc(A.to(backend.device), segk, M * segk, segs, segk, G.to(backend.device), N, dW, N, db, M, N)
        tol = 2e-4 if backend.name == "emu" else 2e-3        # sums of up to ~29 000 products on the GPU leg
        assert_close_with_nonfinite(dW, refW, tol, 1e-5, "dW")
        assert_close_with_nonfinite(db, refb, tol, 1e-5, "db")
        lib.tune("gemm_tn_fullk", 0)
        dW2, db2 = dW0.clone().to(backend.device), db0.clone().to(backend.device)
        ops.gemm_tn_acc(A.to(backend.device), segk, M * segk, segs, segk, G.to(backend.device), N, dW2, N, db2, M, N)
        assert_close_with_nonfinite(dW2, refW, tol, 1e-5, "dW k-tiled")
    finally:
        lib.tune("gemm_tn_fullk", 1)
        lib.tune("gemm_tn_pipe", 1)


@pytest.mark.parametrize("M,segs,segk,N", [(300, 5, 66, 128), (257, 5, 66, 64), (131, 2, 66, 100), (90, 1, 384, 20),
                                           (64, 2, 20, 192), (45, 1, 128, 128), (17, 1, 200, 64), (33, 4, 64, 4)])
def test_gemm_tn_pipelined_whole_k_schedule(backend, M, segs, segk, N):
    """gemm_tn_pipe_kernel (the weight-gradient kernel of the training step: whole K per workgroup, two LDS stages,
    1-3 k-tiles x 1-2 column tiles per wavefront) forced onto small problems so the CPU test double covers every
    instantiation, ragged last steps, masked k / n edges and the bias column sums; natural size on the GPU leg."""
    lib = _lib.get_lib()
    if backend.name == "hip":
        M = M * 97
    lib.tune("gemm_tn_pipe", 2)
    try:
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(segs, M, segk, generator=g)
        G = torch.randn(M, N, generator=g)
        dW0 = torch.randn(segs * segk, N, generator=g)
        db0 = torch.randn(N, generator=g)
        refW = dW0.double() + torch.cat([A[j] for j in range(segs)], dim=1).double().t() @ G.double()
        refb = db0.double() + G.double().sum(0)
        dW, db = dW0.clone().to(backend.device), db0.clone().to(backend.device)
        ops.gemm_tn_acc(A.to(backend.device), segk, M * segk, segs, segk, G.to(backend.device), N, dW, N, db, M, N)
        tol = 2e-4 if backend.name == "emu" else 2e-3        # sums of up to ~29 000 products on the GPU leg
        assert_close_with_nonfinite(dW, refW, tol, 1e-5, "dW")
        assert_close_with_nonfinite(db, refb, tol, 1e-5, "db")
        dW3 = dW0.clone().to(backend.device)
        ops.gemm_tn_acc(A.to(backend.device), segk, M * segk, segs, segk, G.to(backend.device), N, dW3, N, None, M, N)
        assert_close_with_nonfinite(dW3, refW, tol, 1e-5, "dW without bias")
    finally:
        lib.tune("gemm_tn_pipe", 1)


@pytest.mark.parametrize("O,peep", [(8, True), (6, True), (5, False), (64, True)])
def test_lstm_gate_kernels_against_autograd(backend, O, peep):
    """pgt_lstm_gates(_bwd)_f32 against the gate equations of gconv_lstm.py:138-172 written with torch autograd."""
    M = 37 if backend.name == "emu" else 5000
    g = torch.Generator().manual_seed(O)
    P = torch.randn(M, 4 * O, generator=g)
    C = torch.randn(M, O, generator=g)
    ws = [torch.randn(1, O, generator=g) if peep else None for _ in range(3)]
    wH, wC = torch.randn(M, O, generator=g), torch.randn(M, O, generator=g)

    def run(Pt, Ct, wt, fn):
        Hn, Cn = fn(Pt, Ct, *wt)
        ((Hn * wH.to(Hn.device, Hn.dtype)).sum() + (Cn * wC.to(Cn.device, Cn.dtype)).sum()).backward()
        return Hn, Cn

    def ref_fn(Pt, Ct, wi, wf, wo):
        z = lambda w: 0 if w is None else w * 1.0
        I = torch.sigmoid(Pt[:, :O] + (wi * Ct if wi is not None else 0))
        Fg = torch.sigmoid(Pt[:, O:2 * O] + (wf * Ct if wf is not None else 0))
        T = torch.tanh(Pt[:, 2 * O:3 * O])
        Cn = Fg * Ct + I * T
        Og = torch.sigmoid(Pt[:, 3 * O:] + (wo * Cn if wo is not None else 0))
        return Og * torch.tanh(Cn), Cn

    P64, C64 = P.double().requires_grad_(), C.double().requires_grad_()
    w64 = [None if w is None else w.double().requires_grad_() for w in ws]
    Hr, Cr = run(P64, C64, w64, ref_fn)
    Pd, Cd = backend.t(P).requires_grad_(), backend.t(C).requires_grad_()
    wd = [None if w is None else backend.t(w).requires_grad_() for w in ws]
    Hn, Cn = run(Pd, Cd, wd, ops.LSTMGatesFunction.apply)
    assert_close_with_nonfinite(Hn, Hr, 1e-5, 1e-5, "H")
    assert_close_with_nonfinite(Cn, Cr, 1e-5, 1e-5, "C")
    assert_close_with_nonfinite(Pd.grad, P64.grad, 2e-5, 1e-4, "dP")
    assert_close_with_nonfinite(Cd.grad, C64.grad, 2e-5, 1e-4, "dC")
    if peep:
        tol = 1e-4 if backend.name == "emu" else 5e-3       # column sums over M rows (fp32 atomics on the GPU leg)
        for a, b_, nm in zip(wd, w64, ("w_ci", "w_cf", "w_co")):
            assert_close_with_nonfinite(a.grad, b_.grad, tol, 1e-4, nm)


@pytest.mark.parametrize("M,segs,segk,O,fin", [(150, 5, 66, 64, 2), (70, 3, 10, 8, 2), (260, 1, 128, 32, 0),
                                               (90, 2, 7, 4, 3)])
@pytest.mark.parametrize("big", [0, 2])
def test_gemm_with_fused_gru_epilogues_is_bitwise_gemm_then_gates(backend, M, segs, segk, O, fin, big):
    """pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32 == pgt_gemm_f32 followed by pgt_gru_zr_f32 / pgt_gru_h_f32, bit for bit
    (vector and scalar epilogue paths, small and 128-wide tiles, the pipelined kernel, unaligned side outputs)."""
    lib = _lib.get_lib()
    lib.tune("gemm_small_tiles", big)
    try:
        g = torch.Generator().manual_seed(M + O)
        dev = backend.device
        K = segs * segk
        A = torch.randn(segs, M, segk, generator=g).to(dev)
        Wzr, bzr = torch.randn(K, 2 * O, generator=g).to(dev), torch.randn(2 * O, generator=g).to(dev)
        Wh, bh = torch.randn(K, O, generator=g).to(dev), torch.randn(O, generator=g).to(dev)
        H = torch.randn(M, O, generator=g).to(dev)
        C = fin + O
        # unfused
        zr0 = torch.empty(M, 2 * O, device=dev)
        xhr0 = torch.zeros(M, C, device=dev)
        ops.gemm(A, segk, M * segk, segs, segk, Wzr, 2 * O, 1, zr0, 2 * O, 0, 2 * O, bzr, M, 2 * O)
        ops._gru_zr(zr0, H, xhr0, fin)
        ht0 = torch.empty(M, O, device=dev)
        out0a, out1a = torch.empty(M, O, device=dev), torch.zeros(M, C, device=dev)
        ops.gemm(A, segk, M * segk, segs, segk, Wh, O, 1, ht0, O, 0, O, bh, M, O)
        ops._gru_h(ht0, zr0, H, out0a, out1a[:, fin:])
        # fused
        zr1 = torch.full((M, 2 * O), float("nan"), device=dev)
        xhr1 = torch.zeros(M, C, device=dev)
        ops.gemm_gru_zr(A, segk, M * segk, segs, segk, Wzr, 2 * O, 1, bzr, zr1, H, xhr1, fin)
        ht1 = torch.full((M, O), float("nan"), device=dev)
        out0b, out1b = torch.full((M, O), float("nan"), device=dev), torch.zeros(M, C, device=dev)
        ops.gemm_gru_h(A, segk, M * segk, segs, segk, Wh, O, 1, bh, ht1, zr1, H, out0b, out1b[:, fin:])
        assert torch.equal(zr0, zr1) and torch.equal(xhr0, xhr1)
        assert torch.equal(ht0, ht1) and torch.equal(out0a, out0b) and torch.equal(out1a, out1b)
        out0c = torch.full((M, O), float("nan"), device=dev)
        ops.gemm_gru_h(A, segk, M * segk, segs, segk, Wh, O, 1, None, ht1, zr1, H, out0c, None)   # no bias, no out1
        ops.gemm(A, segk, M * segk, segs, segk, Wh, O, 1, ht0, O, 0, O, None, M, O)
        ops._gru_h(ht0, zr0, H, out0a, None)
        assert torch.equal(out0a, out0c)
    finally:
        lib.tune("gemm_small_tiles", 0)


@pytest.mark.gpu
def test_gemm_schedules_agree_at_benchmark_size():
    """Size-independent property at BASELINE.json's full size (M = 1024 x 207 rows, the 330-wide diffusion stack):
    the pipelined kernels and the k-tiled kernels are two independent implementations of the same sums and must agree
    to fp32 rounding (forward NN 330 -> 128 / 64, feature gradient NT 128 / 64 -> 330, weight gradient over 12 steps),
    and the fused gate epilogues must equal GEMM-then-gate-kernel bit for bit."""
    lib = _lib.get_lib()
    if lib.target != "gfx950":
        pytest.skip("product library only")
    dev = torch.device("cuda:0")
    M, S, C, O = 1024 * 207, 5, 66, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn(S, M, C, generator=g).to(dev)
    W = (torch.randn(S * C, 2 * O, generator=g) / 18).to(dev)
    b = torch.randn(2 * O, generator=g).to(dev)
    out = {}
    lib.tune("gemm_bx", 0)            # this test is about the fp32 tile kernels (the split-bf16 kernel has its own below)
    try:
        for db in (1, 0):
            lib.tune("gemm_db", db)
            lib.tune("gemm_tn_pipe", db)
            C1 = torch.empty(M, 2 * O, device=dev)
            ops.gemm(A, C, M * C, S, C, W, 2 * O, 1, C1, 2 * O, 0, 2 * O, b, M, 2 * O)
            C2 = torch.empty(M, O, device=dev)
            ops.gemm(A, C, M * C, S, C, W[:, :O].contiguous(), O, 1, C2, O, 0, O, None, M, O)
            G = torch.empty(S, M, C, device=dev)
            ops.gemm(C1, 2 * O, 0, 1, 2 * O, W, 1, 2 * O, G, C, M * C, C, None, M, S * C)
            dW = torch.zeros(S * C, 2 * O, device=dev)
            db_ = torch.zeros(2 * O, device=dev)
            ops.gemm_tn_acc(A, C, M * C, S, C, C1, 2 * O, dW, 2 * O, db_, M, 2 * O)
            out[db] = (C1, C2, G, dW, db_)
        # the persistent deferred-store schedule at its natural grid (512 workgroups, 3 312 tiles): same sums in the same
        # order as the one-tile kernel -> bitwise, on the feature-gradient shape of the training step (hidden columns
        # only: four 64-wide output segments) and on a plain NN product
        dP = torch.randn(M, O, generator=g).to(dev)
        WH = (torch.randn(4 * O, O, generator=g) / 8).to(dev)
        res = {}
        for dbp in (1, 0):
            lib.tune("gemm_dbp", dbp)
            Gh = torch.full((4, M, O), float("nan"), device=dev)
            ops.gemm(dP, O, 0, 1, O, WH, 1, O, Gh, O, M * O, O, None, M, 4 * O)
            Cn = torch.full((M, 2 * O), float("nan"), device=dev)
            ops.gemm(A, C, M * C, S, C, W, 2 * O, 1, Cn, 2 * O, 0, 2 * O, b, M, 2 * O)
            res[dbp] = (Gh, Cn)
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert_close_with_nonfinite(res[1][0][3], dP.cpu().double() @ WH[3 * O:].cpu().double().t(), 1e-4, 1e-5, "persistent NT")
    finally:
        lib.tune("gemm_db", 1)
        lib.tune("gemm_tn_pipe", 1)
        lib.tune("gemm_dbp", 1)
    names = ("NN 330->128", "NN 330->64", "NT 128->330", "dW", "db")
    for name, x, y in zip(names, out[1], out[0]):
        scale = float(y.abs().max())
        tol = 2e-3 if name in ("dW", "db") else 2e-5           # dW / db: 212 k-term sums, atomics in two orders
        assert float((x - y).abs().max()) <= tol * max(scale, 1.0), name
    # ... and every shape against an fp64 product: 512 sampled rows of the tall outputs, the whole dW / db
    rows = torch.from_numpy(np.random.default_rng(8).choice(M, size=512, replace=False)).sort().values
    A64 = A.cpu().double()                                                     # [S, M, C]
    As = A64[:, rows].permute(1, 0, 2).reshape(512, S * C)                     # sampled rows of the K-segmented operand
    W64, b64 = W.cpu().double(), b.cpu().double()
    for db in (1, 0):
        C1, C2, G, dW, db_ = out[db]
        assert_close_with_nonfinite(C1[rows.to(dev)], As @ W64 + b64, 2e-5, 2e-5, f"db={db} NN 330->128 vs fp64")
        assert_close_with_nonfinite(C2[rows.to(dev)], As @ W64[:, :O], 2e-5, 2e-5, f"db={db} NN 330->64 vs fp64")
        G64 = (C1[rows.to(dev)].cpu().double() @ W64.t()).view(512, S, C).permute(1, 0, 2)
        assert_close_with_nonfinite(G[:, rows.to(dev)], G64, 2e-5, 2e-5, f"db={db} NT 128->330 vs fp64")
        C164 = C1.cpu().double()
        dW64 = A64.permute(1, 0, 2).reshape(M, S * C).t() @ C164
        scale = float(dW64.abs().max())
        assert float((dW.cpu().double() - dW64).abs().max()) <= 2e-4 * scale, f"db={db} dW vs fp64"
        assert float((db_.cpu().double() - C164.sum(0)).abs().max()) <= 2e-4 * float(C164.sum(0).abs().max()), f"db={db} db"
    del A64, C164
    # fused epilogues at full size
    H = torch.randn(M, O, generator=g).to(dev)
    zr0, xhr0 = out[1][0].clone(), torch.zeros(M, C, device=dev)
    ops._gru_zr(zr0, H, xhr0, 2)
    zr1, xhr1 = torch.empty(M, 2 * O, device=dev), torch.zeros(M, C, device=dev)
    ops.gemm_gru_zr(A, C, M * C, S, C, W, 2 * O, 1, b, zr1, H, xhr1, 2)
    lib.tune("gemm_bx", 1)
    assert torch.equal(zr0, zr1) and torch.equal(xhr0, xhr1)


def _bx_case(M, segs, segk, N, nt, seed):
    """Operands of one product C = A . B (+ bias): A in `segs` segments of `segk` columns; B as [K, N] (NN) or, for the
    feature-gradient form, as W^T with unit k stride and the output cut into 64-column segments."""
    g = torch.Generator().manual_seed(seed)
    K = segs * segk
    A = torch.randn(segs, M, segk, generator=g)
    B = torch.randn(K, N, generator=g) / K ** 0.5
    bias = None if nt else torch.randn(N, generator=g)
    A2 = torch.cat([A[j] for j in range(segs)], dim=1).double()
    ref = A2 @ B.double() + (0 if bias is None else bias.double())
    return A, B, bias, ref


def _bx_run(A, B, bias, M, segs, segk, N, nt, dev):
    Ad = A.to(dev)
    if nt:
        Bt = B.t().contiguous().to(dev)                          # [N, K]: unit k stride, column stride K
        nseg_out = N // 64
        Cs = torch.full((nseg_out, M, 64), float("nan"), device=dev)
        ops.gemm(Ad, segk, M * segk, segs, segk, Bt, 1, segs * segk, Cs, 64, M * 64, 64, None, M, N)
        return Cs.permute(1, 0, 2).reshape(M, N)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(Ad, segk, M * segk, segs, segk, B.to(dev), N, 1, C, N, 0, N, None if bias is None else bias.to(dev), M, N)
    return C


SYM_PC_DEFAULT = 1      # csrc/gemm_bx.hip: g_bx_sym_pc


@pytest.mark.parametrize("M,segs,segk,N,nt", [(5000, 5, 66, 128, False), (4131, 5, 66, 64, False), (3000, 3, 34, 96, False),
                                              (7000, 1, 128, 256, True), (2500, 1, 64, 256, True), (3333, 1, 128, 64, True),
                                              (6100, 1, 128, 320, True), (2049, 1, 64, 320, True), (900, 1, 100, 192, True),
                                              (2000, 1, 64, 64, True), (100, 5, 66, 128, False), (31, 1, 64, 64, True),
                                              (1025, 2, 24, 40, False), (640, 1, 16, 128, True)])
def test_gemm_split_bf16_is_at_least_as_accurate_as_the_fp32_kernels(backend, M, segs, segk, N, nt):
    """gemm_bx.hip: fp32 operands split into three bf16 pieces, six piece products on the bf16 matrix pipe, fp32
    accumulation.  Against an fp64 product it must be no worse than the exact-fp32 MFMA kernels on the same operands (it
    is better: big term and corrections are accumulated separately), at ragged sizes, every K bucket (<= 64, <= 128,
    <= 336 columns, zero padded), one and two 32-column blocks per wavefront, segmented inputs and outputs.  The same
    kernel bodies run on the CPU test double (csrc/gemm_bx.hip's platform layer) at a few hundred rows."""
    lib = _lib.get_lib()
    dev = backend.device
    if backend.name == "emu":
        M = min(M, 70 + M % 97)
    A, B, bias, ref = _bx_case(M, segs, segk, N, nt, seed=M + N)
    try:
        lib.tune("gemm_bx", 2)
        C_bx = _bx_run(A, B, bias, M, segs, segk, N, nt, dev).cpu().double()
        lib.tune("gemm_bx", 0)
        C_32 = _bx_run(A, B, bias, M, segs, segk, N, nt, dev).cpu().double()
    finally:
        lib.tune("gemm_bx", 1)
    if nt and N > 128:       # the short-K kernel in its other wavefront organisation (all alike <-> producers / consumers): the same bits
        try:
            lib.tune("gemm_bx", 2)
            lib.tune("gemm_bx_sym_pc", 1 - SYM_PC_DEFAULT)
            C_other = _bx_run(A, B, bias, M, segs, segk, N, nt, dev).cpu().double()
        finally:
            lib.tune("gemm_bx_sym_pc", SYM_PC_DEFAULT)
            lib.tune("gemm_bx", 1)
        assert torch.equal(C_bx, C_other)
    assert torch.isfinite(C_bx).all()
    e_bx, e_32 = (C_bx - ref).abs(), (C_32 - ref).abs()
    scale = float(ref.abs().max())
    assert float(e_bx.max()) <= 2e-6 * max(scale, 1.0), (float(e_bx.max()), scale)
    assert float(e_bx.mean()) <= 1.1 * float(e_32.mean()) + 1e-9, (float(e_bx.mean()), float(e_32.mean()))
    assert float((C_bx - C_32).abs().max()) <= 1e-5 * max(scale, 1.0)
    if (segs, segk, N) in ((5, 66, 128), (1, 128, 256), (1, 64, 320)):
        assert not torch.equal(C_bx, C_32)      # different arithmetic: identical bits would mean the kernel did not run


def test_gemm_split_bf16_exact_cases_and_non_finite_rows(backend):
    lib = _lib.get_lib()
    dev = backend.device
    try:
        lib.tune("gemm_bx", 2)
        # integers: every piece product and every partial sum is exact -> the result is exact
        A = torch.randint(-200, 201, (300, 64)).float()
        W = torch.randint(-200, 201, (64, 64)).float()
        C = torch.empty(300, 64, device=dev)
        ops.gemm(A.to(dev), 64, 0, 1, 64, W.to(dev), 64, 1, C, 64, 0, 64, None, 300, 64)
        assert torch.equal(C.cpu().double(), A.double() @ W.double())
        # asymmetric-B identity check (a transposed or permuted store would show)
        I = torch.eye(64)
        Bm = (torch.arange(64 * 64, dtype=torch.float32).view(64, 64) * 1.0009765625)   # needs all three pieces
        ops.gemm(I.to(dev), 64, 0, 1, 64, Bm.to(dev), 64, 1, C[:64], 64, 0, 64, None, 64, 64)
        assert torch.equal(C[:64].cpu(), Bm)
        # 24-bit operands: x = x1 + x2 + x3 exactly, one product per output -> only the dropped piece products remain
        x = torch.rand(64, 64) + 0.5
        ops.gemm(x.to(dev), 64, 0, 1, 64, (I * 1.2345678).to(dev), 64, 1, C[:64], 64, 0, 64, None, 64, 64)
        assert float((C[:64].cpu().double() - x.double() * float(torch.tensor(1.2345678))).abs().max()) <= 2.0 ** -23 * 2.5
    finally:
        lib.tune("gemm_bx", 1)


@pytest.mark.parametrize("M,segs,segk,N,nt", [(200, 5, 66, 128, False), (170, 5, 66, 64, False), (150, 1, 128, 256, True),
                                              (130, 1, 64, 320, True), (140, 3, 34, 96, False)])
def test_gemm_split_bf16_non_finite_operands_land_where_fp32_puts_them(backend, M, segs, segk, N, nt):
    """inf / nan in the streaming operand or in the weights: a split piece `x - x1` is nan, so the six piece products give
    a nan ROW (operand) or COLUMN (weight) where the reference's fp32 product (torch.matmul, dcrnn.py:81-105) has +-inf /
    nan element by element.  The kernels detect the nan sums per 32 x 32 tile and redo the tile as an fp32 fmaf chain
    (bx_exact_tile): value AND placement must equal the exact-fp32 kernels' and a plain fp32 matmul's, every K bucket,
    one / two column blocks, the symmetric short-K kernel with its second blocks."""
    lib = _lib.get_lib()
    dev = backend.device
    A, B, bias, _ = _bx_case(M, segs, segk, N, nt, seed=3 + N)
    inf, nan = float("inf"), float("nan")
    A[segs - 1, 17, 5] = nan
    A[0, 40, segk - 1] = inf
    A[0, 41, 0] = -inf
    A[segs // 2, 77, 3] = inf
    A[segs // 2, 77, 4] = -inf                 # +inf and -inf in one row: nan where both weights are non-zero
    A[0, M - 1, 1] = inf                       # ragged last block
    K = segs * segk
    Bm = B.clone()                             # [K, N] (the NT form is transposed inside _bx_run)
    Bm[7, 5] = 0.0                             # inf * 0 = nan exactly there (rows 77: k = 3, 4 -> not this one; row 41: k = 0)
    Bm[0, 6] = 0.0                             # row 41 (k = 0): nan in column 6, -+inf elsewhere
    Aflat = torch.cat([A[s] for s in range(segs)], dim=1)
    b0 = 0.0 if bias is None else bias
    ref = Aflat.double() @ Bm.double() + (0.0 if bias is None else bias.double())
    ref32 = Aflat @ Bm + b0                    # torch's fp32 product: the reference's arithmetic
    try:
        lib.tune("gemm_bx", 2)
        out = _bx_run(A, Bm, bias, M, segs, segk, N, nt, dev).cpu()
        lib.tune("gemm_bx", 0)
        out32 = _bx_run(A, Bm, bias, M, segs, segk, N, nt, dev).cpu()
    finally:
        lib.tune("gemm_bx", 1)
    bad_rows = [17, 40, 41, 77, M - 1]
    assert not torch.isfinite(out32[bad_rows]).all()
    assert_close_with_nonfinite(out, out32, 2e-5, 2e-5, "split-bf16 vs exact-fp32 kernels")
    assert_close_with_nonfinite(out, ref32, 2e-5, 2e-5, "split-bf16 vs torch fp32 matmul")
    good = torch.ones(M, dtype=torch.bool)
    good[bad_rows] = False
    assert torch.isfinite(out[good]).all()
    assert float((out[good].double() - ref[good]).abs().max()) <= 2e-6 * float(ref[good].abs().max())
    # a non-finite WEIGHT: its column (all rows) must follow the fp32 product too
    B2 = B.clone()
    B2[11, 9] = inf
    A2, _, _, _ = _bx_case(M, segs, segk, N, nt, seed=3 + N)
    A2[0, 3, 11 if segk > 11 else 1] = 0.0     # 0 * inf = nan in that one row
    try:
        lib.tune("gemm_bx", 2)
        outw = _bx_run(A2, B2, bias, M, segs, segk, N, nt, dev).cpu()
        lib.tune("gemm_bx", 0)
        outw32 = _bx_run(A2, B2, bias, M, segs, segk, N, nt, dev).cpu()
    finally:
        lib.tune("gemm_bx", 1)
    assert torch.isinf(outw32[:, 9]).any()
    assert_close_with_nonfinite(outw, outw32, 2e-5, 2e-5, "non-finite weight")


@pytest.mark.parametrize("M,segs,O,fin,bx", [(9000, 5, 48, 2, 1), (9000, 2, 32, 2, 2), (8200, 5, 128, 2, 1)])
def test_gemm_split_bf16_declines_gate_shapes_it_cannot_tile(backend, M, segs, O, fin, bx):
    """The fused gate epilogues of the split-bf16 kernel need whole 32-column blocks on either side of the z | r boundary,
    K in the 21-step bucket and (candidate gate) at most 64 columns; every other shape must fall through to the fp32
    kernels — bit for bit the result with the kernel switched off (hidden 48: the boundary cuts a block; K = 68: short-K
    kernels have no epilogue; hidden 128: the instantiation would spill)."""
    lib = _lib.get_lib()
    dev = backend.device
    if backend.name == "emu":
        M, bx = 150, 2                  # the size gate (>= 8192 rows) is not what this test is about
    g = torch.Generator().manual_seed(M + O)
    segk = fin + O
    K, C = segs * segk, fin + O
    A = torch.randn(segs, M, segk, generator=g).to(dev)
    Wzr, bzr = (torch.randn(K, 2 * O, generator=g) / K ** 0.5).to(dev), torch.randn(2 * O, generator=g).to(dev)
    Wh, bh = (torch.randn(K, O, generator=g) / K ** 0.5).to(dev), torch.randn(O, generator=g).to(dev)
    H = torch.randn(M, O, generator=g).to(dev)
    res = {}
    try:
        for mode in (bx, 0):
            lib.tune("gemm_bx", mode)
            zr, xhr = torch.full((M, 2 * O), float("nan"), device=dev), torch.zeros(M, C, device=dev)
            ops.gemm_gru_zr(A, segk, M * segk, segs, segk, Wzr, 2 * O, 1, bzr, zr, H, xhr, fin)
            ht, out0 = torch.full((M, O), float("nan"), device=dev), torch.full((M, O), float("nan"), device=dev)
            ops.gemm_gru_h(A, segk, M * segk, segs, segk, Wh, O, 1, bh, ht, zr, H, out0, None)
            res[mode] = (zr, xhr, ht, out0)
    finally:
        lib.tune("gemm_bx", 1)
    if O == 128:
        # hidden 128: the z | r product (256 columns, K = 650) is outside the kernel as well; only the fall-through matters
        pass
    for name, a, b in zip(("zr", "xhr", "ht", "out0"), res[bx], res[0]):
        if O == 48 and name in ("ht", "out0"):
            continue                      # the 48-column candidate gate IS covered (N = hidden <= 64): compared below
        assert torch.equal(a, b), name
    if O == 48:
        for a, b in zip(res[bx][2:], res[0][2:]):
            assert float((a - b).abs().max()) <= 3e-6


@pytest.mark.parametrize("M,O,fin", [(4100, 64, 2), (1000, 32, 2), (33, 64, 0)])
def test_gemm_split_bf16_fused_gru_epilogues(backend, M, O, fin):
    """The gate epilogues of pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32 on the split-bf16 kernel against the fp32 kernels
    (same gate formulas on sums that differ by fp32 rounding) and against an fp64 evaluation."""
    lib = _lib.get_lib()
    dev = backend.device
    if backend.name == "emu":
        M = min(M, 101)
    g = torch.Generator().manual_seed(M + O)
    segs, segk = 5, fin + O
    K, C = segs * segk, fin + O
    A = torch.randn(segs, M, segk, generator=g).to(dev)
    Wzr, bzr = (torch.randn(K, 2 * O, generator=g) / K ** 0.5).to(dev), torch.randn(2 * O, generator=g).to(dev)
    Wh, bh = (torch.randn(K, O, generator=g) / K ** 0.5).to(dev), torch.randn(O, generator=g).to(dev)
    H = torch.randn(M, O, generator=g).to(dev)
    res = {}
    try:
        for bx in (2, 0):
            lib.tune("gemm_bx", bx)
            zr = torch.full((M, 2 * O), float("nan"), device=dev)
            xhr = torch.zeros(M, C, device=dev)
            ops.gemm_gru_zr(A, segk, M * segk, segs, segk, Wzr, 2 * O, 1, bzr, zr, H, xhr, fin)
            ht = torch.full((M, O), float("nan"), device=dev)
            out0, out1 = torch.full((M, O), float("nan"), device=dev), torch.zeros(M, C, device=dev)
            ops.gemm_gru_h(A, segk, M * segk, segs, segk, Wh, O, 1, bh, ht, zr, H, out0, out1[:, fin:])
            res[bx] = (zr, xhr, ht, out0, out1)
    finally:
        lib.tune("gemm_bx", 1)
    for name, a, b in zip(("zr", "xhr", "ht", "out0", "out1"), res[2], res[0]):
        assert torch.isfinite(a).all(), name
        assert float((a - b).abs().max()) <= 3e-6, name
    A2 = torch.cat([A[j] for j in range(segs)], dim=1).cpu().double()
    zr64 = torch.sigmoid(A2 @ Wzr.cpu().double() + bzr.cpu().double())
    assert float((res[2][0].cpu().double() - zr64).abs().max()) <= 1e-6
    assert float((res[2][1][:, fin:].cpu().double() - H.cpu().double() * zr64[:, O:]).abs().max()) <= 4e-6
    ht64 = torch.tanh(A2 @ Wh.cpu().double() + bh.cpu().double())
    z = res[2][0][:, :O].cpu().double()
    assert float((res[2][3].cpu().double() - (z * H.cpu().double() + (1 - z) * ht64)).abs().max()) <= 4e-6


TN_PC_DEFAULT = 1      # csrc/gemm_bx.hip: g_bx_tn_pc


@pytest.mark.parametrize("M,segs,segk,N,force", [(20000, 5, 66, 128, False), (17001, 5, 66, 64, False), (3000, 3, 60, 100, True),
                                                 (1000, 5, 66, 128, True), (517, 2, 66, 33, True), (40, 5, 70, 128, True),
                                                 (2048, 3, 61, 50, True)])
def test_gemm_tn_split_bf16_weight_and_bias_gradient(backend, M, segs, segk, N, force):
    """gemm_bx_tn_kernel: dW += A^T G and db += column sums of G through the bf16 matrix pipe (both operands as three
    bf16 planes, transposed in LDS; the bias gradient from a row of ones) against fp64 and against the fp32 kernels:
    ragged row counts, several launches per product (forced small chunks), accumulation into non-zero dW / db, db absent,
    and the deterministic mode (bitwise reproducible)."""
    lib = _lib.get_lib()
    dev = backend.device
    if backend.name == "emu":
        M, force = min(M, 150 + M % 61), True
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(segs, M, segk, generator=g)
    G = torch.randn(M, N, generator=g)
    dW0, db0 = torch.randn(segs * segk, N, generator=g), torch.randn(N, generator=g)
    A2 = torch.cat([A[j] for j in range(segs)], dim=1).double()
    refW, refb = dW0.double() + A2.t() @ G.double(), db0.double() + G.double().sum(0)
    Ad, Gd = A.to(dev), G.to(dev)
    out = {}
    keep_det = ops.DETERMINISTIC_WEIGHT_GRADIENTS
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = False           # first the fp32-atomics form of every kernel (the library's default is the other)
    try:
        for bx in ((2 if force else 1), 0):
            lib.tune("gemm_bx", bx)
            dW, db = dW0.clone().to(dev), db0.clone().to(dev)
            ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, Gd, N, dW, N, db, M, N)
            out[bx] = (dW, db)
        bx = 2 if force else 1
        lib.tune("gemm_bx", bx)
        dWn = dW0.clone().to(dev)
        ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, Gd, N, dWn, N, None, M, N)             # no bias gradient
        det = []
        old = ops.DETERMINISTIC_WEIGHT_GRADIENTS
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = True
        try:
            for _ in range(2):
                dW, db = dW0.clone().to(dev), db0.clone().to(dev)
                ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, Gd, N, dW, N, db, M, N)
                det.append((dW, db))
            # the other wavefront organisation (all alike <-> producers / consumers): the same planes and the same six piece
            # products per block in the same order — the same bits
            others = []
            for form in (0, 1, 2):                       # all alike; four + four; twelve wavefronts (8 + 4 / 4 + 8 by width)
                lib.tune("gemm_bx_tn_pc", form)
                dW, db = dW0.clone().to(dev), db0.clone().to(dev)
                ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, Gd, N, dW, N, db, M, N)
                others.append((dW, db))
            other = others[0]
            dWo = dW0.clone().to(dev)
            ops.DETERMINISTIC_WEIGHT_GRADIENTS = old
            ops.gemm_tn_acc(Ad, segk, M * segk, segs, segk, Gd, N, dWo, N, None, M, N)         # its atomics form, no bias gradient
        finally:
            ops.DETERMINISTIC_WEIGHT_GRADIENTS = old
            lib.tune("gemm_bx_tn_pc", TN_PC_DEFAULT)
    finally:
        lib.tune("gemm_bx", 1)
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = keep_det
    sw, sb = float(refW.abs().max()), float(refb.abs().max())
    for name, (dW, db) in (("split-bf16", out[bx]), ("fp32", out[0]), ("deterministic", det[0])):
        assert float((dW.cpu().double() - refW).abs().max()) <= 3e-6 * sw, name
        assert float((db.cpu().double() - refb).abs().max()) <= 3e-6 * sb + 1e-5, name
    assert float((dWn.cpu().double() - refW).abs().max()) <= 3e-6 * sw
    assert torch.equal(det[0][0], det[1][0]) and torch.equal(det[0][1], det[1][1])
    for other in others:
        assert torch.equal(det[0][0], other[0]) and torch.equal(det[0][1], other[1])
    assert float((dWo.cpu().double() - refW).abs().max()) <= 3e-6 * sw
    e_bx, e_32 = float((out[bx][0].cpu().double() - refW).abs().mean()), float((out[0][0].cpu().double() - refW).abs().mean())
    if backend.name == "hip":          # at a few hundred rows (test double) both errors are a handful of roundings
        assert e_bx <= 1.5 * e_32 + 1e-9, (e_bx, e_32)


@pytest.mark.parametrize("M,O,fin", [(300, 64, 2), (100, 32, 2)])
def test_gemm_split_bf16_gate_epilogues_and_weight_gradient_with_non_finite_operands(backend, M, O, fin):
    """The fused z | r and candidate-gate products (K cut two / four ways, partial sums through LDS) and the weight-gradient
    kernel with inf / nan operands — what DConv's infinite norm_in on a graph with a zero in-degree node feeds them
    (dcrnn.py:279-290): every output, side outputs included, must carry its non-finite entries
    exactly where the exact-fp32 kernels (== the reference's fp32 arithmetic) put them."""
    lib = _lib.get_lib()
    dev = backend.device
    inf, nan = float("inf"), float("nan")
    g = torch.Generator().manual_seed(M + O)
    segs, segk = 5, fin + O
    K, C = segs * segk, fin + O
    A = torch.randn(segs, M, segk, generator=g)
    A[2, 5, 3] = inf
    A[2, 5, 9] = -inf
    A[4, 33, segk - 1] = -inf
    A[0, 64, 0] = nan
    A[1, M - 1, 7] = inf
    A = A.to(dev)
    Wzr, bzr = (torch.randn(K, 2 * O, generator=g) / K ** 0.5).to(dev), torch.randn(2 * O, generator=g).to(dev)
    Wh, bh = (torch.randn(K, O, generator=g) / K ** 0.5), torch.randn(O, generator=g).to(dev)
    Wh[2 * segk + 3, 4] = 0.0          # inf * 0
    Wh = Wh.to(dev)
    H = torch.randn(M, O, generator=g).to(dev)
    G = torch.randn(M, 2 * O, generator=g)
    G[5, 1] = 0.0                      # inf * 0 in the weight gradient
    G[70, 3] = inf                     # a non-finite gradient row
    G = G.to(dev)
    res = {}
    try:
        for bx in (2, 0):
            lib.tune("gemm_bx", bx)
            zr = torch.full((M, 2 * O), 7.0, device=dev)
            xhr = torch.zeros(M, C, device=dev)
            ops.gemm_gru_zr(A, segk, M * segk, segs, segk, Wzr, 2 * O, 1, bzr, zr, H, xhr, fin)
            zr_fin = torch.sigmoid(torch.randn(M, 2 * O, generator=torch.Generator().manual_seed(1))).to(dev)
            ht = torch.full((M, O), 7.0, device=dev)
            out0, out1 = torch.full((M, O), 7.0, device=dev), torch.zeros(M, C, device=dev)
            ops.gemm_gru_h(A, segk, M * segk, segs, segk, Wh, O, 1, bh, ht, zr_fin, H, out0, out1[:, fin:])
            dW, db = torch.zeros(K, 2 * O, device=dev), torch.zeros(2 * O, device=dev)
            ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, 2 * O, dW, 2 * O, db, M, 2 * O)
            res[bx] = (zr, xhr, ht, out0, out1, dW, db)
    finally:
        lib.tune("gemm_bx", 1)
    assert not torch.isfinite(res[0][5]).all() and not torch.isfinite(res[0][3]).all()
    for name, a, b in zip(("zr", "xhr", "ht", "out0", "out1", "dW", "db"), res[2], res[0]):
        tol = 2e-4 if name in ("dW", "db") else 4e-6
        assert_close_with_nonfinite(a, b, tol, tol, name)


def test_gemm_split_bf16_fuzz_shapes_on_the_test_double(emu_backend):
    """Seeded sweep over shapes around every routing boundary of gemm_bx.hip (K buckets 64 / 128 / 336, one / two / three
    column-block layouts, 256 / 320 columns, odd row counts, segmented inputs and outputs, with and without bias), the
    kernels forced on at any size: every result against fp64.  Shapes the kernels decline run the fp32 kernels — the
    point is that NO routing decision produces a wrong result."""
    lib = _lib.get_lib()
    rng = np.random.default_rng(2024)
    lib.tune("gemm_bx", 2)
    try:
        for case in range(48):
            segs = int(rng.integers(1, 6))
            segk = 2 * int(rng.integers(1, 34))
            if segs * segk > 336:
                segk = 2 * (336 // (2 * segs))
            K = segs * segk
            nt = bool(rng.integers(0, 2)) and K <= 128
            N = int(rng.choice([1, 31, 32, 33, 64, 96, 100, 128])) if K > 128 else int(rng.choice([1, 40, 64, 128, 129, 192, 256, 288, 320]))
            if nt:
                N = 64 * max(1, N // 64)
            M = int(rng.integers(1, 130))
            A, B, bias, ref = _bx_case(M, segs, segk, N, nt, seed=1000 + case)
            if not nt and rng.integers(0, 2):
                bias, ref = None, ref - bias.double()
            C = _bx_run(A, B, bias, M, segs, segk, N, nt, emu_backend.device).double()
            scale = max(float(ref.abs().max()), 1.0)
            assert torch.isfinite(C).all(), (case, M, segs, segk, N, nt)
            assert float((C - ref).abs().max()) <= 3e-6 * scale, (case, M, segs, segk, N, nt, float((C - ref).abs().max()))
        for case in range(12):                                   # weight gradients: 128 < K <= 351, N <= 128
            segs = int(rng.integers(2, 6))
            segk = int(rng.integers(130 // segs + 1, 351 // segs + 1))
            N = int(rng.choice([1, 17, 32, 64, 65, 100, 128]))
            M = int(rng.integers(1, 260))
            g = torch.Generator().manual_seed(2000 + case)
            A, G = torch.randn(segs, M, segk, generator=g), torch.randn(M, N, generator=g)
            dW0, db0 = torch.randn(segs * segk, N, generator=g), torch.randn(N, generator=g)
            A2 = torch.cat([A[j] for j in range(segs)], dim=1).double()
            refW, refb = dW0.double() + A2.t() @ G.double(), db0.double() + G.double().sum(0)
            dW, db = dW0.clone(), db0.clone()
            ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dW, N, db, M, N)
            assert float((dW.double() - refW).abs().max()) <= 3e-6 * max(float(refW.abs().max()), 1.0), (case, M, segs, segk, N)
            assert float((db.double() - refb).abs().max()) <= 3e-6 * max(float(refb.abs().max()), 1.0) + 1e-5, (case, M, segs, segk, N)
    finally:
        lib.tune("gemm_bx", 1)


@pytest.mark.gpu
def test_gemm_split_bf16_at_benchmark_size():
    """The three products the training step routes to the split-bf16 kernel, at BASELINE.json's full size (M = 1024 x 207
    rows out of a 3-step diffusion stack, segment stride 3 M C): 512 sampled rows + the last (ragged) row block against
    fp64, and against the fp32 kernels; the fused z | r epilogue against GEMM-then-gates."""
    lib = _lib.get_lib()
    if lib.target != "gfx950":
        pytest.skip("product library only")
    dev = torch.device("cuda:0")
    M, S, C, O, T = 1024 * 207 - 13, 5, 66, 64, 3
    g = torch.Generator(device="cpu").manual_seed(11)
    TS = torch.randn(S, T, M, C, generator=g).to(dev)
    A = TS[:, 1]
    W = (torch.randn(S * C, 2 * O, generator=g) / 18).to(dev)
    b = torch.randn(2 * O, generator=g).to(dev)
    H = torch.randn(M, O, generator=g).to(dev)
    dG = torch.randn(M, 2 * O, generator=g).to(dev)
    rows = torch.from_numpy(np.random.default_rng(8).choice(M - 40, size=512, replace=False)).sort().values
    rows = torch.cat([rows, torch.arange(M - 40, M)])
    A64 = A[:, rows.to(dev)].cpu().double().permute(1, 0, 2).reshape(len(rows), S * C)
    W64, b64 = W.cpu().double(), b.cpu().double()
    res = {}
    try:
        for bx in (1, 0):
            lib.tune("gemm_bx", bx)
            C1 = torch.full((M, 2 * O), float("nan"), device=dev)
            ops.gemm(A, C, T * M * C, S, C, W, 2 * O, 1, C1, 2 * O, 0, 2 * O, b, M, 2 * O)
            zr, xhr = torch.full((M, 2 * O), float("nan"), device=dev), torch.zeros(M, C, device=dev)
            ops.gemm_gru_zr(A, C, T * M * C, S, C, W, 2 * O, 1, b, zr, H, xhr, 2)
            G = torch.full((4, M, O), float("nan"), device=dev)
            ops.gemm(dG, 2 * O, 0, 1, 2 * O, W[:4 * O], 1, 2 * O, G, O, M * O, O, None, M, 4 * O)
            res[bx] = (C1, zr, xhr, G)
    finally:
        lib.tune("gemm_bx", 1)
    for bx in (1, 0):
        C1, zr, xhr, G = res[bx]
        assert torch.isfinite(C1).all() and torch.isfinite(zr).all() and torch.isfinite(G).all()
        ref = A64 @ W64 + b64
        tol = 2e-6 if bx else 2e-5
        assert_close_with_nonfinite(C1[rows.to(dev)], ref, tol, tol, f"bx={bx} NN 330->128 vs fp64")
        assert_close_with_nonfinite(zr[rows.to(dev)], torch.sigmoid(ref), tol, tol, f"bx={bx} z|r vs fp64")
        assert_close_with_nonfinite(xhr[rows.to(dev)][:, 2:], H[rows.to(dev)].cpu().double() * torch.sigmoid(ref)[:, O:], 4 * tol, tol,
                                    f"bx={bx} r*H vs fp64")
        G64 = (dG[rows.to(dev)].cpu().double() @ W64[:4 * O].t()).view(len(rows), 4, O).permute(1, 0, 2)
        assert_close_with_nonfinite(G[:, rows.to(dev)], G64, 2 * tol, tol, f"bx={bx} NT 128->256 vs fp64")
    for a, b_ in zip(res[1], res[0]):
        assert float((a - b_).abs().max()) <= 2e-5 * max(1.0, float(b_.abs().max()))
    # the weight gradient of the same product over all M rows (two launches of the split-bf16 kernel), against fp64
    A64f = A.cpu().double().permute(1, 0, 2).reshape(M, S * C)
    dW64, db64 = A64f.t() @ dG.cpu().double(), dG.cpu().double().sum(0)
    del A64f
    for bx in (1, 0):
        lib.tune("gemm_bx", bx)
        dW, db_ = torch.zeros(S * C, 2 * O, device=dev), torch.zeros(2 * O, device=dev)
        ops.gemm_tn_acc(A, C, T * M * C, S, C, dG, 2 * O, dW, 2 * O, db_, M, 2 * O)
        assert float((dW.cpu().double() - dW64).abs().max()) <= 2e-5 * float(dW64.abs().max()), f"bx={bx} dW vs fp64"
        assert float((db_.cpu().double() - db64).abs().max()) <= 2e-5 * float(db64.abs().max()) + 1e-3, f"bx={bx} db vs fp64"
    lib.tune("gemm_bx", 1)
    # fused epilogue vs product then gate kernel: the same sums; the fused chain uses the hardware exp / reciprocal
    zr0, xhr0 = res[1][0].clone(), torch.zeros(M, C, device=dev)
    ops._gru_zr(zr0, H, xhr0, 2)
    assert float((zr0 - res[1][1]).abs().max()) <= 3e-7 and float((xhr0 - res[1][2]).abs().max()) <= 3e-6


@pytest.mark.parametrize("M,K,N", [(333, 64, 2), (130, 128, 1), (77, 36, 3), (260, 256, 4), (65, 100, 2), (64, 16, 4)])
def test_gemm_streaming_kernels_for_skinny_shapes(backend, M, K, N):
    """The read-out layer `Linear(hidden, 1..4)` and its two gradients (gemm_skinny_n / gemm_skinny_k /
    gemm_tn_skinny kernels: an extent <= 4 leaves nothing for the matrix pipe).  Forced onto small sizes for the CPU
    test double: ragged row groups, K not a multiple of 64, NN and NT weight layouts, bias, accumulate, padded rows."""
    lib = _lib.get_lib()
    if backend.name == "hip":
        M = M * 301
    lib.tune("gemm_skinny", 2)
    try:
        g = torch.Generator().manual_seed(M + K + N)
        lda = K + 4                                                    # padded rows
        A = torch.randn(M, lda, generator=g)
        W = torch.randn(N, K, generator=g)                             # nn.Linear layout [out, in]
        b = torch.randn(N, generator=g)
        Ad, Wd, bd = A.to(backend.device), W.to(backend.device), b.to(backend.device)
        ref = A[:, :K].double() @ W.double().t() + b.double()
        # forward, NT weights (sbk = 1, sbn = K)
        Y = torch.full((M, N), float("nan"), device=backend.device)
        ops.gemm(Ad, lda, 0, 1, K, Wd, 1, K, Y, N, 0, N, bd, M, N)
        assert_close_with_nonfinite(Y, ref, 2e-4, 1e-5, "skinny forward (NT)")
        ops.gemm(Ad, lda, 0, 1, K, Wd, 1, K, Y, N, 0, N, None, M, N, accumulate=True)
        assert_close_with_nonfinite(Y, 2 * ref - b.double(), 4e-4, 1e-5, "skinny forward accumulate")
        # forward, NN weights ([K, N] row-major)
        Wt = W.t().contiguous().to(backend.device)
        Y2 = torch.full((M, N), float("nan"), device=backend.device)
        ops.gemm(Ad, lda, 0, 1, K, Wt, N, 1, Y2, N, 0, N, bd, M, N)
        assert_close_with_nonfinite(Y2, ref, 2e-4, 1e-5, "skinny forward (NN)")
        # input gradient dX = dY W  (K of this product = N <= 4), into padded rows; untouched padding stays NaN
        dY = torch.randn(M, N, generator=g)
        dYd = dY.to(backend.device)
        dX = torch.full((M, lda), float("nan"), device=backend.device)
        if K % 4 == 0:
            ops.gemm(dYd, N, 0, 1, N, Wd, K, 1, dX, lda, 0, K, None, M, K)
            assert_close_with_nonfinite(dX[:, :K], dY.double() @ W.double(), 1e-4, 1e-5, "skinny input gradient")
            assert torch.isnan(dX[:, K:]).all()
            ops.gemm(dYd, N, 0, 1, N, Wd, K, 1, dX, lda, 0, K, None, M, K, accumulate=True)
            assert_close_with_nonfinite(dX[:, :K], 2 * (dY.double() @ W.double()), 2e-4, 1e-5, "skinny input gradient +=")
        # weight / bias gradient (accumulating), dW stored [K, N]
        dW0 = torch.randn(K, N, generator=g)
        db0 = torch.randn(N, generator=g)
        dW, db = dW0.clone().to(backend.device), db0.clone().to(backend.device)
        ops.gemm_tn_acc(Ad, lda, 0, 1, K, dYd, N, dW, N, db, M, N)
        tol = 2e-4 if backend.name == "emu" else 3e-3
        assert_close_with_nonfinite(dW, dW0.double() + A[:, :K].double().t() @ dY.double(), tol, 1e-5, "skinny dW")
        assert_close_with_nonfinite(db, db0.double() + dY.double().sum(0), tol, 1e-5, "skinny db")
        dW2 = dW0.clone().to(backend.device)
        ops.gemm_tn_acc(Ad, lda, 0, 1, K, dYd, N, dW2, N, None, M, N)
        assert_close_with_nonfinite(dW2, dW0.double() + A[:, :K].double().t() @ dY.double(), tol, 1e-5, "skinny dW, no bias")
        # the same calls through the tile kernels agree (the dispatch is a schedule, not a different result)
        lib.tune("gemm_skinny", 0)
        Y3 = torch.empty(M, N, device=backend.device)
        ops.gemm(Ad, lda, 0, 1, K, Wd, 1, K, Y3, N, 0, N, bd, M, N)
        assert_close_with_nonfinite(Y3, ref, 2e-4, 1e-5, "tile kernel forward")
    finally:
        lib.tune("gemm_skinny", 1)


@pytest.mark.parametrize("M,segs,segk,N", [(700, 5, 66, 128), (520, 1, 64, 256), (390, 2, 32, 192), (1000, 1, 128, 136),
                                           (300, 1, 48, 128)])
def test_gemm_persistent_deferred_store_schedule(backend, M, segs, segk, N):
    """gemm_dbp_kernel (persistent workgroups; the epilogue of tile i rides in the main loop of tile i+1) forced onto
    three workgroups so each walks several tiles: NN with bias, NT with a segmented float2 / float4 output, the fused
    GRU-gate epilogue (bitwise against the one-tile kernel), K with fewer than four k-tiles (flush path), ragged edges."""
    lib = _lib.get_lib()
    lib.tune("gemm_small_tiles", 2)
    lib.tune("gemm_db", 2)
    try:
        g = torch.Generator().manual_seed(M * 3 + N)
        dev = backend.device
        K = segs * segk
        A = torch.randn(segs, M, segk, generator=g)
        W = torch.randn(K, N, generator=g)
        b = torch.randn(N, generator=g)
        Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
        dC = torch.randn(M, N, generator=g).to(dev)
        outs = {}
        for dbp in (0, 2):
            lib.tune("gemm_dbp", dbp)
            C = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(Ad, segk, M * segk, segs, segk, Wd, N, 1, C, N, 0, N, bd, M, N)
            Gs = torch.full((segs, M, segk), float("nan"), device=dev)
            ops.gemm(dC, N, 0, 1, N, Wd, 1, N, Gs, segk, M * segk, segk, None, M, K)
            O = N // 2
            if O % 4 == 0:
                H = torch.randn(M, O, generator=torch.Generator().manual_seed(5)).to(dev)
                zr = torch.full((M, N), float("nan"), device=dev)
                xhr = torch.zeros(M, O + 2, device=dev)
                ops.gemm_gru_zr(Ad, segk, M * segk, segs, segk, Wd, N, 1, bd, zr, H, xhr, 2)
            else:
                zr = xhr = torch.zeros(1, device=dev)
            outs[dbp] = (C, Gs, zr, xhr)
        ref = torch.cat([A[j] for j in range(segs)], dim=1).double() @ W.double() + b.double()
        assert_close_with_nonfinite(outs[2][0], ref, 1e-4, 1e-5, "persistent NN")
        refG = (dC.cpu().double() @ W.double().t()).view(M, segs, segk).permute(1, 0, 2)
        assert_close_with_nonfinite(outs[2][1], refG, 1e-4, 1e-5, "persistent NT, segmented output")
        for a, c in zip(outs[0], outs[2]):                    # same sums in the same order: bitwise
            assert torch.equal(a, c)
    finally:
        lib.tune("gemm_dbp", 1)
        lib.tune("gemm_small_tiles", 0)
        lib.tune("gemm_db", 1)


@pytest.mark.gpu
def test_aggregation_with_hubs_at_north_star_size():
    """The bench line's skewed graph (the band graph of the north-star shape + 20 rows of 2 000 more slots) on the product library:
    the window kernel keeps the 199 980 ordinary rows (bit for bit the reference's roundings), the hubs' slots ride with its tiles
    in 25 pieces per row and the combine launch adds them in a fixed order (against fp64, reproducible); the same rows from a
    workgroup per hub behind the window kernel (PGT_HUB_FOLD=0's form) and from the CSR kernels agree to rounding."""
    lib = _lib.get_lib()
    if lib.target != "gfx950":
        pytest.skip("product library only")
    dev = torch.device("cuda:0")
    n, F_ = 200_000, 64
    ei, ew = syn.hub_graph(n, 8, seed=0)
    gen = torch.Generator(device="cpu").manual_seed(12)
    X, T = torch.randn(n, F_, generator=gen).to(dev), torch.randn(n, F_, generator=gen).to(dev)
    outs = {}
    for fold in (True, False):
        old = ops.USE_HUB_FOLD
        ops.USE_HUB_FOLD = fold
        try:
            G = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
            csr = G.fwd_o
            Y = torch.full((n, F_), float("nan"), device=dev)
            ops.spmm(csr, X, Y)
            e = csr.ellw
            assert e and e.left_out == 20 and (e.tile_rows, e.width, e.config) == (392, 8, 1) and e.scale is not None
            assert (e.hub_col is not None) == fold and (not fold or e.hub_split == 25)
            Y2 = torch.full_like(Y, float("nan"))
            ops.spmm(csr, X, Y2)
            assert torch.equal(Y, Y2)
            Z = torch.full_like(Y, float("nan"))
            ops.spmm(csr, X, Z, T=T, alpha=2.0, beta=-1.0)
            outs[fold] = (Y, Z)
        finally:
            ops.USE_HUB_FOLD = old
    nnz = int(csr.rowptr[-1])
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
    ref = torch.zeros(n, F_, dtype=torch.float64, device=dev).index_add_(0, rows, X.double()[csr.col[:nnz].long()] * csr.val[:nnz].double()[:, None])
    hubs = csr.long_rows.long()
    ordinary = torch.ones(n, dtype=torch.bool, device=dev)
    ordinary[hubs] = False
    for fold, (Y, Z) in outs.items():
        assert_close_with_nonfinite(Y, ref.cpu(), 2e-5, 1e-5, f"fold={fold}: aggregation vs fp64")
        assert_close_with_nonfinite(Z, (2.0 * ref - T.double()).cpu(), 4e-5, 1e-5, f"fold={fold}: Chebyshev epilogue")
    assert torch.equal(outs[True][0][ordinary], outs[False][0][ordinary])                    # the same window kernel rows
    assert torch.equal(outs[True][0][ordinary].cpu(), source_scaled_reference(csr, X)[ordinary.cpu()])
    assert torch.allclose(outs[True][0][hubs], outs[False][0][hubs], atol=2e-5, rtol=1e-5)  # two orders of 2 000 adds
    Yc = torch.full((n, F_), float("nan"), device=dev)
    ops.spmm(csr, X, Yc, ellw=False)                                                         # CSR row tiles + a workgroup per hub
    assert torch.equal(Yc[hubs], outs[False][0][hubs])                                       # (the same hub kernel)
    assert_close_with_nonfinite(Yc, outs[True][0].cpu(), 2e-5, 1e-5, "CSR kernels vs window kernel + hubs")


@pytest.mark.gpu
def test_aggregation_at_north_star_size():
    """BASELINE.json's north-star shape (N = 200 000 nodes, F = 64, in-degree 8; 1.6 M edges) on the product library:
    the shipped kernel (ELLW layout, source-scale mode: P_o = A D_out^-1) against an fp64 gather / index_add of the same
    CSR (1e-5) and, bit for bit, against the reference's fp32 roundings (rounded `norm * x_j`, sequential adds);
    linearity, the Chebyshev epilogue; the per-slot-coefficient mode and the CSR row tiles (streaming vs plain stores,
    tile heights) agree with each other bit for bit and with the shipped kernel to 1e-5.  The uniform-random graph of the
    bench line (no locality: CSR row tiles) against fp64 as well."""
    lib = _lib.get_lib()
    if lib.target != "gfx950":
        pytest.skip("product library only")
    dev = torch.device("cuda:0")
    n, F_ = 200_000, 64
    gen = torch.Generator(device="cpu").manual_seed(11)
    X1, X2 = torch.randn(n, F_, generator=gen).to(dev), torch.randn(n, F_, generator=gen).to(dev)
    for kind, graph in (("local", syn.local_graph), ("uniform", syn.uniform_graph),
                        ("grid2d_hilbert", lambda n_, d_, seed: syn.grid2d_graph(447, "hilbert", seed)),
                        ("grid2d_rowmajor", lambda n_, d_, seed: syn.grid2d_graph(447, "rowmajor", seed)),
                        ("grid2d_shuffled", lambda n_, d_, seed: syn.grid2d_graph(447, "shuffled", seed))):
        ei, ew = graph(n, 8, seed=0)
        if kind.startswith("grid2d"):          # 447 x 447 = 199 809 nodes
            n = 447 * 447
            X1, X2 = X1[:n].contiguous(), X2[:n].contiguous()
        G = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
        csr = G.fwd_o
        nnz = int(csr.rowptr[-1])
        assert nnz == ei.shape[1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
        cols, vals = csr.col[:nnz].long(), csr.val[:nnz].double()

        def reference(X):
            return torch.zeros(n, F_, dtype=torch.float64, device=dev).index_add_(0, rows, X.double()[cols] * vals[:, None])

        Y1, Y2, Y3 = (torch.full((n, F_), float("nan"), device=dev) for _ in range(3))
        ops.spmm(csr, X1, Y1)
        e = csr.ellw
        if kind == "local":
            assert e is not None and e.scale is not None and e.vals is None and (e.tile_rows, e.width) == (392, 8)
            assert 0 < e.far < 1000                                                        # the wrap-around rows
        elif kind == "grid2d_hilbert":
            # not a band (83 % of the slots within +-32 rows) but compact tiles: the ring of a tile's patch rides in the table
            # of distinct outside rows (~85 per tile through ~230 slots)
            assert csr.halo == 0 and e is not None and e.scale is not None and e.far_csr == 0 and e.far > 50_000
            assert e.order is None
        elif kind in ("grid2d_rowmajor", "grid2d_shuffled"):
            # the caller's numbering hides the mesh's locality (bandwidth 447 / none at all): the library lays the operator
            # out in patches of its own (pgt_tile_order_host) and the kernel goes through the order — same answer, same roundings
            assert csr.halo == 0 and e is not None and e.order is not None and e.scale is not None and e.config == 3
            assert e.far_csr <= 0.002 * nnz and e.tile_rows == 392
        else:
            assert not e and csr.halo == 0
        ops.spmm(csr, X2, Y2)
        ref1 = reference(X1)
        assert_close_with_nonfinite(Y1, ref1.cpu(), 1e-5, 1e-5, f"{kind}: aggregation vs fp64")
        ops.spmm(csr, 0.5 * X1 - 2.0 * X2, Y3)
        assert torch.allclose(Y3, 0.5 * Y1 - 2.0 * Y2, atol=2e-5, rtol=1e-5)               # linearity
        ops.spmm(csr, X1, Y3, T=X2, alpha=2.0, beta=-1.0)                                  # 2 P X1 - X2
        assert_close_with_nonfinite(Y3, (2.0 * ref1 - X2.double()).cpu(), 2e-5, 1e-5, f"{kind}: Chebyshev epilogue")
        Yc = torch.full((n, F_), float("nan"), device=dev)
        ops.spmm(csr, X1, Yc, ellw=False)                                                  # CSR row tiles
        try:
            for key, value in (("spmm_tile_nt", 0), ("spmm_tile_nt", 2), ("spmm_tile_rows", 64)):
                lib.tune(key, value)
                Ys = torch.full((n, F_), float("nan"), device=dev)
                ops.spmm(csr, X1, Ys, ellw=False)
                assert torch.equal(Ys, Yc), (key, value)
        finally:
            lib.tune("spmm_tile_nt", 1)
            lib.tune("spmm_tile_rows", 32)
        if kind != "uniform":
            assert torch.equal(Y1.cpu(), source_scaled_reference(csr, X1))                 # the reference's roundings
            assert_close_with_nonfinite(Yc, Y1.cpu(), 1e-5, 1e-5, "CSR tiles vs ELLW")
        if kind == "local":
            # per-slot coefficient mode of the same layout: the CSR kernels' fmaf chain, bit for bit
            e.vals, e.scale = csr.val.new_zeros(e.n_tiles * e.tile_rows * e.width), None
            perslot = ops.Ellw.__new__(ops.Ellw)
            csr.val[0] = csr.val[0] * (1 + 2 ** -20)          # no longer a function of the source: forces per-slot mode
            csr.ellw = None
            ops._force_ellw(csr, 32)
            assert csr.ellw.vals is not None and csr.ellw.scale is None
            Yv, Yc2 = torch.full((n, F_), float("nan"), device=dev), torch.empty(n, F_, device=dev)
            ops.spmm(csr, X1, Yv)
            ops.spmm(csr, X1, Yc2, ellw=False)
            assert torch.equal(Yv, Yc2)
        del G, csr
        n = 200_000


# ------------------------------------------------------------------------------------------------ fuzzing the graph prep

def _fuzz_graph(draw, unique):
    from hypothesis import strategies as st
    n = draw(st.integers(1, 9))
    pairs = st.tuples(st.integers(0, n - 1), st.integers(0, n - 1))
    edges = draw(st.lists(pairs, min_size=0, max_size=24, unique=unique))
    weights = draw(st.lists(st.sampled_from([0.25, 0.5, 1.0, 1.5, 3.0]), min_size=len(edges), max_size=len(edges)))
    ei = torch.tensor(edges, dtype=torch.long).t().reshape(2, -1)
    return n, ei, torch.tensor(weights, dtype=torch.float32)


try:
    from hypothesis import HealthCheck, given, settings, strategies as hst

    _FUZZ = settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)

    @_FUZZ
    @given(data=hst.data())
    def test_fuzz_dconv_prep_on_arbitrary_edge_order(emu_backend, data):
        """Any order of unique, positively weighted edges (self-loops, isolated nodes, sinks, sources): the four
        operators equal the reference's scatter form (`dcrnn.py:277-290`), slots keep edge order, degrees bit-exact."""
        n, ei, ew = _fuzz_graph(data.draw, unique=True)
        g = ops.DConvGraph(emu_backend.t(ei), emu_backend.t(ew), n)
        norm_out, norm_in, rev = F.dconv_norms_scatter(ei, ew, n)
        Po, Pi = dense_from_edges(ei, norm_out, n), dense_from_edges(rev, norm_in, n)
        for name, csr, ref in (("fwd_o", g.fwd_o, Po), ("fwd_i", g.fwd_i, Pi), ("bwd_o", g.bwd_o, Po.t()),
                               ("bwd_i", g.bwd_i, Pi.t())):
            assert_close_with_nonfinite(csr_to_dense(csr, n), ref, 1e-6, 1e-6, name)
        assert torch.equal(g.deg_out.cpu()[:n], torch.zeros(n).scatter_add_(0, ei[0], ew))
        assert torch.equal(g.deg_in.cpu()[:n], torch.zeros(n).scatter_add_(0, ei[1], ew))
        rp, col = g.fwd_o.rowptr.cpu().numpy(), g.fwd_o.col.cpu().numpy()
        for i in range(n):
            assert np.array_equal(col[rp[i]:rp[i + 1]], ei[0][ei[1] == i].numpy())

    @_FUZZ
    @given(data=hst.data(), improved=hst.booleans(), loops=hst.booleans())
    def test_fuzz_gcn_prep_with_duplicates_and_self_loops(emu_backend, data, improved, loops):
        """gcn_norm on multigraphs: duplicate edges stay separate messages, existing self-loops keep their weight and
        move to the end, missing ones get the fill value."""
        n, ei, ew = _fuzz_graph(data.draw, unique=False)
        g = ops.SymGraph("gcn", emu_backend.t(ei), emu_backend.t(ew), n, improved=improved, add_self_loops=loops)
        ei2, w2 = P.gcn_norm(ei, ew, n, improved, loops)
        A = dense_from_edges(ei2, w2, n)
        assert_close_with_nonfinite(csr_to_dense(g.fwd, n), A, 1e-6, 1e-5, "gcn fwd")
        assert_close_with_nonfinite(csr_to_dense(g.bwd, n), A.t(), 1e-6, 1e-5, "gcn bwd")

    @_FUZZ
    @given(data=hst.data(), norm=hst.sampled_from(["sym", "rw", None]))
    def test_fuzz_cheb_prep_with_duplicates_and_self_loops(emu_backend, data, norm):
        """ChebConv.__norm__ (lambda_max = None -> PyG's default) on multigraphs with self-loops and isolated nodes."""
        n, ei, ew = _fuzz_graph(data.draw, unique=False)
        g = ops.SymGraph("cheb", emu_backend.t(ei), emu_backend.t(ew), n, normalization=norm, lambda_max=None, variant=0)
        ei2, w2 = F.cheb_norm(ei, ew, n, norm, None, torch.float32)
        L = dense_from_edges(ei2, w2, n)
        assert_close_with_nonfinite(csr_to_dense(g.fwd, n), L, 1e-6, 1e-5, "cheb fwd")
        assert_close_with_nonfinite(csr_to_dense(g.bwd, n), L.t(), 1e-6, 1e-5, "cheb bwd")
except ImportError:      # hypothesis is optional
    pass


try:
    from hypothesis import HealthCheck, given, settings, strategies as hst

    @settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(M=hst.integers(1, 280), segs=hst.integers(1, 4), segk=hst.integers(1, 40), N=hst.integers(1, 140),
           pad_a=hst.sampled_from([0, 1, 2, 4]), nt=hst.booleans(), bias=hst.booleans(), acc=hst.booleans(),
           mode=hst.sampled_from(["default", "big", "big_db", "persistent", "skinny"]),
           seg_out=hst.booleans())
    def test_fuzz_gemm_entry_point(emu_backend, M, segs, segk, N, pad_a, nt, bias, acc, mode, seg_out):
        """pgt_gemm_f32 over random shapes and layouts -- K-segmented A with padded rows, NN / NT weights, bias,
        accumulate, column-segmented output -- with every kernel family forced in turn (64 x 64 tiles, 128-wide tiles,
        the pipelined kernel, the persistent deferred-store kernel, the streaming kernels): whatever the dispatch
        picks must produce the same product."""
        lib = _lib.get_lib()
        knobs = {"default": {}, "big": {"gemm_small_tiles": 2, "gemm_db": 0}, "big_db": {"gemm_small_tiles": 2, "gemm_db": 2},
                 "persistent": {"gemm_small_tiles": 2, "gemm_db": 2, "gemm_dbp": 2}, "skinny": {"gemm_skinny": 2}}[mode]
        defaults = {"gemm_small_tiles": 0, "gemm_db": 1, "gemm_dbp": 1, "gemm_skinny": 1}
        for k, v in knobs.items():
            lib.tune(k, v)
        try:
            g = torch.Generator().manual_seed(M * 131 + N * 7 + segk)
            K = segs * segk
            lda = segk + pad_a
            A = torch.randn(segs, M, lda, generator=g)
            W = torch.randn(K, N, generator=g)
            b = torch.randn(N, generator=g) if bias else None
            dev = emu_backend.device
            Wd = (W.t().contiguous() if nt else W).to(dev)
            sbk, sbn = (1, K) if nt else (N, 1)
            ref = torch.cat([A[j, :, :segk] for j in range(segs)], dim=1).double() @ W.double()
            if bias:
                ref = ref + b.double()
            # output: plain [M, N], or column segments of width cs at stride M * cs (the diffusion-stack gradient layout)
            cs = 1
            if seg_out:
                divs = [d for d in range(1, N + 1) if N % d == 0]
                cs = divs[(M + segk) % len(divs)]
            if seg_out and cs != N:
                C0 = torch.randn(N // cs, M, cs, generator=g)
                C = C0.clone().to(dev)
                ops.gemm(A.to(dev), lda, M * lda, segs, segk, Wd, sbk, sbn, C, cs, M * cs, cs, None if b is None else b.to(dev),
                         M, N, accumulate=acc)
                got = C.cpu().permute(1, 0, 2).reshape(M, N)
                base = C0.permute(1, 0, 2).reshape(M, N)
            else:
                C0 = torch.randn(M, N, generator=g)
                C = C0.clone().to(dev)
                ops.gemm(A.to(dev), lda, M * lda, segs, segk, Wd, sbk, sbn, C, N, 0, N, None if b is None else b.to(dev),
                         M, N, accumulate=acc)
                got, base = C.cpu(), C0
            want = ref + base.double() if acc else ref
            assert_close_with_nonfinite(got, want, 2e-4, 2e-5, f"gemm {mode}")
        finally:
            for k in knobs:
                lib.tune(k, defaults[k])

    @settings(max_examples=100, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(M=hst.integers(1, 260), segs=hst.integers(1, 4), segk=hst.integers(1, 70), N=hst.integers(1, 140),
           pad_a=hst.sampled_from([0, 2, 4]), bias=hst.booleans(),
           mode=hst.sampled_from(["default", "big", "whole_k", "pipelined", "skinny"]))
    def test_fuzz_weight_gradient_entry_point(emu_backend, M, segs, segk, N, pad_a, bias, mode):
        """pgt_gemm_tn_acc_f32 (dW += A^T G, db += column sums) over random shapes with every schedule forced in turn."""
        lib = _lib.get_lib()
        knobs = {"default": {}, "big": {"gemm_small_tiles": 2, "gemm_tn_fullk": 0, "gemm_tn_pipe": 0},
                 "whole_k": {"gemm_tn_fullk": 2, "gemm_tn_pipe": 0}, "pipelined": {"gemm_tn_pipe": 2},
                 "skinny": {"gemm_skinny": 2}}[mode]
        defaults = {"gemm_small_tiles": 0, "gemm_tn_fullk": 1, "gemm_tn_pipe": 1, "gemm_skinny": 1}
        for k, v in knobs.items():
            lib.tune(k, v)
        try:
            g = torch.Generator().manual_seed(M * 17 + N * 3 + segk)
            lda = segk + pad_a
            A = torch.randn(segs, M, lda, generator=g)
            G = torch.randn(M, N, generator=g)
            dW0, db0 = torch.randn(segs * segk, N, generator=g), torch.randn(N, generator=g)
            dev = emu_backend.device
            dW, db = dW0.clone().to(dev), (db0.clone().to(dev) if bias else None)
            ops.gemm_tn_acc(A.to(dev), lda, M * lda, segs, segk, G.to(dev), N, dW, N, db, M, N)
            Acat = torch.cat([A[j, :, :segk] for j in range(segs)], dim=1).double()
            assert_close_with_nonfinite(dW, dW0.double() + Acat.t() @ G.double(), 3e-4, 2e-5, f"dW {mode}")
            if bias:
                assert_close_with_nonfinite(db, db0.double() + G.double().sum(0), 3e-4, 2e-5, f"db {mode}")
        finally:
            for k in knobs:
                lib.tune(k, defaults[k])

    @settings(max_examples=100, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(data=hst.data(), F_=hst.integers(1, 300), use_t=hst.booleans(), rows=hst.sampled_from([32, 64]),
           unroll=hst.sampled_from([4, 8]), halo=hst.sampled_from([0, 0, 32]), nt=hst.sampled_from([0, 2]),
           cap=hst.sampled_from([0, 8, 20]), scaled=hst.booleans(), cfg=hst.sampled_from([0, 1, 2]))
    def test_fuzz_aggregation_entry_point(emu_backend, data, F_, use_t, rows, unroll, halo, nt, cap, scaled, cfg):
        """pgt_spmm_csr_f32 / pgt_spmm_ellw_f32 on random CSR operators (empty rows, heavy rows, duplicates) for every
        feature width up to 300, with and without the `alpha A X + beta T` epilogue, across tile shapes, store flavours
        and the ELLW layout (F = 64 only; per-slot and source-scaled coefficients, sources anywhere: most slots then take
        the out-of-window path)."""
        lib = _lib.get_lib()
        n = data.draw(hst.integers(1, 90))
        deg = data.draw(hst.lists(hst.integers(0, 9), min_size=n, max_size=n))
        if data.draw(hst.booleans()):
            deg[data.draw(hst.integers(0, n - 1))] = 70                    # one heavy row
        rp = torch.tensor([0] + list(np.cumsum(deg)), dtype=torch.int32)
        nnz = int(rp[-1])
        gen = torch.Generator().manual_seed(n * 31 + F_)
        col = torch.randint(0, n, (nnz,), generator=gen, dtype=torch.int32)
        val = torch.randn(nnz, generator=gen)
        if scaled:
            val = torch.randn(n, generator=gen)[col.long()]
        dev = emu_backend.device
        csr = ops.Csr(n, max(nnz, 1), dev)
        csr.rowptr.copy_(rp)
        csr.col[:nnz].copy_(col)
        csr.val[:nnz].copy_(val)
        X, T = torch.randn(n, F_, generator=gen), torch.randn(n, F_, generator=gen)
        alpha, beta = (2.0, -1.0) if use_t else (0.5, 0.0)
        lib.tune("spmm_tile_rows", rows); lib.tune("spmm_unroll", unroll); lib.tune("spmm_tile_nt", nt)
        lib.tune("spmm_ellw_rows", cap)
        lib.tune("spmm_ellw_cfg", cfg)
        try:
            Y = torch.full((n, F_), float("nan")).to(dev)
            use_ellw = halo > 0 and F_ % 64 == 0 and nnz > 0
            if use_ellw:
                e = ops._force_ellw(csr, halo)
                assert (e is None) == (max(deg) > 32)
                assert e is None or (e.scale is not None) == scaled
            ops.spmm(csr, X.to(dev), Y, T=T.to(dev) if use_t else None, alpha=alpha, beta=beta, ellw=use_ellw)
        finally:
            lib.tune("spmm_tile_rows", 32); lib.tune("spmm_unroll", 8); lib.tune("spmm_tile_nt", 1)
            lib.tune("spmm_ellw_rows", 0)
            lib.tune("spmm_ellw_cfg", 0)
        rows_i = torch.repeat_interleave(torch.arange(n), torch.tensor(deg))
        ref = torch.zeros(n, F_, dtype=torch.float64).index_add_(0, rows_i, X.double()[col.long()] * val.double()[:, None])
        ref = alpha * ref + (beta * T.double() if use_t else 0.0)
        assert_close_with_nonfinite(Y, ref, 2e-5, 2e-5, "aggregation")

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(data=hst.data(), C=hst.integers(1, 70), K=hst.integers(2, 3), B=hst.integers(1, 3), pairs=hst.sampled_from([1, 2]),
           folded=hst.booleans())
    def test_fuzz_lds_resident_stack(emu_backend, data, C, K, B, pairs, folded):
        """pgt_dconv_stack_slab(_bwd)_f32 on random small digraphs (any column count incl. odd ones, one or two column
        pairs per lane) against the one-launch-per-hop path, forward and adjoint."""
        lib = _lib.get_lib()
        n = data.draw(hst.integers(1, 24))
        pairs_ = hst.tuples(hst.integers(0, n - 1), hst.integers(0, n - 1))
        edges = data.draw(hst.lists(pairs_, min_size=1, max_size=4 * n, unique=True))
        ei = torch.tensor(edges, dtype=torch.long).t().reshape(2, -1)
        ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(n + C)) + 0.2
        g = ops.DConvGraph(emu_backend.t(ei), emu_backend.t(ew), n)
        if not ops.slab_fits(g, C, K):
            return
        lib.tune("slab_pairs", pairs)
        try:
            S = 2 * K - 1
            gen = torch.Generator().manual_seed(C * 7 + n)
            X = torch.randn(B, n, C, generator=gen)
            TSn = torch.zeros(S, 1, n * B, C)
            TSn[0, 0] = X.permute(1, 0, 2).reshape(n * B, C)
            ops._stack_fwd(g, TSn, 0, K, n)
            TSb = torch.zeros(S, 1, B * n, C)
            TSb[0, 0] = X.reshape(B * n, C)
            ops._slab_fwd(g, TSb[0, 0], B * n * C, B, C, K)
            ref = TSn.view(S, n, B, C).permute(0, 2, 1, 3).reshape(S, B * n, C)
            assert_close_with_nonfinite(TSb.view(S, B * n, C), ref, 2e-6, 2e-6, "forward stack")
            Gsrc = torch.randn(S, B, n, C, generator=gen)
            Gn = Gsrc.permute(0, 2, 1, 3).reshape(S, n * B, C).contiguous()
            ops._stack_bwd(g, Gn, K, n, folded)
            Gb = Gsrc.reshape(S, B * n, C).contiguous()
            ops._slab_bwd(g, Gb[0], B * n * C, B, C, K, folded)
            refg = Gn[0].view(n, B, C).permute(1, 0, 2).reshape(B * n, C)
            assert_close_with_nonfinite(Gb[0], refg, 4e-6, 4e-6, "adjoint")
        finally:
            lib.tune("slab_pairs", 2)
except ImportError:      # hypothesis is optional
    pass


@pytest.mark.parametrize("M,segs,segk,N,mode", [(300, 5, 66, 128, "default"), (2500, 2, 10, 7, "default"),
                                                (700, 1, 64, 2, "skinny"), (4100, 3, 40, 64, "pipelined"),
                                                (900, 5, 66, 64, "whole_k"), (130, 1, 33, 70, "default")])
def test_deterministic_weight_gradient_matches_and_is_reproducible(backend, M, segs, segk, N, mode):
    """pgt_gemm_tn_det_f32 (per-slab partial sums stored, then added in slab order) against the fp64 product and the
    atomic entry point, for every weight-gradient schedule; two runs give bit-identical results."""
    from pytorch_geometric_temporal_amd import ops
    lib = _lib.get_lib()
    knobs = {"default": {}, "whole_k": {"gemm_tn_fullk": 2, "gemm_tn_pipe": 0}, "pipelined": {"gemm_tn_pipe": 2},
             "skinny": {"gemm_skinny": 2}}[mode]
    defaults = {"gemm_tn_fullk": 1, "gemm_tn_pipe": 1, "gemm_skinny": 1}
    if backend.name == "hip":
        M *= 40
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(segs, M, segk, generator=g).to(backend.device)
    G = torch.randn(M, N, generator=g).to(backend.device)
    dW0 = torch.randn(segs * segk, N, generator=g)
    db0 = torch.randn(N, generator=g)
    for k, v in knobs.items():
        lib.tune(k, v)
    keep_det = ops.DETERMINISTIC_WEIGHT_GRADIENTS
    try:
        outs = []
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = True
        for _ in range(2):
            dW, db = dW0.clone().to(backend.device), db0.clone().to(backend.device)
            ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dW, N, db, M, N)
            outs.append((dW.cpu(), db.cpu()))
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = False
        dWa, dba = dW0.clone().to(backend.device), db0.clone().to(backend.device)
        ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dWa, N, dba, M, N)
    finally:
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = keep_det
        for k in knobs:
            lib.tune(k, defaults[k])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # reproducible
    Acat = torch.cat([A[j].cpu() for j in range(segs)], dim=1).double()
    tol = 3e-4 if backend.name == "emu" else 3e-3
    assert_close_with_nonfinite(outs[0][0], dW0.double() + Acat.t() @ G.cpu().double(), tol, 2e-5, "dW deterministic")
    assert_close_with_nonfinite(outs[0][1], db0.double() + G.cpu().double().sum(0), tol, 2e-5, "db deterministic")
    assert_close_with_nonfinite(dWa, outs[0][0], tol, 2e-5, "atomic vs deterministic")


@pytest.mark.parametrize("F_", [64, 32, 6, 200])
def test_spmm_long_rows_are_split_across_the_workgroup(backend, F_):
    """pgt_spmm_csr_long_f32: rows longer than ops.LONG_ROW slots (hubs) are skipped by the row tiles and produced by one
    workgroup each (lane groups round-robin over the slots, LDS reduction in group order) — against the fp64 reference
    and the plain entry point, with the epilogue, empty rows next to hubs, and a width the tile kernels do not cover."""
    n = 500 if backend.name == "emu" else 20_000
    csr = banded_csr(n, 0, 9, 25, seed=11, device=backend.device, far_frac=0.1, heavy_row=min(n - 1, 123))
    rp = csr.rowptr.cpu()
    lens = rp[1:] - rp[:-1]
    long_rows = torch.nonzero(lens > ops.LONG_ROW).flatten().to(torch.int32)
    assert long_rows.numel() == 1 and int(lens.max()) == 300
    csr.long_rows = long_rows.to(backend.device)
    g = torch.Generator().manual_seed(F_)
    X, T = torch.randn(n, F_, generator=g).to(backend.device), torch.randn(n, F_, generator=g).to(backend.device)
    Y = torch.full((n, F_), float("nan"), device=backend.device)
    ops.spmm(csr, X, Y, ellw=False)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "long rows")
    ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0, ellw=False)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "long rows + epilogue")
    csr.long_rows = None
    Yp = torch.empty_like(Y)
    ops.spmm(csr, X, Yp, T=T, alpha=2.0, beta=-1.0, ellw=False)
    short = torch.ones(n, dtype=torch.bool)
    short[long_rows.long()] = False
    assert torch.equal(Y[short.to(Y.device)], Yp[short.to(Y.device)])        # every other row: the same kernel, bit for bit


def test_long_rows_are_listed_at_graph_preparation(backend):
    n = 5000
    ei, ew = syn.local_graph(n, 4, window=64, seed=0)
    rng = np.random.default_rng(1)
    hubs = np.array([17, 4000])
    src = np.concatenate([rng.choice(n, 400, replace=False) for _ in hubs])
    e2 = np.concatenate([ei, np.stack([src, np.repeat(hubs, 400)])], axis=1)
    key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    e2 = e2[:, key]
    g = ops.DConvGraph(backend.t(e2), None, n)
    assert g.fwd_o.long_rows is not None and sorted(g.fwd_o.long_rows.tolist()) == [17, 4000]
    assert g.fwd_o.max_len >= 400 and g.fwd_o.ellw is None and g.fwd_o.short_len <= 8
    assert g.bwd_o.long_rows is None                                       # the transposed operator has no hub ROWS
    X = torch.randn(n, 64).to(backend.device)
    Y = torch.empty_like(X)
    ops.spmm(g.fwd_o, X, Y)
    assert_close_with_nonfinite(Y, spmm_reference(g.fwd_o, X, None, 1.0, 0.0), 5e-5, 1e-5, "hub graph")


@pytest.mark.parametrize("fold", [True, False])
def test_spmm_ellw_leaves_hub_rows_out_of_the_layout(backend, fold):
    """A locality-ordered operator with a few rows wider than the layout's 32 slots (a hub of hundreds, a junction of 49) keeps the
    ELLW window kernel for its ordinary rows: the layout is planned for the longest ORDINARY row and leaves the others out (first
    slot = 0xFFFE: the kernel neither gathers nor stores them).  fold: the hubs' slots ride with the tiles as pieces (pgt_ellw.hub_*: one partial row per wavefront, a combine
    launch adds them in a fixed order) at F = 64; otherwise — and at other widths — pgt_spmm_csr_rows_f32 produces exactly those rows
    behind the window kernel.  Ordinary rows bit for bit the reference's roundings, hub rows against fp64 and reproducible, with the
    epilogue and an aliased T (the window kernel must not touch a hub's T row)."""
    n = 5000
    ei, ew = syn.local_graph(n, 4, window=64, seed=0)
    rng = np.random.default_rng(1)
    hubs, extra = np.array([17, 4000]), (400, 45)            # a hub, and a junction just over the layout's 32 slots
    src = np.concatenate([rng.choice(n, k, replace=False) for k in extra])
    e2 = np.concatenate([ei, np.stack([src, np.repeat(hubs, extra)])], axis=1)
    w2 = np.concatenate([ew, (0.5 + rng.random(src.size)).astype(np.float32)])
    key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    old = ops.USE_HUB_FOLD
    ops.USE_HUB_FOLD = fold
    try:
        g = ops.DConvGraph(backend.t(e2[:, key]), backend.t(w2[key]), n)
        csr = g.fwd_o
        assert csr.long_rows is not None and csr.max_len >= 400 and 4 <= csr.short_len <= 8 and csr.halo == 32
        assert csr.long_rows.tolist() == [17] and sorted(csr.left_rows.tolist()) == [17, 4000]   # > 128 slots / > the layout's 32
        e = ops.ellw_of(csr)
        assert e is not None and e.left_out == 2 and e.width == 8 and e.scale is not None        # P_o: source-scaled
        assert (e.hub_col is not None) == fold and (not fold or e.hub_split * 2 <= e.n_tiles)
        gen = torch.Generator().manual_seed(5)
        X, T = torch.randn(n, 64, generator=gen).to(backend.device), torch.randn(n, 64, generator=gen).to(backend.device)
        X[3, 5] = float("inf")                                   # (row 0 is what an unused piece entry reads; row 3 a real source)
        Y = torch.full((n, 64), float("nan"), device=backend.device)
        ops.spmm(csr, X, Y)
        assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "window kernel + hubs")
        ordinary = torch.ones(n, dtype=torch.bool)
        ordinary[torch.from_numpy(hubs)] = False
        assert torch.equal(Y.cpu()[ordinary].nan_to_num(7.0, 8.0, 9.0), source_scaled_reference(csr, X)[ordinary].nan_to_num(7.0, 8.0, 9.0))
        Y2 = torch.full_like(Y, float("nan"))
        ops.spmm(csr, X, Y2)
        assert torch.equal(Y.nan_to_num(7.0, 8.0, 9.0), Y2.nan_to_num(7.0, 8.0, 9.0))           # a fixed order of adds: reproducible
        if not fold:
            Yc = torch.empty_like(Y)
            ops.spmm(csr, X, Yc, ellw=False)                                   # CSR row tiles + the same long-row kernel
            assert torch.equal(Y.cpu()[17].nan_to_num(7.0, 8.0, 9.0), Yc.cpu()[17].nan_to_num(7.0, 8.0, 9.0))
            # (the junction's 49 slots run through the row tiles there, one sequential chain: another order of the same adds)
            assert_close_with_nonfinite(Y[4000:4001], Yc[4000:4001], 5e-5, 1e-5, "junction row")
        ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0)
        assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "epilogue")
        Tc = T.clone()
        ops.spmm(csr, X, Tc, T=Tc, alpha=2.0, beta=1.0)
        assert_close_with_nonfinite(Tc, spmm_reference(csr, X, T, 2.0, 1.0), 5e-5, 1e-5, "aliased T")
        # two column chunks: the pieces are for F = 64, the hub rows come from pgt_spmm_csr_rows_f32
        Xw = torch.randn(n, 128, generator=gen).to(backend.device)
        Yw = torch.full((n, 128), float("nan"), device=backend.device)
        ops.spmm(csr, Xw, Yw)
        assert_close_with_nonfinite(Yw, spmm_reference(csr, Xw, None, 1.0, 0.0), 5e-5, 1e-5, "F = 128")
        Xw = torch.randn(n, 320, generator=gen).to(backend.device)             # wider than the hub kernel: the CSR kernels take it
        Yw = torch.full((n, 320), float("nan"), device=backend.device)
        ops.spmm(csr, Xw, Yw)
        assert_close_with_nonfinite(Yw, spmm_reference(csr, Xw, None, 1.0, 0.0), 5e-5, 1e-5, "F = 320")
        # the transposed operator has hub COLUMNS, no hub rows: an ordinary layout (or none), nothing left out
        et = ops.ellw_of(g.bwd_o)
        assert g.bwd_o.long_rows is None and (et is None or et.left_out == 0)
    finally:
        ops.USE_HUB_FOLD = old
    # an operator nobody measured (no list of its wide rows): a row wider than the layout's 32 slots keeps it off the layout
    mid = banded_csr(n, 0, 6, 30, seed=6, device=backend.device, heavy_row=40)
    mid.long_rows, mid.left_rows, mid.short_len = None, None, -1
    rp = mid.rowptr.cpu()
    assert int((rp[1:] - rp[:-1]).max()) == 300
    assert ops._force_ellw(mid, 32) is None


def test_spmm_ellw_renumbered_layout_leaves_hub_rows_out(backend):
    """A graph in an arbitrary numbering (a mesh under a random permutation of its ids) WITH a hub and a junction: the library's
    patch order carries the ordinary rows (pgt_ellw.order), the two wide rows are left out of the layout by their layout positions
    and their slots ride with the tiles — the pieces and the combine launch name X / Y rows in the CALLER's numbering.  Ordinary rows
    bit for bit the reference's roundings, the wide rows against fp64, the epilogue, an aliased T, a width the pieces do not cover."""
    side = 70 if backend.name == "emu" else 150
    n = side * side
    rng = np.random.default_rng(21)
    ei, ew = syn.grid2d_graph(side, "rowmajor", seed=2)
    ei = rng.permutation(n)[ei]
    wide, extra = np.array([n // 3, n - 7]), (300, 40)
    src = np.concatenate([rng.choice(n, k, replace=False) for k in extra])
    e2 = np.concatenate([ei, np.stack([src, np.repeat(wide, extra)])], axis=1)
    w2 = np.concatenate([ew, (0.5 + rng.random(src.size)).astype(np.float32)])
    key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    G = ops.DConvGraph(backend.t(e2[:, key]), backend.t(w2[key]), n)
    csr = G.fwd_o
    assert csr.halo == 0 and sorted(csr.left_rows.tolist()) == sorted(wide.tolist()) and csr.short_len <= 9
    gen = torch.Generator().manual_seed(8)
    X, T = torch.randn(n, 64, generator=gen).to(backend.device), torch.randn(n, 64, generator=gen).to(backend.device)
    Y = torch.full((n, 64), float("nan"), device=backend.device)
    ops.spmm(csr, X, Y)
    e = csr.ellw
    assert e and e.order is not None and e.config == 3 and e.left_out == 2 and e.hub_col is not None
    assert sorted(e.order[e.csr.left_rows.long()].tolist()) == sorted(wide.tolist())        # layout positions <-> the caller's rows
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, None, 1.0, 0.0), 5e-5, 1e-5, "renumbered + wide rows")
    ordinary = torch.ones(n, dtype=torch.bool)
    ordinary[torch.from_numpy(wide)] = False
    assert torch.equal(Y.cpu()[ordinary], source_scaled_reference(csr, X)[ordinary])
    Y2 = torch.full_like(Y, float("nan"))
    ops.spmm(csr, X, Y2)
    assert torch.equal(Y, Y2)
    ops.spmm(csr, X, Y, T=T, alpha=2.0, beta=-1.0)
    assert_close_with_nonfinite(Y, spmm_reference(csr, X, T, 2.0, -1.0), 5e-5, 1e-5, "epilogue")
    Tc = T.clone()
    ops.spmm(csr, X, Tc, T=Tc, alpha=2.0, beta=1.0)
    assert_close_with_nonfinite(Tc, spmm_reference(csr, X, T, 2.0, 1.0), 5e-5, 1e-5, "aliased T")
    Xw = torch.randn(n, 128, generator=gen).to(backend.device)
    Yw = torch.full((n, 128), float("nan"), device=backend.device)
    ops.spmm(csr, Xw, Yw)
    assert_close_with_nonfinite(Yw, spmm_reference(csr, Xw, None, 1.0, 0.0), 5e-5, 1e-5, "F = 128")


@pytest.mark.gpu
def test_split_bf16_kernels_reproduce_their_output_bit_for_bit_at_benchmark_size():
    """Round 3's `scripts/bx_determinism_probe.py` as a test (the round-2 `vmcnt` race showed as ~4 differing rows per 200 000
    between two launches on the same operands): every kernel of csrc/gemm_bx.hip that has no atomics — the plain and the two
    gate-fused K-split products, both feature-gradient shapes, the deterministic weight gradient — launched four times on the
    same operands at M = 211 968 must give identical bits; sampled rows against fp64."""
    dev = torch.device("cuda:0")
    M, S, C, O, reps = 211968, 5, 66, 64, 4
    K = S * C
    g = torch.Generator().manual_seed(0)
    A = torch.randn(S, M, C, generator=g).to(dev)
    Wzr, bzr = (torch.randn(K, 2 * O, generator=g) / K ** 0.5).to(dev), torch.randn(2 * O, generator=g).to(dev)
    Wh, bh = (torch.randn(K, O, generator=g) / K ** 0.5).to(dev), torch.randn(O, generator=g).to(dev)
    H = torch.randn(M, O, generator=g).to(dev)

    def same(outs, what):
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), f"{what}: {int((o != outs[0]).any(dim=-1).sum())} rows differ between two launches"

    outs = []
    for _ in range(reps):
        Cc = torch.empty(M, 2 * O, device=dev)
        ops.gemm(A, C, M * C, S, C, Wzr, 2 * O, 1, Cc, 2 * O, 0, 2 * O, bzr, M, 2 * O)
        outs.append(Cc)
    same(outs, "330 -> 128 plain")
    idx = torch.arange(0, M, M // 256, device=dev)
    ref = torch.einsum("smc,sco->mo", A[:, idx].double(), Wzr.double().view(S, C, 2 * O)) + bzr.double()
    assert float((outs[0][idx].double() - ref).abs().max()) < 2e-5
    zrs, xhrs = [], []
    for _ in range(reps):
        zr, xhr = torch.empty(M, 2 * O, device=dev), torch.zeros(M, C, device=dev)
        ops.gemm_gru_zr(A, C, M * C, S, C, Wzr, 2 * O, 1, bzr, zr, H, xhr, 2)
        zrs.append(zr); xhrs.append(xhr)
    same(zrs, "330 -> 128 + z | r gates"); same(xhrs, "H * R")
    hts, sts = [], []
    for _ in range(reps):
        ht, o0, o1 = torch.empty(M, O, device=dev), torch.empty(M, O, device=dev), torch.zeros(M, C, device=dev)
        ops.gemm_gru_h(A, C, M * C, S, C, Wh, O, 1, bh, ht, zrs[0], H, o0, o1[:, 2:])
        hts.append(ht); sts.append(o0)
    same(hts, "330 -> 64 + candidate gate"); same(sts, "new state")
    for Kd in (128, 64):
        dP = torch.randn(M, Kd, generator=g).to(dev)
        WH = (torch.randn(320, Kd, generator=g) / Kd ** 0.5).to(dev)
        outs = []
        for _ in range(reps):
            G = torch.empty(5, M, O, device=dev)
            ops.gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, O, M * O, O, None, M, 320)
            outs.append(G)
        same(outs, f"{Kd} -> 320 feature gradient")
        ref = dP[idx].double() @ WH.double().t()
        assert float((outs[0].permute(1, 0, 2).reshape(M, 320)[idx].double() - ref).abs().max()) < 2e-5
    Gm = torch.randn(M, 2 * O, generator=g).to(dev)
    keep_det = ops.DETERMINISTIC_WEIGHT_GRADIENTS
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = True
    try:
        outs = []
        for _ in range(reps):
            dW, db = torch.zeros(K, 2 * O, device=dev), torch.zeros(2 * O, device=dev)
            ops.gemm_tn_acc(A, C, M * C, S, C, Gm, 2 * O, dW, 2 * O, db, M, 2 * O)
            outs.append(torch.cat([dW, db[None]], 0))
        same(outs, "deterministic weight gradient")
    finally:
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = keep_det


@pytest.mark.parametrize("cap,cus", [(0, 0), (48, 3)])
def test_spmm_ellw_renumbered_layout_answers_in_the_callers_numbering(backend, cap, cus):
    """An operator whose tiles are not compact in the caller's numbering (a mesh under a random permutation of its node
    ids; a banded graph shuffled the same way, with 30 % long-range edges so that tiles overflow their table of outside
    rows) laid out in a numbering of the library's own (pgt_tile_order_host + pgt_ellw.order): `spmm` still answers in the
    CALLER's numbering, per-slot mode bit for bit the CSR kernels' fmaf chain, source-scale mode bit for bit the
    reference's roundings (the rows keep their slots in the caller's order), with the epilogue, an aliased T, column
    chunks, strided operands; shapes the window kernel does not cover run the caller's CSR."""
    lib = _lib.get_lib()
    lib.tune("spmm_ellw_rows", cap)
    lib.tune("spmm_ellw_cus", cus)
    try:
        side = 30 if backend.name == "emu" else 150
        n = side * side
        rng = np.random.default_rng(11)
        ei, ew = syn.grid2d_graph(side, "rowmajor", seed=2)
        shuffle = rng.permutation(n)
        G = ops.DConvGraph(backend.t(shuffle[ei]), backend.t(ew), n)
        g = torch.Generator().manual_seed(5)
        X, T = torch.randn(n, 64, generator=g).to(backend.device), torch.randn(n, 64, generator=g).to(backend.device)
        for csr, scaled in ((G.fwd_o, True), (G.fwd_i, False)):
            e = ops._force_renumbered(csr)
            assert e is not None and e.order is not None and e.config == 3 and e.halo == 0
            assert (e.scale is not None) == scaled and (cap == 0 or e.tile_rows <= cap)
            assert torch.equal(torch.sort(e.order.cpu()).values, torch.arange(n, dtype=torch.int32))
            assert e.far_csr == 0 and (e.far > 0 or e.n_tiles == 1)          # a patch's ring fits its table
            Y = torch.full((n, 64), float("nan"), device=backend.device)
            ops.spmm(csr, X, Y)
            if e.far:
                # the outside rows' X rows handed to the kernel in the caller's numbering (pgt_ellw.far_src = order[far_col]):
                # two dependent loads like the window rows instead of three — the same rows, the same bits as through `order`
                fc = e.far_col.long()
                assert e.far_src is not None and torch.equal(e.far_src[fc >= 0].long(), e.order[fc[fc >= 0]].long())
                assert bool((e.far_src[fc < 0] == -1).all())
                kept, e.far_src = e.far_src, None
                Yo = torch.full_like(Y, float("nan"))
                ops.spmm(csr, X, Yo)
                e.far_src = kept
                assert torch.equal(Y, Yo)
            Yc = torch.empty_like(Y)
            ops.spmm(csr, X, Yc, ellw=False)
            if scaled:
                assert torch.equal(Y.cpu(), source_scaled_reference(csr, X))
                assert_close_with_nonfinite(Y, Yc, 1e-5, 1e-5, "source-scaled vs CSR kernels")
            else:
                assert torch.equal(Y, Yc)
            Ya, Yb = T.clone(), T.clone()
            ops.spmm(csr, X, Ya, T=Ya, alpha=0.5, beta=2.0)
            ops.spmm(csr, X, Yb, T=Yb, alpha=0.5, beta=2.0, ellw=False)
            if scaled:
                assert_close_with_nonfinite(Ya, Yb, 1e-5, 1e-5, "epilogue")
            else:
                assert torch.equal(Ya, Yb)
        # overflowing tables (slots served through the layout's CSR inside the gather), ragged rows, empty rows, wide operands
        n2 = 500 if backend.name == "emu" else 20_000
        base = banded_csr(n2, 0, 20, 12, seed=9, device=backend.device, far_frac=0.3)
        sh = torch.from_numpy(rng.permutation(n2).astype(np.int32))
        rp = base.rowptr.cpu().long()
        lens = (rp[1:] - rp[:-1])
        inv = torch.empty(n2, dtype=torch.long)
        inv[sh.long()] = torch.arange(n2)
        csr = ops.Csr.__new__(ops.Csr)
        csr.n_rows, csr.halo, csr.max_len, csr.nnz, csr.ellw, csr.long_rows = n2, 0, -1, -1, None, None
        new_lens = lens[inv]                                   # row i of the shuffled operator = row inv[i] of the banded one
        new_rp = torch.zeros(n2 + 1, dtype=torch.long)
        new_rp[1:] = torch.cumsum(new_lens, 0)
        take = torch.cat([torch.arange(rp[inv[i]], rp[inv[i] + 1]) for i in range(n2)]) if backend.name == "emu" else \
            (torch.repeat_interleave(rp[inv], new_lens) + torch.arange(int(new_rp[-1])) - torch.repeat_interleave(new_rp[:-1], new_lens))
        csr.rowptr = new_rp.to(torch.int32).to(backend.device)
        csr.col = sh[base.col.cpu().long()[take]].to(backend.device)
        csr.val = base.val.cpu()[take].to(backend.device)
        e = ops._force_renumbered(csr)
        assert e is not None and e.vals is not None and (e.far_csr > 0 or cap != 0)      # uncapped tiles overflow their table
        big = torch.randn(n2, 200, generator=g).to(backend.device)
        Xw, Tw = big[:, 4:196], torch.randn(n2, 192, generator=g).to(backend.device)
        Ya, Yb = torch.full((n2, 192), float("nan"), device=backend.device), torch.empty(n2, 192, device=backend.device)
        ops.spmm(csr, Xw, Ya, T=Tw, alpha=2.0, beta=-1.0)
        ops.spmm(csr, Xw, Yb, T=Tw, alpha=2.0, beta=-1.0, ellw=False)
        assert torch.equal(Ya, Yb)
        assert_close_with_nonfinite(Ya, spmm_reference(csr, Xw, Tw, 2.0, -1.0), 5e-5, 1e-5, "renumbered, wide")
        for F_ in (32, 66):                                   # not a multiple of 64 / unaligned: the caller's CSR
            Xf = torch.randn(n2, F_, generator=g).to(backend.device)
            Yf = torch.empty_like(Xf)
            ops.spmm(csr, Xf, Yf)
            assert_close_with_nonfinite(Yf, spmm_reference(csr, Xf, None, 1.0, 0.0), 5e-5, 1e-5, f"F={F_}")
        Xo, Yo = torch.randn(n2, 65, generator=g).to(backend.device)[:, 1:], torch.empty(n2, 64, device=backend.device)
        ops.spmm(csr, Xo, Yo)
        assert_close_with_nonfinite(Yo, spmm_reference(csr, Xo, None, 1.0, 0.0), 5e-5, 1e-5, "unaligned X")
    finally:
        lib.tune("spmm_ellw_rows", 0)
        lib.tune("spmm_ellw_cus", 0)


try:
    from hypothesis import HealthCheck, given, settings, strategies as hst

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(n=hst.integers(1, 90), deg=hst.integers(0, 12), components=hst.integers(1, 4), tile=hst.sampled_from([4, 8, 20, 0]),
           far=hst.floats(0.0, 0.5), scaled=hst.booleans(), epilogue=hst.booleans(), seed=hst.integers(0, 10_000))
    def test_fuzz_renumbered_layout(emu_backend, n, deg, components, tile, far, scaled, epilogue, seed):
        """pgt_tile_order_host + the renumbered window kernel over random operators: empty rows, self-loops, duplicate slots,
        several components, fewer rows than a tile, tiles forced down to four rows, long-range slots that overflow the table —
        the order is a permutation, the layout operator is the caller's in that numbering with every row's slots in the
        caller's order, and the product equals the CSR kernels' on the caller's numbering (per-slot mode bit for bit)."""
        lib = _lib.get_lib()
        rng = np.random.default_rng(seed)
        degs = rng.integers(0, deg + 1, size=n)
        rowptr = np.zeros(n + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum(degs)
        rows = np.repeat(np.arange(n), degs)
        comp = rows % components                                        # sources mostly in the row's own component, nearby
        col = (rows + components * rng.integers(-3, 4, size=rows.size)) % n
        col = np.where(col % components == comp, col, rows)             # (wrap-around may leave the component: self-loop then)
        jump = rng.random(rows.size) < far
        col[jump] = rng.integers(0, n, size=int(jump.sum()))
        val = (0.25 + rng.random(n)).astype(np.float32)[col] if scaled else rng.standard_normal(rows.size).astype(np.float32)
        shuffle = rng.permutation(n)                                    # the caller's numbering hides whatever structure there is
        inv = np.empty(n, dtype=np.int64)
        inv[shuffle] = np.arange(n)
        new_rows = shuffle[rows]
        order = np.argsort(new_rows, kind="stable")                     # keeps every row's slots in their original order
        csr = ops.Csr.__new__(ops.Csr)
        csr.n_rows, csr.halo, csr.max_len, csr.nnz, csr.ellw, csr.long_rows = n, 0, -1, -1, None, None
        rp2 = np.zeros(n + 1, dtype=np.int32)
        rp2[1:] = np.cumsum(np.bincount(new_rows, minlength=n))
        dev = emu_backend.device
        csr.rowptr = torch.from_numpy(rp2).to(dev)
        csr.col = torch.from_numpy(shuffle[col][order].astype(np.int32)).to(dev) if rows.size else torch.zeros(1, dtype=torch.int32)
        csr.val = torch.from_numpy(val[order]).to(dev) if rows.size else torch.zeros(1)
        lib.tune("spmm_ellw_rows", tile)
        lib.tune("spmm_ellw_cus", 2)
        try:
            e = ops._force_renumbered(csr)
            if rows.size == 0:
                assert e is None
                return
            assert e is not None and e.order is not None and e.config == 3
            o = e.order.cpu().numpy()
            assert np.array_equal(np.sort(o), np.arange(n))
            pos = np.empty(n, dtype=np.int64)
            pos[o] = np.arange(n)
            lay_rp, lay_col = e.csr.rowptr.cpu().numpy(), e.csr.col.cpu().numpy()
            for p in range(n):                                          # layout row p = the caller's row o[p], slots in the caller's order
                a, b = rp2[o[p]], rp2[o[p] + 1]
                assert lay_rp[p + 1] - lay_rp[p] == b - a
                assert np.array_equal(lay_col[lay_rp[p]:lay_rp[p + 1]], pos[csr.col.cpu().numpy()[a:b]])
            g = torch.Generator().manual_seed(seed)
            X, T = torch.randn(n, 64, generator=g).to(dev), torch.randn(n, 64, generator=g).to(dev)
            Ya, Yb = torch.full((n, 64), float("nan")), torch.empty(n, 64)
            kw = dict(T=T, alpha=1.5, beta=-0.5) if epilogue else {}
            ops.spmm(csr, X, Ya, **kw)
            ops.spmm(csr, X, Yb, ellw=False, **kw)
            if e.vals is not None:
                assert torch.equal(Ya, Yb)
            else:
                assert_close_with_nonfinite(Ya, Yb, 1e-5, 1e-5, "source-scale mode vs the CSR kernels")
                if not epilogue:
                    assert torch.equal(Ya, source_scaled_reference(csr, X))
        finally:
            lib.tune("spmm_ellw_rows", 0)
            lib.tune("spmm_ellw_cus", 0)
except ImportError:      # hypothesis is optional
    pass


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_kernel_equals_torch_adam(backend, wd):
    """dp.FlatAdam (pgt_adam_f32: one elementwise launch over the flat parameter buffer, the step count on the device) against
    torch.optim.Adam over the individual parameters: eight updates, with and without weight decay."""
    from pytorch_geometric_temporal_amd import dp
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3))
    mod = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3))
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(backend.device)
    opt_r = torch.optim.Adam(ref.parameters(), lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=wd)
    flat = dp.FlatParameters(mod.parameters())
    opt_m = flat.adam(lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=wd)
    x, y = torch.randn(12, 6), torch.randn(12, 3)
    for _ in range(8):
        opt_r.zero_grad()
        (ref(x) - y).pow(2).mean().backward()
        opt_r.step()
        opt_m.zero_grad()
        (mod(backend.t(x)) - backend.t(y)).pow(2).mean().backward()
        opt_m.step()
    assert float(opt_m.steps) == 8.0
    for pr, pm in zip(ref.parameters(), mod.parameters()):
        assert_close_with_nonfinite(pm.detach(), pr.detach(), 2e-6, 2e-5, "parameters after eight updates")
    opt_m.reset()
    assert float(opt_m.steps) == 0.0 and float(opt_m.exp_avg.abs().sum()) == 0.0
