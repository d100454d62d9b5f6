"""Edge cases of the path: empty edge lists, isolated nodes, single-node graphs, K = 1,
one period, non-contiguous inputs, in-place edited edge weights, wrong dtypes / devices — drop-in modules against the
fp32 CPU oracle, inf / nan placement included."""
import pytest
import torch

from conftest import assert_close_with_nonfinite
from oracle import functional as F
from pytorch_geometric_temporal_amd import _lib, ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.conv import ChebConv, GCNConv
from pytorch_geometric_temporal_amd.nn.recurrent import A3TGCN, DCRNN, TGCN, BatchedDCRNN

ATOL, RTOL = 1e-5, 1e-5


def _params(m, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_((torch.rand(p.shape, generator=g) - 0.5))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_empty_edge_list(backend):
    """E = 0: GCNConv degenerates to the self-loop (A_hat = I), ChebConv to its diagonal terms."""
    n = 9
    ei = torch.zeros(2, 0, dtype=torch.int64)
    X = torch.randn(n, 3)
    m = GCNConv(3, 5)
    p = _params(m)
    out = m.to(backend.device)(backend.t(X), backend.t(ei))
    assert_close_with_nonfinite(out, F.gcn_conv(X, ei, None, p["lin.weight"], p["bias"]), ATOL, RTOL, "gcn E=0")
    c = ChebConv(3, 4, 3)
    pc = _params(c, 1)
    outc = c.to(backend.device)(backend.t(X), backend.t(ei), lambda_max=2.0)
    refc = F.cheb_conv(X, ei, None, [pc[f"lins.{k}.weight"] for k in range(3)], pc["bias"], lambda_max=2.0)
    assert_close_with_nonfinite(outc, refc, ATOL, RTOL, "cheb E=0")


def test_isolated_and_sink_nodes_produce_the_reference_nonfinite_pattern(backend):
    """A source without in-edges makes 1/deg_in infinite (dcrnn.py:70-74): the same inf / nan entries must appear,
    in the single-graph cell and in the batched sequence."""
    n = 12
    ei = torch.tensor([[0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 2], [1, 2, 3, 4, 5, 6, 7, 8, 9, 5, 9]])   # 10, 11 isolated; 0 has no in-edge
    ew = torch.rand(ei.size(1)) + 0.5
    X, H = torch.randn(n, 2), torch.randn(n, 4)
    m = DCRNN(2, 4, 3)
    p = _params(m)
    out = m.to(backend.device)(backend.t(X), backend.t(ei), backend.t(ew), backend.t(H))
    ref = F.dcrnn_cell(X, ei, ew, H, p)
    assert not torch.isfinite(ref).all()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "dcrnn cell")
    mb = BatchedDCRNN(2, 4, 2)
    pb = _params(mb, 3)
    Xb = torch.randn(2, 3, n, 2)
    outb = mb.to(backend.device)(backend.t(Xb), backend.t(ei), backend.t(ew))
    assert_close_with_nonfinite(outb, F.batched_dcrnn(Xb, ei, ew, pb), ATOL, RTOL, "batched")


def test_single_node_k1_and_one_period(backend):
    ei = torch.tensor([[0], [0]])
    ew = torch.tensor([2.0])
    X, H = torch.randn(1, 3), torch.randn(1, 2)
    for K in (1, 2):
        m = DCRNN(3, 2, K)
        p = _params(m, K)
        out = m.to(backend.device)(backend.t(X), backend.t(ei), backend.t(ew), backend.t(H))
        assert_close_with_nonfinite(out, F.dcrnn_cell(X, ei, ew, H, p), ATOL, RTOL, f"N=1 K={K}")
    t = TGCN(3, 2)
    pt = _params(t, 5)
    out = t.to(backend.device)(backend.t(X), backend.t(ei), backend.t(ew), backend.t(H))
    assert_close_with_nonfinite(out, F.tgcn_cell(X, ei, ew, H, pt), ATOL, RTOL, "tgcn N=1")
    ei2, ew2 = (torch.from_numpy(a) for a in syn.sensor_graph(10, 40, seed=1))
    a = A3TGCN(3, 2, periods=1)
    pa = _params(a, 6)
    Xp = torch.randn(10, 3, 1)
    out = a.to(backend.device)(backend.t(Xp), backend.t(ei2), backend.t(ew2))
    assert_close_with_nonfinite(out, F.a3tgcn(Xp, ei2, ew2, None, pa), ATOL, RTOL, "a3tgcn periods=1")


def test_non_contiguous_inputs_and_inplace_edited_weights(backend):
    n = 15
    ei_np, ew_np = syn.sensor_graph(n, 70, seed=2, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    m = TGCN(4, 3)
    p = _params(m, 7)
    m = m.to(backend.device)
    Xt = torch.randn(4, n)                          # feature-major storage, handed over as a transposed view
    eid, ewd = backend.t(ei), backend.t(ew)
    out = m(backend.t(Xt).t(), eid, ewd)
    assert_close_with_nonfinite(out, F.tgcn_cell(Xt.t(), ei, ew, None, p), ATOL, RTOL, "transposed view")
    with torch.no_grad():
        ewd.mul_(3.0)                               # same storage, new values: the prepared graph must be rebuilt
    out2 = m(backend.t(Xt).t(), eid, ewd)
    assert_close_with_nonfinite(out2, F.tgcn_cell(Xt.t(), ei, ew * 3.0, None, p), ATOL, RTOL, "after in-place edit")
    # a strided edge_index (every second column of a wider tensor)
    wide = torch.zeros(2, 2 * ei.size(1), dtype=torch.int64)
    wide[:, ::2] = ei
    out3 = m(backend.t(Xt).t(), backend.t(wide)[:, ::2], backend.t(ew))
    assert_close_with_nonfinite(out3, F.tgcn_cell(Xt.t(), ei, ew, None, p), ATOL, RTOL, "strided edge_index")


def test_wrong_dtype_device_and_index_range_fail_loudly(backend):
    n = 6
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    m = DCRNN(2, 2, 2).to(backend.device)
    X = backend.t(torch.randn(n, 2))
    with pytest.raises(TypeError):
        m(X.double(), backend.t(ei))
    with pytest.raises((TypeError, _lib.PgtError)):
        m(X, backend.t(ei).int())
    with pytest.raises(IndexError, match="outside"):
        m(X, backend.t(torch.tensor([[0, 7], [1, 2]])))
    with pytest.raises(ValueError):
        ops.dconv_graph(backend.t(torch.zeros(3, 4, dtype=torch.int64)), None, n)
    if backend.name == "hip":
        with pytest.raises(_lib.PgtError, match="no CPU fallback"):
            m(torch.randn(n, 2), backend.t(ei))


@pytest.mark.parametrize("K", [1, 2, 3])
def test_operands_of_the_wrong_width_are_refused_before_any_launch(backend, K):
    """The one-workgroup cell / sequence kernels read the stacked weights and the state by the sizes they are TOLD (the C entry
    points only null-check pointers): X or H of another width than the module's must raise in Python, on every path — the
    reference fails in its first matmul.  (Round 4's advisor: DCRNN(4, 8, K=2) on X [N, 6] returned garbage.)"""
    from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2
    n = 12
    ei_np, ew_np = syn.sensor_graph(n, 50, seed=2, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    cell = DCRNN(4, 8, K).to(backend.device)
    good_x, good_h = backend.t(torch.randn(n, 4)), backend.t(torch.randn(n, 8))
    assert cell(good_x, ei, ew, good_h).shape == (n, 8)
    cases = [(torch.randn(n, 6), None), (torch.randn(n, 2), None), (torch.randn(n, 4), torch.randn(n, 5)),
             (torch.randn(n, 4), torch.randn(n - 3, 8))]
    if K > 1:                                                    # (K = 1 never touches the graph: fewer rows than nodes is legal there,
        cases.append((torch.randn(n - 1, 4), None))              #  in the reference too, dcrnn.py:79-82)
    for x, h in cases:
        with pytest.raises((ValueError, IndexError)):
            cell(backend.t(x), ei, ew, None if h is None else backend.t(h))
    seq = BatchedDCRNN(2, 2, K=max(K, 2)).to(backend.device)           # hidden 2: the whole sequence in one workgroup
    assert seq(backend.t(torch.randn(3, 4, n, 2)), ei, ew).shape == (3, 4, n, 2)
    for x in (torch.randn(3, 4, n, 3), torch.randn(3, 4, n, 1)):       # (more nodes than the edge list names are isolated nodes: legal)
        with pytest.raises((ValueError, IndexError)):
            seq(backend.t(x), ei, ew)
    for m, x, h in ((TGCN(4, 32).to(backend.device), torch.randn(n, 6), None),          # hidden 32: the one-launch cell
                    (TGCN(4, 32).to(backend.device), torch.randn(n, 4), torch.randn(n, 16)),
                    (TGCN2(4, 32, 1).to(backend.device), torch.randn(2, n, 3), None),
                    (TGCN(4, 8).to(backend.device), torch.randn(n, 5), None)):
        with pytest.raises((ValueError, IndexError)):
            m(backend.t(x), ei, ew, None if h is None else backend.t(h))
