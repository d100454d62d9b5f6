"""The call forms of the reference's own in-scope tests (test/recurrent_test.py:78-532, test/attention_test.py:105-367),
restated as a table so that they also run where /root/reference is absent — the `-m gpu` leg on the MI355X, at the reference
tests' full sizes.  tests/test_reference_suite.py runs the reference's files themselves (import swap) where they exist; this
file is what keeps a call form from going missing on the product library (round 4 lost
`conv(x, edge_index, attention, edge_weight, batch, lambda_max)` that way).

Every row: constructor arguments as the reference test passes them, the successive calls (each later call may feed the previous
result back as the state), and the result shape the reference asserts.  Values are pinned elsewhere (tests/golden); here the
assertions are the reference's (shapes) plus finiteness where the mock graph allows it (watts_strogatz `graph.edges()` lists
are one-directional: DConv's 1/deg_in is inf there by the reference's own arithmetic).
"""
import networkx as nx
import numpy as np
import pytest
import torch

from pytorch_geometric_temporal_amd.nn import attention as A
from pytorch_geometric_temporal_amd.nn import recurrent as R


def mock_graph(n, per_node, seed):
    g = nx.watts_strogatz_graph(n, per_node, 0.5, seed=seed)
    return torch.as_tensor(np.array(list(g.edges())).T, dtype=torch.long)


def uniform(rng, shape, lo=-1.0, hi=1.0):
    return torch.as_tensor(rng.uniform(lo, hi, shape), dtype=torch.float32)


# (name, class, constructor kwargs, X shape as a function of n, state arity, result shape as a function of n)
RECURRENT = [
    ("gconv_lstm", R.GConvLSTM, dict(in_channels=64, out_channels=16, K=2), lambda n: (n, 64), 2, lambda n: (n, 16)),
    ("gconv_gru", R.GConvGRU, dict(in_channels=64, out_channels=16, K=2), lambda n: (n, 64), 1, lambda n: (n, 16)),
    ("tgcn", R.TGCN, dict(in_channels=64, out_channels=16), lambda n: (n, 64), 1, lambda n: (n, 16)),
    ("a3tgcn", R.A3TGCN, dict(in_channels=64, out_channels=16, periods=7), lambda n: (n, 64, 7), 1, lambda n: (n, 16)),
    ("a3tgcn2", R.A3TGCN2, dict(in_channels=64, out_channels=16, periods=7, batch_size=8), lambda n: (8, n, 64, 7), 1, lambda n: (8, n, 16)),
    ("dcrnn_K2", R.DCRNN, dict(in_channels=64, out_channels=16, K=2), lambda n: (n, 64), 1, lambda n: (n, 16)),
    ("dcrnn_K3", R.DCRNN, dict(in_channels=64, out_channels=16, K=3), lambda n: (n, 64), 1, lambda n: (n, 16)),
    ("gc_lstm", R.GCLSTM, dict(in_channels=64, out_channels=16, K=2), lambda n: (n, 64), 2, lambda n: (n, 16)),
]


@pytest.mark.parametrize("name,cls,kw,xshape,arity,oshape", RECURRENT, ids=[r[0] for r in RECURRENT])
def test_recurrent_cells_accept_the_reference_tests_call_forms(backend, name, cls, kw, xshape, arity, oshape):
    """layer(X, edge_index) / layer(X, edge_index, edge_weight) / layer(X, edge_index, edge_weight, H[, C]) on the reference
    tests' 100-node watts_strogatz graph (recurrent_test.py:78-143, 174-315, 369-400)."""
    n = 100
    rng = np.random.default_rng(len(name))
    ei = backend.t(mock_graph(n, 10, seed=1))
    ew = backend.t(uniform(rng, (ei.shape[1],), 0.0, 1.0))
    X = backend.t(uniform(rng, xshape(n)))
    torch.manual_seed(0)
    layer = cls(**kw).to(backend.device)
    state = ()
    for args in ((X, ei), (X, ei, ew), None, None):           # the third and fourth calls pass the previous result as the state
        out = layer(*args) if args is not None else layer(X, ei, ew, *state)
        outs = out if arity == 2 else (out,)
        assert len(outs) == arity
        for o in outs:
            assert tuple(o.shape) == oshape(n)
            if not name.startswith("dcrnn"):
                assert bool(torch.isfinite(o).all()), name
        state = tuple(outs)


@pytest.mark.parametrize("cls,kw", [(R.EvolveGCNH, dict(in_channels=8, num_of_nodes=100)), (R.EvolveGCNO, dict(in_channels=8))],
                         ids=["evolve_gcn_h", "evolve_gcn_o"])
def test_evolvegcn_accepts_the_reference_tests_call_forms(backend, cls, kw):
    """X = layer(X, edge_index); X = layer(X, edge_index, edge_weight): the result feeds the next call (recurrent_test.py:485-532)."""
    rng = np.random.default_rng(5)
    ei = backend.t(mock_graph(100, 10, seed=2))
    ew = backend.t(uniform(rng, (ei.shape[1],), 0.0, 1.0))
    X = backend.t(uniform(rng, (100, 8)))
    torch.manual_seed(0)
    layer = cls(**kw).to(backend.device)
    X = layer(X, ei)
    assert tuple(X.shape) == (100, 8)
    X = layer(X, ei, ew)
    assert tuple(X.shape) == (100, 8) and bool(torch.isfinite(X).all())


def test_temporalconv_and_stconv_accept_the_reference_tests_call_forms(backend):
    """attention_test.py:105-176: TemporalConv(100, 10, 3) and STConv(300, 100, 8, 10, 3, K=2) on [10, 5, 300, 100]; the edge list
    is a 300-node, 15-per-node watts_strogatz graph with weights."""
    full = backend.name == "hip"
    B, T, n, C = (10, 5, 300, 100) if full else (3, 5, 40, 12)
    rng = np.random.default_rng(6)
    X = backend.t(uniform(rng, (B, T, n, C)))
    ei = backend.t(mock_graph(n, 15 if full else 6, seed=3))
    ew = backend.t(uniform(rng, (ei.shape[1],), 0.0, 1.0))
    torch.manual_seed(0)
    tc = A.TemporalConv(in_channels=C, out_channels=10, kernel_size=3).to(backend.device)
    H = tc(X)
    assert tuple(H.shape) == (B, T - 2, n, 10) and bool(torch.isfinite(H).all())
    st = A.STConv(num_nodes=n, in_channels=C, hidden_channels=8, out_channels=10, kernel_size=3, K=2).to(backend.device)
    H = st(X, ei, ew)
    assert tuple(H.shape) == (B, T - 4, n, 10) and bool(torch.isfinite(H).all())


def test_chebconvattention_accepts_the_reference_tests_call_forms(backend):
    """attention_test.py:183-217, all five calls: no weight / weight / weight + lambda_max=3.0 on the star graph, then the two-graph
    batch with `batch` alone and with `batch` + lambda_max = tensor([2., 3.])."""
    dev = backend.device
    torch.manual_seed(0)
    conv = A.ChebConvAttention(16, 32, K=3, normalization="sym").to(dev)
    assert repr(conv) == "ChebConvAttention(16, 32, K=3, normalization=sym)"
    ei = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]], device=dev)
    n, B = 4, 3
    ew = torch.rand(ei.size(1), device=dev)
    x = torch.randn(B, n, 16, device=dev)
    att = torch.softmax(torch.rand(B, n, n), dim=1).to(dev)
    for args, kw in (((x, ei, att), {}), ((x, ei, att, ew), {}), ((x, ei, att, ew), {"lambda_max": 3.0})):
        out = conv(*args, **kw)
        assert tuple(out.shape) == (B, n, 32) and bool(torch.isfinite(out).all())
    batch = torch.tensor([0, 0, 1, 1], device=dev)
    ei = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]], device=dev)
    ew = torch.rand(ei.size(1), device=dev)
    x = torch.randn(B, n, 16, device=dev)
    lambda_max = torch.tensor([2.0, 3.0], device=dev)
    out4 = conv(x, ei, att, ew, batch)
    out5 = conv(x, ei, att, ew, batch, lambda_max)
    assert tuple(out4.shape) == tuple(out5.shape) == (B, n, 32)
    assert bool(torch.isfinite(out4).all()) and bool(torch.isfinite(out5).all())
    # graph 0 has lambda 2 = the default: its rows agree with out4, graph 1's (lambda 3) do not
    assert torch.allclose(out4[:, :2], out5[:, :2], atol=1e-6) and not torch.allclose(out4[:, 2:], out5[:, 2:], atol=1e-4)


@pytest.mark.parametrize("model", ["astgcn", "mstgcn"])
def test_astgcn_and_mstgcn_accept_one_edge_list_or_one_per_time_step(backend, model):
    """attention_test.py:219-307 / 310-367: ASTGCN(2, 2, 3, 64, 64, 1, 12, 12, 307, normalization, bias) for normalization None,
    "sym" (no bias) and "rw", MSTGCN(2, 2, 3, 64, 64, 1, 12, 12); forward on a list of 12 edge lists and on a single one;
    result [32, 307, 12].  Full size on the GPU; on the CPU test double 19 nodes x 2 windows."""
    full = backend.name == "hip"
    n, B, T = (307, 32, 12) if full else (19, 2, 12)
    rng = np.random.default_rng(8)
    X = backend.t(uniform(rng, (B, n, 2, T)))
    seq = [backend.t(mock_graph(n, 15 if full else 6, seed=10 + t)) for t in range(T)]
    torch.manual_seed(0)
    if model == "astgcn":
        models = [A.ASTGCN(2, 2, 3, 64, 64, 1, 12, T, n, norm, bias).to(backend.device)
                  for norm, bias in ((None, True), ("sym", False), ("rw", True))]
    else:
        models = [A.MSTGCN(2, 2, 3, 64, 64, 1, 12, T).to(backend.device)]
    for m in models:
        with torch.no_grad():
            for edges in (seq, seq[0]):
                out = m(X, edges)
                assert tuple(out.shape) == (B, n, 12) and bool(torch.isfinite(out).all())
