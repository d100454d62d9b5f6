"""DCRNN family through the drop-in modules: forward parity against the fixtures produced by
the reference's own module files, forward + backward parity against the CPU oracle, API / state_dict compatibility."""
import pytest
import torch

from conftest import assert_close_with_nonfinite, golden_names, load_golden
from oracle import functional as F
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN, BatchedDCRNN, BatchedDConv, DConv

ATOL, RTOL = 1e-5, 1e-5   # north_star: "results match the reference PyG/CPU forward to 1e-5 fp32"


def _load(module, params, device):
    module.load_state_dict(params, strict=True)     # reference checkpoints load unchanged
    return module.to(device)


@pytest.mark.parametrize("name", golden_names("dcrnn_"))
def test_dcrnn_forward_matches_reference_fixture(backend, name):
    g = load_golden(name)
    K = int(g["meta"]["K"])
    X, H0, ei, ew = (backend.t(g["in"][k]) for k in ("X", "H0", "edge_index", "edge_weight"))
    m = _load(DCRNN(X.size(1), H0.size(1), K), g["param"], backend.device)
    with torch.no_grad():
        assert_close_with_nonfinite(m(X, ei), g["out"]["H_noweight"], ATOL, RTOL, "no weight")
        assert_close_with_nonfinite(m(X, ei, ew), g["out"]["H_weight"], ATOL, RTOL, "weight")
        assert_close_with_nonfinite(m(X, ei, ew, H0), g["out"]["H_weight_hidden"], ATOL, RTOL, "weight+hidden")


def test_dconv_and_batched_dconv_match_reference_fixture(backend):
    g = load_golden("dconv_sensor_asym_K3")
    X, ei, ew = (backend.t(g["in"][k]) for k in ("X", "edge_index", "edge_weight"))
    for cls, key in ((DConv, "H"), (BatchedDConv, "H_batched")):
        m = _load(cls(6, 10, 3), g["param"], backend.device)
        with torch.no_grad():
            assert_close_with_nonfinite(m(X, ei, ew), g["out"][key], ATOL, RTOL, cls.__name__)


@pytest.mark.parametrize("name", golden_names("batched_dcrnn_"))
def test_batched_dcrnn_forward_matches_reference_fixture(backend, name):
    if backend.name == "emu" and "metrla" in name:
        pytest.skip("METR-LA-sized fixture runs on the GPU leg only (the CPU test double is a fiber emulator)")
    g = load_golden(name)
    X, ei, ew = (backend.t(g["in"][k]) for k in ("X", "edge_index", "edge_weight"))
    O = g["param"]["conv_x_z.weight"].shape[3]
    m = _load(BatchedDCRNN(X.size(3), O, int(g["meta"]["K"])), g["param"], backend.device)
    with torch.no_grad():
        out = m(X, ei, ew)
    assert out.shape == g["out"]["out"].shape
    assert_close_with_nonfinite(out, g["out"]["out"], ATOL, RTOL, name)


@pytest.mark.parametrize("K", [1, 2, 3])
def test_dcrnn_backward_matches_oracle_autograd(backend, K):
    torch.manual_seed(K)
    n, fin, O = 24, 3, 5
    ei_np, ew_np = syn.sensor_graph(n, 150, seed=K, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    m = DCRNN(fin, O, K)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.5, 0.5)
    params64 = {k: v.detach().double().requires_grad_() for k, v in m.state_dict().items()}
    m = m.to(backend.device)
    X, H = torch.randn(n, fin), torch.randn(n, O)
    w = torch.randn(n, O)
    Xd, Hd = backend.t(X).requires_grad_(), backend.t(H).requires_grad_()
    out = m(Xd, backend.t(ei), backend.t(ew), Hd)
    (out * backend.t(w)).sum().backward()
    X64, H64 = X.double().requires_grad_(), H.double().requires_grad_()
    ref = F.dcrnn_cell(X64, ei, ew.double(), H64, params64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward")
    assert_close_with_nonfinite(Xd.grad, X64.grad, 5e-5, 1e-4, "dX")
    assert_close_with_nonfinite(Hd.grad, H64.grad, 5e-5, 1e-4, "dH")
    for name, p in m.named_parameters():
        assert_close_with_nonfinite(p.grad, params64[name].grad, 1e-4, 1e-4, name)


@pytest.mark.parametrize("x_grad,O,bx", [(True, 4, 1), (False, 4, 1), (False, 64, 1), (False, 64, 2), (True, 64, 2),
                                              (False, 32, 2)])
def test_batched_dcrnn_backward_matches_oracle_autograd(backend, x_grad, O, bx):
    """x_grad=False: the input is data; the backward then computes only the hidden-state columns of the stack gradient
    (O = 64: 320 columns, split into a 256-wide and a 64-wide feature-gradient GEMM).  bx = 2: every dense product of the
    step on the split-bf16 kernels (csrc/gemm_bx.hip: gates with fused epilogues, feature gradients, weight gradients),
    which otherwise start at 8 192 rows — forward and all gradients against the fp64 oracle autograd."""
    from pytorch_geometric_temporal_amd import _lib, ops
    _lib.get_lib().tune("gemm_bx", bx)
    min_rows = ops.ONE_FEATURE_GRADIENT_MIN_ROWS
    if bx == 2:
        ops.ONE_FEATURE_GRADIENT_MIN_ROWS = 0      # ... including the single 320-column feature-gradient product
    try:
        _batched_dcrnn_backward_case(backend, x_grad, O)
    finally:
        _lib.get_lib().tune("gemm_bx", 1)
        ops.ONE_FEATURE_GRADIENT_MIN_ROWS = min_rows


def test_batched_dcrnn_zero_in_degree_graph_through_the_split_bf16_kernels(backend):
    """The reference's own mock graph (test/recurrent_test.py:16-23: watts_strogatz_graph(100, 10, 0.5).edges(), i.e. one
    direction per edge) leaves nodes without in-edges, so 1 / deg_in is infinite (dcrnn.py:279-290) and DConv's terms carry
    inf / nan (dcrnn.py:292-325).  With B >= 82 graphs the products have >= 8 192 rows and run on the split-bf16 kernels,
    whose piece arithmetic would turn an infinite operand into a nan ROW: the per-tile exact-fp32 redo (bx_exact_tile)
    must put every inf / nan where the reference's fp32 arithmetic (the oracle) puts it.  Also: forward and every gradient
    equal, placement included, to the same step on the exact-fp32 kernels."""
    from pytorch_geometric_temporal_amd import _lib, ops
    lib = _lib.get_lib()
    hip = backend.name == "hip"
    n, B, T, O, K = 100, (82 if hip else 2), (3 if hip else 2), 64, 3
    ei = torch.from_numpy(syn.watts_strogatz_directed(n, 10, 0.5, seed=0))
    ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(1)) + 0.5
    torch.manual_seed(3)
    m = BatchedDCRNN(2, O, K)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.3, 0.3)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(backend.device)
    X = torch.randn(B, T, n, 2)
    w = torch.randn(B, T, n, O)
    ref = F.batched_dcrnn(X, ei, ew, params)
    assert torch.isinf(ref).any() or torch.isnan(ref).any()
    assert torch.isfinite(ref).any()
    res = {}
    min_rows = ops.ONE_FEATURE_GRADIENT_MIN_ROWS
    try:
        for bx in ((1 if hip else 2), 0):
            lib.tune("gemm_bx", bx)
            ops.ONE_FEATURE_GRADIENT_MIN_ROWS = 0 if bx else min_rows
            m.zero_grad()
            out = m(backend.t(X), backend.t(ei), backend.t(ew))
            (out * backend.t(w)).sum().backward()
            res[bx] = (out.detach().clone(), [p.grad.clone() for p in m.parameters()])
    finally:
        lib.tune("gemm_bx", 1)
        ops.ONE_FEATURE_GRADIENT_MIN_ROWS = min_rows
    fast, exact = res[1 if hip else 2], res[0]
    assert_close_with_nonfinite(exact[0], ref, ATOL, RTOL, "exact-fp32 kernels vs oracle")
    assert_close_with_nonfinite(fast[0], ref, ATOL, RTOL, "split-bf16 kernels vs oracle")
    for a, b, (name, _) in zip(fast[1], exact[1], m.named_parameters()):
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.isinf(a), torch.isinf(b)), name


def _batched_dcrnn_backward_case(backend, x_grad, O):
    torch.manual_seed(0)
    B, T, n, fin, K = 2, 3, 18, 2, 3
    ei_np, ew_np = syn.sensor_graph(n, 110, seed=9, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    m = BatchedDCRNN(fin, O, K)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.5, 0.5)
    params64 = {k: v.detach().double().requires_grad_() for k, v in m.state_dict().items()}
    m = m.to(backend.device)
    X = torch.randn(B, T, n, fin)
    w = torch.randn(B, T, n, O)
    Xd = backend.t(X).requires_grad_(x_grad)
    out = m(Xd, backend.t(ei), backend.t(ew))
    (out * backend.t(w)).sum().backward()
    X64 = X.double().requires_grad_()
    ref = F.batched_dcrnn(X64, ei, ew.double(), params64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward")
    if x_grad:
        assert_close_with_nonfinite(Xd.grad, X64.grad, 5e-5, 1e-4, "dX")
    for name, p in m.named_parameters():
        assert_close_with_nonfinite(p.grad, params64[name].grad, 2e-4 if O >= 32 else 1e-4, 1e-4, name)


@pytest.mark.parametrize("n_nodes", [18, 330])
def test_batched_dcrnn_output_is_the_references_contiguous_layout(backend, n_nodes, request):
    """BatchedDCRNN returns the reference's contiguous [B, T, N, O] tensor (torch.stack(outputs, dim=1), dcrnn.py:463-475);
    every step's candidate-gate epilogue stores H_t straight into out[:, t] through a two-level row map and the backward
    reads the gradient in that layout (both row layouts: LDS-resident batch-major stack for the small graph, node-major
    rows for the large one).  Values and every gradient against the plain time-major sequence ([T, M, O] states + an
    explicit transposition) — the same kernels without the row map."""
    from pytorch_geometric_temporal_amd.nn.conv import Linear
    from pytorch_geometric_temporal_amd.nn.recurrent.dcrnn import _cell_weights
    from pytorch_geometric_temporal_amd import ops
    torch.manual_seed(1)
    B, T, fin, K = 3, 2, 2, 2
    O = 4 if n_nodes < 100 else 62
    ei_np, ew_np = syn.sensor_graph(n_nodes, 5 * n_nodes, seed=4, symmetric=False)
    ei, ew = backend.t(ei_np), backend.t(ew_np)
    g = ops.dconv_graph(ei, ew, n_nodes, strict_dense=False)
    bm = n_nodes < 100
    if not bm:        # the larger graph on the node-major path (it would fit the LDS column by column)
        request.getfixturevalue("monkeypatch").setattr(ops, "slab_fits", lambda *a, **k: False)
    X = torch.randn(B, T, n_nodes, fin)
    w = backend.t(torch.randn(B, T, n_nodes, 3))
    res = []
    for direct in (True, False):
        torch.manual_seed(2)
        rnn, head = BatchedDCRNN(fin, O, K).to(backend.device), Linear(O, 3).to(backend.device)
        Xd = backend.t(X).requires_grad_()
        if direct:
            h = rnn(Xd, ei, ew)
            assert h.shape == (B, T, n_nodes, O) and h.is_contiguous()
        else:
            Wzr, bzr, Wh, bh = _cell_weights(rnn.conv_x_z, rnn.conv_x_r, rnn.conv_x_h)
            H0 = torch.zeros(B * n_nodes, O, device=backend.device)
            if bm:
                Xs = Xd.permute(1, 0, 2, 3).reshape(T, B * n_nodes, fin)
                h = ops.DCRNNSeqFunction.apply(Xs, H0, Wzr, bzr, Wh, bh, g, K, B, True)
                h = h.view(T, B, n_nodes, O).permute(1, 0, 2, 3)
            else:
                Xs = Xd.permute(1, 2, 0, 3).reshape(T, n_nodes * B, fin)
                h = ops.DCRNNSeqFunction.apply(Xs, H0, Wzr, bzr, Wh, bh, g, K, B)
                h = h.view(T, n_nodes, B, O).permute(2, 0, 1, 3)
        y = head(torch.relu(h))
        assert y.shape == (B, T, n_nodes, 3)
        (y * w).sum().backward()
        res.append((h.detach().clone().contiguous(), y.detach().clone().contiguous(), Xd.grad.clone(),
                    [p.grad.clone() for p in list(rnn.parameters()) + list(head.parameters())]))
    (h0, y0, gx0, gp0), (h1, y1, gx1, gp1) = res
    assert torch.equal(h0, h1) and torch.equal(y0, y1)
    assert_close_with_nonfinite(gx1, gx0, 1e-6, 1e-6, "dX")
    for a, b in zip(gp0, gp1):
        assert_close_with_nonfinite(b, a, 1e-5, 1e-5, "parameter gradient")   # atomics: summation order differs


def test_reference_api_surface(backend):
    # constructor / attribute / parameter-name surface of dcrnn.py:21-37,128-160,343-361
    m = DCRNN(in_channels=4, out_channels=8, K=2, bias=True)
    assert (m.in_channels, m.out_channels, m.K, m.bias) == (4, 8, 2, True)
    assert sorted(k for k, _ in m.named_parameters()) == sorted(
        f"conv_x_{g}.{p}" for g in "zrh" for p in ("weight", "bias"))
    assert m.conv_x_z.weight.shape == (2, 2, 12, 8) and m.conv_x_z.bias.shape == (8,)
    assert float(m.conv_x_h.bias.abs().sum()) == 0.0          # zeros init (dcrnn.py:37)
    with pytest.raises(AssertionError):
        DConv(4, 8, 0)                                         # assert K > 0 (dcrnn.py:23)
    b = BatchedDCRNN(2, 2, 3).to(backend.device)
    ei_np, ew_np = syn.sensor_graph(12, 60, seed=1)
    out = b(torch.zeros(2, 3, 12, 2, device=backend.device), backend.t(ei_np), backend.t(ew_np))
    assert out.shape == (2, 3, 12, 2)
    with pytest.raises(ValueError):
        b(torch.zeros(2, 3, 12, 5, device=backend.device), backend.t(ei_np), backend.t(ew_np))


def test_graph_prep_is_cached_by_identity_not_by_value(backend):
    from pytorch_geometric_temporal_amd import ops
    ei_np, ew_np = syn.sensor_graph(12, 60, seed=2)
    ei, ew = backend.t(ei_np), backend.t(ew_np)
    g1 = ops.dconv_graph(ei, ew, 12)
    assert ops.dconv_graph(ei, ew, 12) is g1
    ew.mul_(2.0)                                               # in-place edit bumps the version counter
    assert ops.dconv_graph(ei, ew, 12) is not g1


def test_graph_cache_and_writes_the_version_counter_does_not_record(backend):
    """`edge_weight.data.mul_()` leaves `_version` alone, so the identity-keyed cache keeps the old operators (documented on
    ops._GraphCache; the reference compares VALUES with torch.equal, dcrnn.py:446-447, one host sync each).  `forget` after such a
    write, or verify mode (PGT_GRAPH_VERIFY=1: a value checksum per hit, one sync per forward), brings the new values in."""
    from pytorch_geometric_temporal_amd import ops
    ei_np, ew_np = syn.sensor_graph(12, 60, seed=2)
    ei, ew = backend.t(ei_np), backend.t(ew_np)
    m = DCRNN(3, 5, 2).to(backend.device)
    X = backend.t(torch.randn(12, 3))
    with torch.no_grad():
        before = m(X, ei, ew)
        g1 = ops.dconv_graph(ei, ew, 12, strict_dense=True)
        ew.data[::2] *= 3.0                                     # (slice assignment through .data: no version bump)
        assert ops.dconv_graph(ei, ew, 12, strict_dense=True) is g1 and torch.equal(m(X, ei, ew), before)   # the documented hazard
        ops.GRAPH_CACHE.forget(ei, ew)
        after = m(X, ei, ew)
        assert not torch.equal(after, before)
        want = m(X, ei.clone(), ew.clone())
        assert torch.equal(after, want)
        keep = ops.GRAPH_CACHE.verify
        ops.GRAPH_CACHE.verify = True
        try:
            ops.GRAPH_CACHE.clear()
            a = m(X, ei, ew)
            assert torch.equal(a, want) and torch.equal(m(X, ei, ew), want)      # a hit: checksums agree
            ew.data[::3] *= 0.5
            assert torch.equal(m(X, ei, ew), m(X, ei.clone(), ew.clone()))       # the write is seen
        finally:
            ops.GRAPH_CACHE.verify = keep
            ops.GRAPH_CACHE.clear()


def test_no_cpu_fallback():
    """A CPU tensor handed to the product path must raise, not silently compute somewhere else."""
    from pytorch_geometric_temporal_amd import _lib
    _lib._set_library_for_testing(None)
    m = DCRNN(2, 3, 2)
    ei_np, ew_np = syn.sensor_graph(10, 40, seed=3)
    with pytest.raises(_lib.PgtError):
        m(torch.zeros(10, 2), torch.from_numpy(ei_np), torch.from_numpy(ew_np))


try:
    from hypothesis import HealthCheck, given, settings, strategies as hst

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(data=hst.data(), K=hst.integers(1, 3), weighted=hst.booleans(), with_h=hst.booleans())
    def test_fuzz_dcrnn_cell_on_arbitrary_graphs(emu_backend, data, K, weighted, with_h):
        """DCRNN cell on random small digraphs in arbitrary edge order (self-loops, sinks, sources, isolated nodes):
        values AND the placement of inf / nan (sources without in-edges: `1/deg_in = inf`) equal
        the oracle's restatement of `dcrnn.py:42-111,172-219`."""
        n = data.draw(hst.integers(1, 8))
        pairs = hst.tuples(hst.integers(0, n - 1), hst.integers(0, n - 1))
        edges = data.draw(hst.lists(pairs, min_size=1, max_size=20, unique=True))
        ei = torch.tensor(edges, dtype=torch.long).t().reshape(2, -1)
        ew = torch.tensor(data.draw(hst.lists(hst.sampled_from([0.5, 1.0, 2.0]), min_size=len(edges),
                                              max_size=len(edges))), dtype=torch.float32) if weighted else None
        n_x = int(ei.max()) + 1 if data.draw(hst.booleans()) else n      # X may have trailing nodes without edges
        if n_x < int(ei.max()) + 1:
            n_x = int(ei.max()) + 1
        fin, O = 2, 3
        torch.manual_seed(len(edges) * 7 + K)
        m = DCRNN(fin, O, K)
        with torch.no_grad():
            for p in m.parameters():
                p.uniform_(-0.5, 0.5)
        params = {k: v.detach().clone() for k, v in m.state_dict().items()}
        X = torch.randn(n_x, fin)
        H = torch.randn(n_x, O) if with_h else None
        with torch.no_grad():
            out = m.to(emu_backend.device)(emu_backend.t(X), emu_backend.t(ei),
                                            None if ew is None else emu_backend.t(ew),
                                            None if H is None else emu_backend.t(H))
            ref = F.dcrnn_cell(X, ei, ew, H, params)
        assert_close_with_nonfinite(out, ref, 2e-5, 1e-5, "fuzz dcrnn")

    @settings(max_examples=50, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(data=hst.data(), K=hst.integers(1, 3), B=hst.integers(1, 3), T=hst.integers(1, 3))
    def test_fuzz_batched_dcrnn_on_multigraphs(emu_backend, data, K, B, T):
        """BatchedDConv / BatchedDCRNN keep duplicate edges as separate messages and every weight as given (the
        scatter form, `dcrnn.py:277-290`) -- unlike DConv's dense path.  Random multigraphs in
        arbitrary edge order against the oracle, inf / nan placement included."""
        n = data.draw(hst.integers(1, 7))
        pairs = hst.tuples(hst.integers(0, n - 1), hst.integers(0, n - 1))
        edges = data.draw(hst.lists(pairs, min_size=1, max_size=18))
        ei = torch.tensor(edges, dtype=torch.long).t().reshape(2, -1)
        ew = torch.tensor(data.draw(hst.lists(hst.sampled_from([0.5, 1.0, 2.0]), min_size=len(edges),
                                              max_size=len(edges))), dtype=torch.float32)
        fin, O = 2, 3
        torch.manual_seed(len(edges) * 5 + K + B)
        m = BatchedDCRNN(fin, O, K)
        with torch.no_grad():
            for p in m.parameters():
                p.uniform_(-0.5, 0.5)
        params = {k: v.detach().clone() for k, v in m.state_dict().items()}
        X = torch.randn(B, T, n, fin)
        with torch.no_grad():
            out = m.to(emu_backend.device)(emu_backend.t(X), emu_backend.t(ei), emu_backend.t(ew))
            ref = F.batched_dcrnn(X, ei, ew, params)
        assert_close_with_nonfinite(out, ref, 2e-5, 1e-5, "fuzz batched dcrnn")
        conv = BatchedDConv(fin + O, O, K)
        conv.load_state_dict({"weight": params["conv_x_z.weight"], "bias": params["conv_x_z.bias"]})
        XH = torch.randn(n, fin + O)
        with torch.no_grad():
            a = conv.to(emu_backend.device)(emu_backend.t(XH), emu_backend.t(ei), emu_backend.t(ew))
            b = F.batched_dconv(XH, ei, ew, params["conv_x_z.weight"], params["conv_x_z.bias"])
        assert_close_with_nonfinite(a, b, 2e-5, 1e-5, "fuzz batched dconv")

    @settings(max_examples=20, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(data=hst.data(), K=hst.integers(1, 3), B=hst.integers(1, 3), T=hst.integers(1, 3), O=hst.sampled_from([4, 8, 6]),
           need_x=hst.booleans())
    def test_fuzz_batched_dcrnn_host_paths_agree(emu_backend, data, K, B, T, O, need_x, monkeypatch):
        """The schedules behind BatchedDCRNN are interchangeable: batch-major rows with the LDS-resident stack vs
        node-major rows with one launch per hop, fused vs separate gate epilogues, hidden-columns-only vs full
        stack gradient -- same outputs, same parameter and input gradients."""
        from pytorch_geometric_temporal_amd import ops
        n = data.draw(hst.integers(2, 7))
        pairs = hst.tuples(hst.integers(0, n - 1), hst.integers(0, n - 1))
        edges = data.draw(hst.lists(pairs, min_size=1, max_size=3 * n, unique=True))
        ei = torch.tensor(edges, dtype=torch.long).t().reshape(2, -1)
        keep = ei[0] != ei[1]
        ei = torch.cat([ei[:, keep], torch.arange(n).repeat(2, 1)], dim=1)          # unit diagonal: finite 1/deg_in
        ew = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(n)) + 0.3
        torch.manual_seed(n * 3 + K)
        m = BatchedDCRNN(2, O, K).to(emu_backend.device)
        with torch.no_grad():
            for p in m.parameters():
                p.uniform_(-0.5, 0.5)
        X = torch.randn(B, T, n, 2)
        W = torch.randn(B, T, n, O)
        results = []
        real_fits = ops.slab_fits
        for slab in (True, False):
            for fuse in (True, False):
                for skip in (True, False):
                    monkeypatch.setattr(ops, "slab_fits", real_fits if slab else (lambda *a, **k: False))
                    monkeypatch.setattr(ops, "FUSE_GATE_EPILOGUES", fuse)
                    monkeypatch.setattr(ops, "SKIP_INPUT_COLUMNS_WHEN_UNUSED", skip)
                    m.zero_grad()
                    Xd = emu_backend.t(X).requires_grad_(need_x)
                    out = m(Xd, emu_backend.t(ei), emu_backend.t(ew))
                    assert out.is_contiguous()
                    (out * emu_backend.t(W)).sum().backward()
                    results.append((out.detach().clone(), Xd.grad.clone() if need_x else None,
                                    [p.grad.clone() for p in m.parameters()]))
        base = results[0]
        for r in results[1:]:
            assert_close_with_nonfinite(r[0], base[0], 2e-5, 2e-5, "output")
            if need_x:
                assert_close_with_nonfinite(r[1], base[1], 1e-4, 1e-4, "dX")
            for a, b in zip(r[2], base[2]):
                assert_close_with_nonfinite(a, b, 2e-4, 2e-4, "parameter gradient")
        ref = F.batched_dcrnn(X, ei, ew, {k: v.detach().clone() for k, v in m.state_dict().items()})
        assert_close_with_nonfinite(base[0], ref, 2e-5, 2e-5, "oracle")
except ImportError:      # hypothesis is optional
    pass


@pytest.mark.parametrize("n,fin,O,bias,hidden,x_grad", [(20, 4, 32, True, False, False),     # BASELINE configs[0]: H = None
                                                        (75, 3, 5, True, True, True),        # three row groups, ragged last
                                                        (33, 6, 64, False, True, False),     # widest hidden, bias-free
                                                        (1, 2, 3, True, True, True)])
def test_dcrnn_k1_cell_is_one_launch_each_way_and_matches_the_oracle(backend, n, fin, O, bias, hidden, x_grad):
    """K = 1 (dcrnn.py:79-82: no hop ever runs): the cell goes through csrc/small_cell.hip — forward, d/dX, d/dH and all
    parameter gradients against the fp64 oracle; H = None is the reference's zero state (dcrnn.py:167-170)."""
    from pytorch_geometric_temporal_amd import ops
    torch.manual_seed(n + O)
    ei_np, ew_np = syn.sensor_graph(n, max(1, 4 * n), seed=n, symmetric=False) if n > 1 else \
        (torch.zeros(2, 1, dtype=torch.long).numpy(), torch.ones(1).numpy())
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    assert ops.cell_k1_fits(n, fin, O) and not ops.cell_k1_fits(n, fin, 65) and not ops.cell_k1_fits(5000, fin, O)
    m = DCRNN(fin, O, 1, bias=bias)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.6, 0.6)
    params64 = {k: v.detach().double().requires_grad_() for k, v in m.state_dict().items()}
    m = m.to(backend.device)
    X, H, w = torch.randn(n, fin), torch.randn(n, O), torch.randn(n, O)
    Xd = backend.t(X).requires_grad_(x_grad)
    Hd = backend.t(H).requires_grad_() if hidden else None
    calls = []
    orig = ops.DCRNNCellK1Function.apply
    try:
        ops.DCRNNCellK1Function.apply = lambda *a: (calls.append(1), orig(*a))[1]
        out = m(Xd, backend.t(ei), backend.t(ew), Hd)
    finally:
        ops.DCRNNCellK1Function.apply = orig
    assert calls == [1]
    (out * backend.t(w)).sum().backward()
    X64 = X.double().requires_grad_()
    H64 = H.double().requires_grad_() if hidden else None
    ref = F.dcrnn_cell(X64, ei, ew.double(), H64, params64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward")
    if x_grad:
        assert_close_with_nonfinite(Xd.grad, X64.grad, 2e-5, 1e-4, "dX")
    else:
        assert Xd.grad is None
    if hidden:
        assert_close_with_nonfinite(Hd.grad, H64.grad, 2e-5, 1e-4, "dH")
    for name, p in m.named_parameters():
        assert_close_with_nonfinite(p.grad, params64[name].grad, 5e-5, 1e-4, name)
    # the same cell through the general path (diffusion-stack kernels with K = 1) agrees to rounding
    g = ops.dconv_graph(backend.t(ei), backend.t(ew), n, strict_dense=False)
    from pytorch_geometric_temporal_amd.nn.recurrent.dcrnn import _cell_weights
    with torch.no_grad():
        Wzr, bzr, Wh, bh = _cell_weights(m.conv_x_z, m.conv_x_r, m.conv_x_h)
        H0 = backend.t(H) if hidden else torch.zeros(n, O, device=backend.device)
        general = ops.DCRNNSeqFunction.apply(backend.t(X).unsqueeze(0), H0, Wzr, bzr, Wh, bh, g, 1, 1)[0]
    assert_close_with_nonfinite(out.detach(), general, 2e-6, 1e-5, "one-launch cell vs general path")


def test_batched_dcrnn_states_route_a_skinny_torch_linear_readout_to_the_streaming_kernels(backend):
    """The reference's examples apply `torch.nn.Linear(out, 1 … 4)` to BatchedDCRNN's `[B, T, N, out]` result
    (pems_bay_main.py / dcrnn_example.py): that one call runs on this package's skinny kernels (same values, same
    gradients as torch's own product); anything else done with the result is ordinary torch and yields plain tensors."""
    import torch.nn.functional as TF
    from pytorch_geometric_temporal_amd import ops
    from pytorch_geometric_temporal_amd.nn.recurrent.dcrnn import _StatesTensor
    torch.manual_seed(3)
    n, B, T, fin, O = 12, 3, 4, 2, 8
    ei_np, ew_np = syn.sensor_graph(n, 60, seed=1, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    m = BatchedDCRNN(fin, O, 2).to(backend.device)
    head = torch.nn.Linear(O, 2).to(backend.device)
    wide = torch.nn.Linear(O, 16).to(backend.device)
    X = backend.t(torch.randn(B, T, n, fin))
    calls = []
    orig = ops.linear
    try:
        ops.linear = lambda *a: (calls.append(1), orig(*a))[1]
        h = m(X, ei, ew)
        assert type(h) is _StatesTensor and h.shape == (B, T, n, O) and h.is_contiguous()
        y = head(h)                                   # routed
        assert calls == [1] and type(y) is torch.Tensor
        z = wide(h)                                   # 16 output features: torch's own product
        assert calls == [1] and type(z) is torch.Tensor
        assert type(h * 2) is torch.Tensor and type(h.sum()) is torch.Tensor and type(h.shape) is torch.Size
        # the reference's own models put a relu before the read-out (dcrnn_example.py:27-28, tgcn/metr_la_main.py:43-44):
        # relu(states) keeps the routing, so that read-out runs on the streaming kernels as well
        for r in (torch.relu(h), TF.relu(h), h.relu()):
            assert type(r) is _StatesTensor
        fused = []
        orig_ro = ops.ReadoutFunction.apply
        ops.ReadoutFunction.apply = lambda *a: (fused.append(a[3]), orig_ro(*a))[1]
        try:
            yr = TF.linear(torch.relu(h), head.weight, head.bias)
        finally:
            ops.ReadoutFunction.apply = orig_ro
        # relu -> Linear: one pass over the pre-relu states (csrc/readout.hip), not ops.linear on the relu tensor
        assert type(yr) is torch.Tensor and calls == [1] and fused == [True]
        assert_close_with_nonfinite(yr, TF.linear(torch.relu(h.as_subclass(torch.Tensor)), head.weight, head.bias), 1e-6, 1e-5,
                                    "relu -> read-out")
        with torch.autocast(device_type=backend.device.type, dtype=torch.bfloat16):       # under autocast: the stock op, stock dtype
            ya = head(h)
        assert calls == [1] and ya.dtype == torch.bfloat16
        import io, pickle
        buf = io.BytesIO()
        torch.save(h.detach(), buf)
        buf.seek(0)
        assert type(torch.load(buf)) is torch.Tensor
        assert type(pickle.loads(pickle.dumps(h.detach()))) is torch.Tensor
    finally:
        ops.linear = orig
    (y.square().mean() + z.mean()).backward()
    g_head, g_conv = head.weight.grad.clone(), m.conv_x_z.weight.grad.clone()
    # the same computation on plain tensors
    head.zero_grad(); wide.zero_grad(); m.zero_grad()
    m.readout_interception = False
    h2 = m(X, ei, ew)
    assert type(h2) is torch.Tensor
    y2, z2 = head(h2), wide(h2)
    (y2.square().mean() + z2.mean()).backward()
    assert_close_with_nonfinite(y, y2, 1e-6, 1e-5, "read-out")
    assert_close_with_nonfinite(g_head, head.weight.grad, 1e-6, 1e-4, "d/d read-out weight")
    assert_close_with_nonfinite(g_conv, m.conv_x_z.weight.grad, 1e-6, 1e-4, "d/d conv weight")
    m.readout_interception = True
    sd = m.state_dict()                                # nothing about the module's parameters changes
    assert all(type(v) is torch.Tensor for v in sd.values())


def test_dcrnn_k1_cell_takes_strided_rows(backend):
    """X and H handed in as column slices of wider tensors (row stride > row width): read in place."""
    torch.manual_seed(5)
    n, fin, O = 40, 3, 6
    m = DCRNN(fin, O, 1).to(backend.device)
    big_x, big_h = backend.t(torch.randn(n, fin + 5)), backend.t(torch.randn(n, O + 3))
    X, H = big_x[:, 2:2 + fin], big_h[:, 1:1 + O]
    assert X.stride(0) == fin + 5 and not X.is_contiguous()
    ei = backend.t(torch.zeros(2, 1, dtype=torch.long))
    with torch.no_grad():
        a = m(X, ei, None, H)
        b = m(X.contiguous(), ei, None, H.contiguous())
    assert torch.equal(a, b)


@pytest.mark.parametrize("Fin,O,K,B,T", [(2, 2, 3, 5, 4), (2, 8, 2, 3, 3), (1, 4, 1, 2, 2), (3, 5, 3, 1, 1)])
def test_one_workgroup_sequence_kernels_equal_the_general_path(backend, Fin, O, K, B, T):
    """csrc/seq_small.hip (the whole sequence of a sample in one workgroup, one launch each way) against the general path
    (LDS-resident stacks + MFMA products + gate kernels, one launch per phase): the states, dX and every parameter gradient of
    BatchedDCRNN — the reference's own BatchedDCRNN(2, 2, K = 3) among the shapes."""
    from pytorch_geometric_temporal_amd import ops
    torch.manual_seed(Fin * 10 + O)
    n = 23
    ei_np, ew_np = syn.sensor_graph(n, 110, seed=4, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    m = BatchedDCRNN(Fin, O, K).to(backend.device)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.5, 0.5)
    X = torch.randn(B, T, n, Fin)
    w = backend.t(torch.randn(B, T, n, O))
    res = {}
    for small in (True, False):
        ops.USE_SEQ_SMALL = small
        try:
            m.zero_grad()
            Xd = backend.t(X).requires_grad_()
            out = m(Xd, ei, ew)
            (out * w).sum().backward()
            res[small] = (out.detach().clone(), Xd.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
        finally:
            ops.USE_SEQ_SMALL = True
    assert ops.seq_small_fits(ops.dconv_graph(ei, ew, n), Fin, O, K)
    assert_close_with_nonfinite(res[True][0], res[False][0], 2e-6, 1e-5, "states")
    assert_close_with_nonfinite(res[True][1], res[False][1], 1e-5, 1e-4, "dX")
    for k in res[True][2]:
        ref = res[False][2][k]
        assert_close_with_nonfinite(res[True][2][k], ref, 2e-5 * float(ref.abs().max()) + 1e-7, 1e-4, k)


@pytest.mark.parametrize("n,E,K,B,T,bias,x_grad", [(37, 251, 3, 3, 3, True, False),     # ragged last row tile, odd slot counts
                                                     (48, 300, 2, 2, 2, True, False),     # K = 2: three segments
                                                     (21, 120, 3, 5, 2, False, True)])    # no bias; the input wants a gradient
def test_hidden_64_sequences_in_one_launch_equal_the_per_step_launches(backend, n, E, K, B, T, bias, x_grad):
    """csrc/seq64.hip (all T steps of a sample in one workgroup: hops out of LDS, split-bf16 products on the terms while they
    are there, gate chains on the accumulators) against the per-step launches of ops.DCRNNSeqFunction: the states, and — through
    the general backward on what the launch saved — dX and every parameter gradient.  The diffusion terms it saves must be the
    per-step path's bit for bit (the same fmaf chain in slot order)."""
    from pytorch_geometric_temporal_amd import ops
    torch.manual_seed(n + K)
    ei_np, ew_np = syn.sensor_graph(n, E, seed=2, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    m = BatchedDCRNN(2, 64, K, bias=bias).to(backend.device)
    m.readout_interception = False                 # plain tensors: the result's grad_fn is the sequence function itself
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.2, 0.2)
    X = torch.randn(B, T, n, 2)
    w = backend.t(torch.randn(B, T, n, 64))
    g = ops.dconv_graph(ei, ew, n)
    assert ops.seq64_fits(g, 2, 64, K)
    res, saved = {}, {}
    keep = (ops.USE_SEQ64, ops.SEQ64_MIN_BATCH)
    # "one_launch": csrc/seq64.hip both ways; "adjoint_only": the per-step forward with the one-launch adjoint (what batches
    # below SEQ64_MIN_BATCH run); "per_step": neither
    modes = {"one_launch": (True, 1), "adjoint_only": (True, 1 << 30), "per_step": (False, 1)}
    try:
        for mode, (use, min_b) in modes.items():
            ops.USE_SEQ64, ops.SEQ64_MIN_BATCH = use, min_b
            m.zero_grad()
            Xd = backend.t(X).requires_grad_(x_grad)
            out = m(Xd, ei, ew)
            assert (type(out.grad_fn).__name__ == "DCRNNSeq64FunctionBackward") == (mode == "one_launch")
            saved[mode] = [t.clone() for t in out.grad_fn.saved_tensors[:4]]   # both stacks, Z | R, the candidates
            (out * w).sum().backward()
            res[mode] = (out.detach().clone(), Xd.grad.clone() if x_grad else None,
                         {k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        ops.USE_SEQ64, ops.SEQ64_MIN_BATCH = keep
    assert torch.equal(res["adjoint_only"][0], res["per_step"][0])
    assert_close_with_nonfinite(res["one_launch"][0], res["per_step"][0], 3e-6, 1e-5, "states")
    # segment 0 of step 0 of the gate stack = [X_0 | H_0]: identical inputs -> every term of that step identical
    assert torch.equal(saved["one_launch"][0][:, 0], saved["per_step"][0][:, 0]), "diffusion terms of the first step"
    for a, b, what in zip(saved["one_launch"], saved["per_step"], ("gate stack", "candidate stack", "Z | R", "candidates")):
        assert_close_with_nonfinite(a, b, 5e-6, 1e-5, what)
    for mode in ("one_launch", "adjoint_only"):
        if x_grad:
            assert_close_with_nonfinite(res[mode][1], res["per_step"][1], 1e-5, 1e-4, f"dX ({mode})")
        for k in res[mode][2]:
            ref = res["per_step"][2][k]
            assert_close_with_nonfinite(res[mode][2][k], ref, 2e-5 * float(ref.abs().max()) + 1e-7, 1e-4, f"{k} ({mode})")


def test_hidden_64_one_launch_path_is_taken_only_where_it_applies(backend):
    """Dispatch: hidden width 64 with two input channels on a graph with finite coefficients from SEQ64_MIN_BATCH samples on; a node
    without incoming edges (infinite 1 / deg, dcrnn.py:71-77) keeps the per-step launches, whose products place inf / nan like the
    reference's."""
    from pytorch_geometric_temporal_amd import ops
    ei_np, ew_np = syn.sensor_graph(30, 170, seed=1, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    g = ops.dconv_graph(ei, ew, 30)
    assert g.finite and ops.seq64_fits(g, 2, 64, 3) and ops.seq64_fits(g, 2, 64, 2)
    assert not ops.seq64_fits(g, 2, 32, 3) and not ops.seq64_fits(g, 3, 64, 3) and not ops.seq64_fits(g, 2, 64, 4)
    src = torch.tensor([0, 1, 2, 3, 4, 5]); dst = torch.tensor([1, 2, 3, 4, 5, 1])          # node 0 has no incoming edge
    g2 = ops.dconv_graph(backend.t(torch.stack([src, dst])), None, 6)
    assert not g2.finite and not ops.seq64_fits(g2, 2, 64, 3)
    calls = []
    orig = ops.DCRNNSeq64Function.apply
    try:
        ops.DCRNNSeq64Function.apply = staticmethod(lambda *a: calls.append(1) or orig(*a))
        m = BatchedDCRNN(2, 64, 3).to(backend.device)
        keep = ops.SEQ64_MIN_BATCH
        ops.SEQ64_MIN_BATCH = 3
        try:
            with torch.no_grad():
                m(backend.t(torch.randn(2, 2, 30, 2)), ei, ew)
                assert not calls
                m(backend.t(torch.randn(3, 2, 30, 2)), ei, ew)
                assert len(calls) == 1
        finally:
            ops.SEQ64_MIN_BATCH = keep
    finally:
        ops.DCRNNSeq64Function.apply = orig


@pytest.mark.parametrize("K,hidden", [(2, True), (3, True), (3, False)])
def test_dcrnn_cell_with_hops_on_a_small_graph_is_one_launch_each_way(backend, K, hidden):
    """DCRNN(in, out, K > 1) on a graph of tens of nodes (Chickenpox; test/recurrent_test.py:274-315 uses K = 2, 3): one
    pgt_dcrnn_seq_small_f32 call forward, one backward, equal to the general path incl. d/dH."""
    from pytorch_geometric_temporal_amd import _lib, ops
    torch.manual_seed(K)
    n, fin, O = 20, 4, 32
    ei_np, ew_np = syn.sensor_graph(n, 102, seed=1, symmetric=True)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    m = DCRNN(fin, O, K).to(backend.device)
    X, H = torch.randn(n, fin), torch.randn(n, O)
    w = backend.t(torch.randn(n, O))
    lib = _lib.get_lib()
    calls = []
    orig = lib.call
    res = {}
    for small in (True, False):
        ops.USE_SEQ_SMALL = small
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            m.zero_grad()
            Xd, Hd = backend.t(X).requires_grad_(), (backend.t(H).requires_grad_() if hidden else None)
            calls.clear()
            out = m(Xd, ei, ew, Hd)
            fwd_calls = [c for c in calls if c not in ("pgt_dcrnn_pack_weights_f32", "pgt_dconv_prep", "pgt_csr_locality")]
            (out * w).sum().backward()
            res[small] = (out.detach().clone(), Xd.grad.clone(), None if Hd is None else Hd.grad.clone(),
                          {k: p.grad.clone() for k, p in m.named_parameters()}, fwd_calls, list(calls))
        finally:
            lib.call = orig
            ops.USE_SEQ_SMALL = True
    assert res[True][4] == ["pgt_dcrnn_seq_small_f32"], res[True][4]
    assert [c for c in res[True][5] if "seq_small" in c] == ["pgt_dcrnn_seq_small_f32", "pgt_dcrnn_seq_small_bwd_f32"]
    assert_close_with_nonfinite(res[True][0], res[False][0], 2e-6, 1e-5, "H'")
    assert_close_with_nonfinite(res[True][1], res[False][1], 1e-5, 1e-4, "dX")
    if hidden:
        assert_close_with_nonfinite(res[True][2], res[False][2], 1e-5, 1e-4, "dH")
    for k in res[True][3]:
        ref = res[False][3][k]
        assert_close_with_nonfinite(res[True][3][k], ref, 2e-5 * float(ref.abs().max()) + 1e-7, 1e-4, k)


def test_dconv_on_a_graph_whose_numbering_hides_its_locality(backend):
    """`DConv(64, 64, K=3)` on a mesh whose node ids are shuffled (a sensor list in file order: dcrnn.py:300-313 aggregates over any
    edge list): every hop of the stack and of its adjoint runs the window kernel through a patch order of the library's own
    (ops.RenumberedEllw) — the module's output and every gradient equal the CSR row tiles' on the caller's numbering."""
    from pytorch_geometric_temporal_amd import ops
    side = 34 if backend.name == "emu" else 120           # (the CPU double is slow: a smaller mesh, the size gate lowered for it)
    n = side * side
    min_rows, ops.ELLW_MIN_ROWS = ops.ELLW_MIN_ROWS, min(ops.ELLW_MIN_ROWS, n)
    try:
        ei, ew = syn.grid2d_graph(side, "shuffled", seed=4)
        ei, ew = backend.t(ei), backend.t(ew)
        torch.manual_seed(2)
        m = DConv(64, 64, 3).to(backend.device)
        X0 = backend.t(torch.randn(n, 64))
        outs = []
        for renumber in (True, False):
            ops.GRAPH_CACHE.clear()
            saved, ops.USE_RENUMBER = ops.USE_RENUMBER, renumber
            try:
                X = X0.clone().requires_grad_(True)
                m.zero_grad()
                H = m(X, ei, ew)
                (H * torch.linspace(-1, 1, 64, device=H.device)).sum().backward()
                graphs = [v for v in ops.GRAPH_CACHE._d.values() if hasattr(v, "fwd_o")] + \
                         [w for v in ops.GRAPH_CACHE._d.values() if isinstance(v, tuple) for w in v if hasattr(w, "fwd_o")]
            finally:
                ops.USE_RENUMBER = saved
            assert graphs
            for csr in (graphs[0].fwd_o, graphs[0].fwd_i, graphs[0].bwd_o, graphs[0].bwd_i):
                assert bool(csr.ellw and csr.ellw.order is not None) == renumber
                if renumber:      # the four operators of a graph (two directions, forward / transposed) share ONE set of patches
                    assert torch.equal(csr.ellw.order, graphs[0].fwd_o.ellw.order)
            outs.append((H.detach(), X.grad.clone(), m.weight.grad.clone()))
    finally:
        ops.ELLW_MIN_ROWS = min_rows
    for a, b, what in zip(outs[0], outs[1], ("H", "dX", "dW")):
        assert_close_with_nonfinite(a, b, 1e-5 * max(1.0, float(b.abs().max())), 1e-5, what)


def test_dconv_on_a_shuffled_graph_with_hubs(backend):
    """`DConv(64, 64, K=3)` on a mesh in an arbitrary numbering with a hub (300 more in-edges) and a junction (40): the hub is a wide
    ROW of P_o and of P_i's transpose and a wide COLUMN of the other two operators; all four are laid out in ONE patch order grown
    without the hub nodes, the wide rows are left out of the layouts and ride with the tiles — output and every gradient equal the CSR
    kernels' on the caller's numbering."""
    import sys
    import numpy as np
    from pytorch_geometric_temporal_amd import ops
    side = 40 if backend.name == "emu" else 120           # (the CPU double is slow: a smaller mesh, the size gate lowered for it)
    n = side * side
    min_rows, ops.ELLW_MIN_ROWS = ops.ELLW_MIN_ROWS, min(ops.ELLW_MIN_ROWS, n)
    rng = np.random.default_rng(5)
    ei, ew = syn.grid2d_graph(side, "shuffled", seed=4)
    wide, extra = np.array([n // 5, n - 11]), (300, 40)
    src = np.concatenate([rng.choice(n, k, replace=False) for k in extra])
    e2 = np.concatenate([ei, np.stack([src, np.repeat(wide, extra)])], axis=1)
    w2 = np.concatenate([ew, (0.5 + rng.random(src.size)).astype(np.float32)])
    key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    ei, ew = backend.t(e2[:, key]), backend.t(w2[key])
    torch.manual_seed(3)
    m = DConv(64, 64, 3).to(backend.device)
    X0 = backend.t(torch.randn(n, 64))
    outs = []
    for ellw in (True, False):
        ops.GRAPH_CACHE.clear()
        saved, ops.USE_ELLW = ops.USE_ELLW, ellw
        try:
            X = X0.clone().requires_grad_(True)
            m.zero_grad()
            H = m(X, ei, ew)
            (H * torch.linspace(-1, 1, 64, device=H.device)).sum().backward()
            graphs = [v for v in ops.GRAPH_CACHE._d.values() if hasattr(v, "fwd_o")] + \
                     [w for v in ops.GRAPH_CACHE._d.values() if isinstance(v, tuple) for w in v if hasattr(w, "fwd_o")]
        finally:
            ops.USE_ELLW = saved
            if not ellw or sys.exc_info()[0] is not None:
                ops.ELLW_MIN_ROWS = min_rows
        assert graphs
        g = graphs[0]
        if ellw:
            left = [0 if c.ellw.left_out is None else c.ellw.left_out for c in (g.fwd_o, g.fwd_i, g.bwd_o, g.bwd_i)]
            assert all(c.ellw and c.ellw.order is not None for c in (g.fwd_o, g.fwd_i, g.bwd_o, g.bwd_i)), "four renumbered layouts"
            assert sorted(left) == [0, 0, 2, 2], left                      # wide rows in two operators, wide columns in the other two
            for c in (g.fwd_i, g.bwd_o, g.bwd_i):     # one set of patches per tile height (a wider slot block plans lower tiles)
                assert c.ellw.tile_rows != g.fwd_o.ellw.tile_rows or torch.equal(c.ellw.order, g.fwd_o.ellw.order)
        outs.append((H.detach(), X.grad.clone(), m.weight.grad.clone()))
    for a, b, what in zip(outs[0], outs[1], ("H", "dX", "dW")):
        assert_close_with_nonfinite(a, b, 2e-5 * max(1.0, float(b.abs().max())), 1e-5, what)


def test_training_step_gradients_are_bitwise_reproducible_by_default(backend):
    """The reference's CPU path gives the same gradients run after run; so does this one with its default settings: the weight
    gradients are per-slab partial sums added in a fixed order (ops.DETERMINISTIC_WEIGHT_GRADIENTS, on by default), the sequence
    kernels and the read-out use no float atomics.  On the GPU the batch is large enough for the split-bf16 weight-gradient kernel
    (>= 16 384 rows over all steps), whose workgroups retire in a different order every run."""
    from pytorch_geometric_temporal_amd import ops
    assert ops.DETERMINISTIC_WEIGHT_GRADIENTS
    n, E, B, T = (207, 1515, 8, 12) if backend.name == "hip" else (23, 140, 2, 3)
    ei_np, ew_np = syn.sensor_graph(n, E, seed=3, symmetric=False)
    ei, ew = backend.t(torch.from_numpy(ei_np)), backend.t(torch.from_numpy(ew_np))
    torch.manual_seed(5)
    m = BatchedDCRNN(2, 64, 3).to(backend.device)
    lin = torch.nn.Linear(64, 1).to(backend.device)
    X = backend.t(torch.randn(B, T, n, 2))
    y = backend.t(torch.randn(B, T, n, 1))
    runs = []
    for _ in range(3):
        m.zero_grad()
        lin.zero_grad()
        loss = ((lin(torch.relu(m(X, ei, ew))) - y) ** 2).mean()
        loss.backward()
        runs.append([p.grad.clone() for p in list(m.parameters()) + list(lin.parameters())] + [loss.detach().clone()])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
