"""Parity at BASELINE.json's OWN shapes (configs 2-5), against fixtures produced by the reference's actual module files
(tests/golden/baseline_*.npz, oracle/make_golden.py) and — for the gradients — against the fp64 oracle.

The multi-megabyte inputs are rebuilt from seeds (oracle/baseline_cases.py); a fixture holds the parameters, a sample
of the reference output and fp64 checksums of every slice of it.  The GPU legs go through libpgt_hip.so on cuda:0; the
CPU legs pin the oracle itself to the same fixtures.
"""
import math

import pytest
import torch

from conftest import assert_close_with_nonfinite, load_golden
from oracle import baseline_cases as BC
from oracle import functional as F
from pytorch_geometric_temporal_amd.nn.recurrent import A3TGCN2, TGCN2, BatchedDCRNN, EvolveGCNH

ATOL, RTOL = 1e-5, 1e-5   # north_star: forward within 1e-5 of the reference CPU forward (fp32)


def _check_sums(out, ref_sums, keep_dims, what):
    """Checksum of every slice: |sum - ref| <= 1e-5 * (sum |x| + sqrt(n)) — catches a wrong element anywhere in the
    tensor (the sampled slices carry the element-wise 1e-5 check)."""
    got = BC.slice_sums(out, keep_dims)
    mag = BC.slice_sums(out.abs(), keep_dims)
    n = out.numel() / got.numel()
    ref = ref_sums.double().reshape(got.shape)
    bad = (got - ref).abs() > 1e-5 * (mag + math.sqrt(n))
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} slice checksum(s) off, worst {float((got - ref).abs().max()):.3e}"


def _grad_check(model, params64, atol, rtol):
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        assert_close_with_nonfinite(p.grad, params64[name].grad, atol * float(params64[name].grad.abs().max() + 1), rtol, name)


def _params64(g):
    return {k: v.double().requires_grad_() for k, v in g["param"].items()}


# ------------------------------------------------------------------------------------------------ config 2

@pytest.fixture(params=["per_step_launches", "one_launch_sequences"])
def dcrnn64_path(request):
    """Both forward paths of the hidden-64 model at B = 64: the per-step launches (the default below SEQ64_MIN_BATCH samples) and
    the one-launch sequence kernel (csrc/seq64.hip, the default from there on: what bench.py's B = 1024 runs)."""
    from pytorch_geometric_temporal_amd import ops
    keep = (ops.USE_SEQ64, ops.SEQ64_MIN_BATCH)
    ops.USE_SEQ64, ops.SEQ64_MIN_BATCH = request.param == "one_launch_sequences", 1
    yield request.param
    ops.USE_SEQ64, ops.SEQ64_MIN_BATCH = keep


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1515, 1722])
def test_config2_benchmarked_dcrnn_forward_matches_reference_fixture(E, dcrnn64_path):
    """BatchedDCRNN(2, 64, K=3), 207 nodes, B = 64 x 12 steps: the model bench.py times, through the slab kernels at
    C = 66 and the gate-fused GEMMs — and through the one-launch sequence kernel —, against the reference module's output."""
    dev = torch.device("cuda:0")
    g = load_golden("baseline_c2_batched_dcrnn64")
    ei, ew, X = BC.metrla(E)
    m = BatchedDCRNN(2, 64, 3)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(dev)
    with torch.no_grad():
        out = m(X.to(dev), ei.to(dev), ew.to(dev))
    assert tuple(out.shape) == (64, 12, 207, 64) and out.is_contiguous()
    sample = out[list(BC.METRLA_SAMPLE_B)][:, list(BC.METRLA_SAMPLE_T)]
    assert_close_with_nonfinite(sample, g["out"][f"sample_E{E}"], ATOL, RTOL, f"E={E} sampled (b, t) slices")
    _check_sums(out, g["out"][f"sums_bt_E{E}"], (0, 1), f"E={E}")


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1515, 1722])
def test_config2_benchmarked_dcrnn_gradients_match_fp64_oracle(E, dcrnn64_path):
    """Every parameter gradient of the benchmarked model (hidden 64, B = 64) against autograd through the fp64 oracle
    (op for op the reference).  dW comes from fp32 atomics: tolerance 2e-4 of the gradient's scale."""
    dev = torch.device("cuda:0")
    g = load_golden("baseline_c2_batched_dcrnn64")
    ei, ew, X = BC.metrla(E)
    w = BC.rand((64, 12, 207, 64), 220)
    m = BatchedDCRNN(2, 64, 3)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(dev)
    out = m(X.to(dev), ei.to(dev), ew.to(dev))
    (out * w.to(dev)).sum().backward()
    p64 = _params64(g)
    ref = F.batched_dcrnn(X.double(), ei, ew.double(), p64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward vs fp64 oracle")
    _grad_check(m, p64, 2e-4, 2e-4)


@pytest.mark.gpu
def test_config2_bench_batch_1024_sampled_against_the_oracle():
    """The batch bench.py times: B = 1024 windows (211 968 rows per product: the split-bf16 kernels, the single
    320-column feature-gradient product, two weight-gradient launches per stack).  Samples are independent
    (dcrnn.py:363-369 replicates the graph block-diagonally), so the fp64 oracle on samples {0, 511, 1023} pins
    (a) their forward outputs and (b) — with the loss weights zero elsewhere — every parameter gradient of the whole
    batch; (c) linearity in the loss weights ties the full-batch gradient to its two halves at size."""
    dev = torch.device("cuda:0")
    g = load_golden("baseline_c2_batched_dcrnn64")
    ei, ew, _ = BC.metrla(1515)
    B, pick = 1024, [0, 511, 1023]
    X = BC.rand((B, 12, 207, 2), 1201)
    w = BC.rand((B, 12, 207, 64), 1202)
    m = BatchedDCRNN(2, 64, 3)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(dev)
    Xd, eid, ewd = X.to(dev), ei.to(dev), ew.to(dev)

    def grads(wmask):
        m.zero_grad()
        out = m(Xd, eid, ewd)
        (out * wmask).sum().backward()
        return out.detach(), [p.grad.clone() for p in m.parameters()]

    w3 = torch.zeros_like(w)
    w3[pick] = w[pick]
    out, g3 = grads(w3.to(dev))
    assert tuple(out.shape) == (B, 12, 207, 64) and bool(torch.isfinite(out).all())
    p64 = _params64(g)
    ref = F.batched_dcrnn(X[pick].double(), ei, ew.double(), p64)
    (ref * w[pick].double()).sum().backward()
    assert_close_with_nonfinite(out[pick], ref, ATOL, RTOL, "samples 0, 511, 1023 vs fp64 oracle")
    for (name, _), got in zip(m.named_parameters(), g3):
        r = p64[name].grad
        assert_close_with_nonfinite(got, r, 2e-4 * float(r.abs().max() + 1), 2e-4, name)
    # every sample is a valid forward (checksum-level): the batch is a stack of independent samples, so a sample's output
    # may not depend on its position — samples 0 and 511 swapped give swapped outputs
    Xs = Xd.clone()
    Xs[0], Xs[511] = Xd[511], Xd[0]
    with torch.no_grad():
        outs = m(Xs, eid, ewd)
    assert float((outs[0] - out[511]).abs().max()) <= 2e-6 and float((outs[511] - out[0]).abs().max()) <= 2e-6
    # linearity of the gradient in the loss weights, at size
    wd = w.to(dev)
    lo, hi = wd.clone(), wd.clone()
    lo[B // 2:] = 0
    hi[:B // 2] = 0
    _, gf = grads(wd)
    _, gl = grads(lo)
    _, gh = grads(hi)
    for (name, _), a, b, c in zip(m.named_parameters(), gf, gl, gh):
        scale = float(a.abs().max()) + 1.0
        assert float((a - (b + c)).abs().max()) <= 3e-4 * scale, name


def test_config2_oracle_reproduces_the_reference_fixture():
    """The fp32 oracle (oracle/functional.py) against the reference module's output at the benchmarked shape
    (a quarter of the batch: the samples are independent)."""
    g = load_golden("baseline_c2_batched_dcrnn64")
    ei, ew, X = BC.metrla(1515)
    with torch.no_grad():
        out = F.batched_dcrnn(X[:1], ei, ew, g["param"])
    assert_close_with_nonfinite(out[0, list(BC.METRLA_SAMPLE_T)], g["out"]["sample_E1515"][0], 2e-6, 2e-6, "b = 0")


# ------------------------------------------------------------------------------------------------ config 3

@pytest.mark.gpu
def test_config3_a3tgcn2_pemsbay_forward_and_gradients():
    """A3TGCN2(2, 32, periods=12), 325 nodes / 2 694 edges, B = 64 (attentiontemporalgcn.py:130-157)."""
    dev = torch.device("cuda:0")
    g = load_golden("baseline_c3_a3tgcn2_pemsbay")
    ei, ew, X, H0 = BC.pemsbay()
    m = A3TGCN2(2, 32, periods=12, batch_size=64)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(dev)
    sel = list(BC.PEMSBAY_SAMPLE_B)
    with torch.no_grad():
        o1 = m(X.to(dev), ei.to(dev), ew.to(dev))
        o2 = m(X.to(dev), ei.to(dev), ew.to(dev), H0.to(dev))
    assert tuple(o1.shape) == (64, 325, 32)
    assert_close_with_nonfinite(o1[sel], g["out"]["sample_weight"], ATOL, RTOL, "weight")
    assert_close_with_nonfinite(o2[sel], g["out"]["sample_weight_hidden"], ATOL, RTOL, "weight+hidden")
    _check_sums(o1, g["out"]["sums_weight"], (0, 1), "weight")
    _check_sums(o2, g["out"]["sums_weight_hidden"], (0, 1), "weight+hidden")
    # gradients against the fp64 oracle
    w = BC.rand((64, 325, 32), 320)
    Xd, Hd = X.to(dev).requires_grad_(), H0.to(dev).requires_grad_()
    out = m(Xd, ei.to(dev), ew.to(dev), Hd)
    (out * w.to(dev)).sum().backward()
    p64 = _params64(g)
    X64, H64 = X.double().requires_grad_(), H0.double().requires_grad_()
    ref = F.a3tgcn(X64, ei, ew.double(), H64, p64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward vs fp64 oracle")
    assert_close_with_nonfinite(Xd.grad, X64.grad, 5e-5, 1e-4, "dX")
    assert_close_with_nonfinite(Hd.grad, H64.grad, 5e-5, 1e-4, "dH")
    _grad_check(m, p64, 1e-4, 1e-4)


def test_config3_oracle_reproduces_the_reference_fixture():
    g = load_golden("baseline_c3_a3tgcn2_pemsbay")
    ei, ew, X, H0 = BC.pemsbay()
    sel = list(BC.PEMSBAY_SAMPLE_B)
    with torch.no_grad():
        o2 = F.a3tgcn(X[sel], ei, ew, H0[sel], g["param"])
    assert_close_with_nonfinite(o2, g["out"]["sample_weight_hidden"], 2e-6, 2e-6, "weight+hidden")


# ------------------------------------------------------------------------------------------------ config 4

@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["local", "uniform"])
def test_config4_tgcn2_50k_nodes_forward_and_gradients(kind):
    """TGCN2(2, 32), 50 000 nodes / 400 000 edges, B = 8 (temporalgcn.py:187-233): the wide aggregation kernel with
    its XCD column-slab mapping at real size, locality-ordered and uniform-random neighbours."""
    dev = torch.device("cuda:0")
    g = load_golden("baseline_c4_tgcn2_50k")
    ei, ew, X, H0 = BC.graph50k(kind)
    nodes = BC.sample_nodes_50k()
    assert torch.equal(nodes, g["in"]["sample_nodes"])
    m = TGCN2(2, 32, batch_size=8)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(dev)
    Xd, Hd = X.to(dev).requires_grad_(), H0.to(dev).requires_grad_()
    out = m(Xd, ei.to(dev), ew.to(dev), Hd)
    assert tuple(out.shape) == (8, 50_000, 32)
    assert_close_with_nonfinite(out[:, nodes.to(dev)], g["out"][f"sample_{kind}"], ATOL, RTOL, f"{kind}: sampled nodes")
    got = out.detach().double().cpu().view(8, 100, 500 * 32)
    bad = (got.sum(-1) - g["out"][f"sums_{kind}"]).abs() > 1e-5 * (got.abs().sum(-1) + math.sqrt(16000))
    assert not bool(bad.any()), f"{kind}: {int(bad.sum())} block checksum(s) off"
    w = BC.rand((8, 50_000, 32), 420)
    (out * w.to(dev)).sum().backward()
    p64 = _params64(g)
    X64, H64 = X.double().requires_grad_(), H0.double().requires_grad_()
    ref = F.tgcn_cell(X64, ei, ew.double(), H64, p64)
    (ref * w.double()).sum().backward()
    assert_close_with_nonfinite(out, ref, ATOL, RTOL, "forward vs fp64 oracle")
    assert_close_with_nonfinite(Xd.grad, X64.grad, 5e-5, 1e-4, "dX")
    assert_close_with_nonfinite(Hd.grad, H64.grad, 5e-5, 1e-4, "dH")
    _grad_check(m, p64, 2e-4, 2e-4)


def _batched_tgcn_against_oracle(device, ei, ew, X, y, hidden, atol_fwd, grad_tol):
    """The reference example's model and loss (examples/indexBatching/tgcn/metr_la_main.py:29-56, :86-87) through bench_tgcn's
    BatchedTGCN on `device` against the fp64 oracle running the same T-step loop (oracle/functional.py tgcn_cell per step, torch
    relu / Linear / masked MAE): forward within `atol_fwd`, every parameter gradient within `grad_tol` of its largest entry."""
    import bench_tgcn as BT
    torch.manual_seed(5)
    m = BT.BatchedTGCN(2, hidden, 2)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.4, 0.4)
    p64 = {k: v.detach().double().clone().requires_grad_() for k, v in m.state_dict().items()}
    cell64 = {k[len("tgnn."):]: v for k, v in p64.items() if k.startswith("tgnn.")}
    x64 = X.double()
    h = torch.zeros(X.size(0), X.size(1), hidden, dtype=torch.float64)
    outs = []
    for t in range(X.size(-1)):
        h = F.tgcn_cell(x64[..., t], ei, ew.double(), h, cell64)
        outs.append(torch.nn.functional.linear(torch.relu(h), p64["linear.weight"], p64["linear.bias"]).unsqueeze(1))
    ref = torch.cat(outs, 1)
    BT.masked_mae_loss(ref * BT.STD + BT.MEAN, y.double() * BT.STD + BT.MEAN).backward()
    m = m.to(device)
    out = m(X.to(device), ei.to(device), ew.to(device))
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close_with_nonfinite(out, ref, atol_fwd, atol_fwd, "forward of the T-step loop")
    BT.masked_mae_loss(out * BT.STD + BT.MEAN, y.to(device) * BT.STD + BT.MEAN).backward()
    for name, p in m.named_parameters():
        g64 = p64[name].grad
        assert_close_with_nonfinite(p.grad, g64, grad_tol * float(g64.abs().max()) + 1e-9, grad_tol, name)


@pytest.mark.gpu
def test_config4_cell_kernels_row_and_column_forms_agree_on_every_row_repeatedly():
    """The fused T-GCN cell at config 4's size (400 000 rows): the row-per-lane kernels (hand-issued prefetches, hand-placed waits)
    against the column-per-lane kernels of round 4 (compiler-placed waits), EVERY element, five launches of each — a prefetched
    register consumed before its load landed shows up as a handful of rows far off (csrc/gemm_bx.hip's round-2 race was four rows in
    200 000) — and bit-identical results from launch to launch."""
    from pytorch_geometric_temporal_amd import _lib
    from pytorch_geometric_temporal_amd.ops import ptr, stream_of
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    M, Fin, O = 400_000, 2, 32
    C = Fin + O
    torch.manual_seed(5)
    AX, H, dHn = (torch.randn(M, Fin, device=dev), torch.randn(M, O, device=dev), torch.randn(M, O, device=dev))
    Wzr, bzr = torch.randn(C, 2 * O, device=dev) * 0.2, torch.randn(2 * O, device=dev) * 0.1
    Wh, bh = torch.randn(C, O, device=dev) * 0.2, torch.randn(O, device=dev) * 0.1
    nws = int(lib._pgt_tgcn_cell_bwd_ws_floats(Fin, O))

    def run():
        ZR, HT, Hn, dH = (torch.empty(M, 2 * O, device=dev), torch.empty(M, O, device=dev), torch.empty(M, O, device=dev),
                          torch.empty(M, O, device=dev))
        dW, ws = torch.empty(C * 3 * O + 3 * O, device=dev), torch.empty(nws, device=dev)
        lib.call("pgt_tgcn_cell_f32", ptr(AX), Fin, ptr(H), O, ptr(Wzr), ptr(bzr), ptr(Wh), ptr(bh), M, Fin, O, ptr(ZR), ptr(HT), ptr(Hn), O,
                 stream_of(lib, H))
        lib.call("pgt_tgcn_cell_bwd_f32", ptr(dHn), O, ptr(AX), Fin, ptr(H), O, ptr(ZR), ptr(HT), ptr(Wzr), ptr(Wh), M, Fin, O, ptr(dH), O,
                 ptr(dW[:C * 2 * O]), ptr(dW[C * 3 * O:C * 3 * O + 2 * O]), ptr(dW[C * 2 * O:C * 3 * O]), ptr(dW[C * 3 * O + 2 * O:]),
                 ptr(ws), nws, stream_of(lib, H))
        torch.cuda.synchronize()
        return ZR, HT, Hn, dH, dW
    try:
        lib.tune("tgcn_rows", 0)
        ref = run()
    finally:
        lib.tune("tgcn_rows", 1)
    first = None
    for it in range(5):
        got = run()
        for a, b, what, tol in zip(got, ref, ("Z|R", "candidate", "H'", "dH", "weight gradients"), (2e-6, 2e-6, 2e-6, 2e-5, 0.0)):
            if what == "weight gradients":
                assert_close_with_nonfinite(a, b, 2e-5 * float(b.abs().max()), 1e-5, f"{what}, launch {it}")
            else:
                d = (a - b).abs()
                assert float(d.max()) <= tol + 1e-5 * float(b.abs().max()), f"{what}, launch {it}: {int((d > tol).sum())} elements off, max {float(d.max()):.3e}"
        if first is None:
            first = got
        else:
            for a, b, what in zip(got, first, ("Z|R", "candidate", "H'", "dH", "weight gradients")):
                assert torch.equal(a, b), f"{what}: launch {it} differs from launch 0"


def test_config4_batched_tgcn_training_loop_small(backend):
    """The T-step BatchedTGCN loop (TGCN2 -> relu -> Linear per step, hidden state carried, masked-MAE) at a size the CPU
    double runs in seconds: 60 nodes, B = 3, T = 5."""
    from pytorch_geometric_temporal_amd.dataset import synthetic as syn
    ei_np, ew_np = syn.local_graph(60, 6, seed=2)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    torch.manual_seed(3)
    X, y = torch.randn(3, 60, 2, 5), torch.randn(3, 5, 60, 2)
    y[0, 0, :7] = 0.0                                              # the mask of the loss has something to mask
    _batched_tgcn_against_oracle(backend.device, ei, ew, X, y, 8, 1e-5, 2e-4)


@pytest.mark.gpu
def test_config4_batched_tgcn_training_loop_at_size():
    """configs[3] as SURVEY 8(d) defines it: 50 000 nodes / 400 000 edges, T = 12, hidden 32 (B = 2 keeps the fp64 oracle to
    a few seconds): forward 1e-5 and every gradient against the oracle."""
    import bench_tgcn as BT
    ei, ew, _, _ = BC.graph50k("local")
    series = torch.from_numpy(__import__("pytorch_geometric_temporal_amd.dataset.synthetic", fromlist=["x"]).traffic_series(40, 50_000, seed=1))
    x, y = BT.windows(series, torch.tensor([0, 9]))
    _batched_tgcn_against_oracle(torch.device("cuda:0"), ei, ew, x.contiguous(), y, 32, 1e-5, 2e-4)


# ------------------------------------------------------------------------------------------------ config 5

def _covid_signal():
    from pytorch_geometric_temporal_amd.dataset import EnglandCovidDatasetLoader
    return EnglandCovidDatasetLoader().get_dataset(lags=8)


def _run_covid(backend):
    g = load_golden("baseline_c5_evolvegcnh_covid")
    signal = _covid_signal()
    m = EvolveGCNH(129, 8)
    m.load_state_dict(g["param"], strict=True)
    m = m.to(backend.device)
    outs, wmax = [], 0.0
    with torch.no_grad():
        for snap in signal:        # the evolved weight is carried from snapshot to snapshot (evolvegcnh.py:97-100)
            outs.append(m(backend.t(snap.x), backend.t(snap.edge_index), backend.t(snap.edge_attr)))
            wmax = max(wmax, float(snap.edge_attr.max()))
    assert len(outs) == int(g["meta"]["snapshots"]) == 53
    assert wmax == float(g["meta"]["max_weight"]) > 9e5
    ref = g["out"]["out"]
    for s, o in enumerate(outs):
        assert_close_with_nonfinite(o, ref[s], 2e-5, 2e-5, f"snapshot {s}")


def test_config5_evolvegcnh_on_the_vendored_covid_graphs(backend):
    """Every snapshot of dataset/england_covid.json (a new directed graph with weights up to 9.6e5 each day) through
    EvolveGCNH(129, 8), against the reference module fed by the reference loader (evolvegcnh.py:78-102)."""
    _run_covid(backend)


def test_config5_oracle_reproduces_the_reference_fixture():
    """oracle/functional.evolvegcnh_step (TopK summary -> GRU step on the weight -> GCN propagate) over the 53 vendored
    covid snapshots against the reference module's outputs (the bench's CPU baseline for config 5 times this oracle)."""
    g = load_golden("baseline_c5_evolvegcnh_covid")
    p = g["param"]
    W = p["initial_weight"][0]
    with torch.no_grad():
        for s, snap in enumerate(_covid_signal()):
            out, W = F.evolvegcnh_step(snap.x, snap.edge_index, snap.edge_attr, W, p)
            assert_close_with_nonfinite(out, g["out"]["out"][s], 2e-5, 2e-5, f"snapshot {s}")


# ------------------------------------------------------------------------------------------------ config 1

def test_config1_chickenpox_epoch_cost_and_gradients(backend):
    """BASELINE.json configs[0]: the reference example's epoch (examples/recurrent/dcrnn_example.py:19-46) — our
    ChickenpoxDatasetLoader + temporal_signal_split + DCRNN(4, 32, 1) + torch Linear, 103 train snapshots, cost = mean
    MSE, one backward — against the cost, the first predictions and EVERY parameter gradient of the reference's own
    loader / iterator / module running the same loop."""
    from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader
    from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN
    from pytorch_geometric_temporal_amd.signal import temporal_signal_split
    g = load_golden("baseline_c1_chickenpox_epoch")

    class RecurrentGCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.recurrent = DCRNN(4, 32, 1)
            self.linear = torch.nn.Linear(32, 1)

        def forward(self, x, edge_index, edge_weight):
            return self.linear(torch.relu(self.recurrent(x, edge_index, edge_weight)))

    model = RecurrentGCN()
    model.load_state_dict(g["param"], strict=True)
    model = model.to(backend.device)
    train, _ = temporal_signal_split(ChickenpoxDatasetLoader().get_dataset(), train_ratio=0.2)
    train = train.to(backend.device)
    cost, preds, n = 0, [], 0
    for snap in train:
        y_hat = model(snap.x, snap.edge_index, snap.edge_attr)
        if n < 3:
            preds.append(y_hat.detach())
        cost = cost + torch.mean((y_hat - snap.y) ** 2)
        n += 1
    assert n == int(g["meta"]["snapshots"]) == 103
    cost = cost / n
    cost.backward()
    assert_close_with_nonfinite(torch.stack(preds), g["out"]["pred_head"], ATOL, RTOL, "first predictions")
    assert_close_with_nonfinite(cost.detach(), g["out"]["cost"], 1e-6, 1e-5, "epoch cost")
    for name, p in model.named_parameters():
        ref = g["out"]["grad/" + name]
        assert_close_with_nonfinite(p.grad, ref, 1e-6 + 1e-4 * float(ref.abs().max()), 1e-4, "grad " + name)
