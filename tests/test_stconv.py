"""The dense halves of the ST-Conv block on csrc/tconv.hip: `TemporalConv.forward` (stgcn.py:27-44) as one
row-shifted K-segmented product with the gate on the accumulators, `BatchNorm2d(num_nodes)` (stgcn.py:129, :156-159) on the
[B, T, N, C] layout, and the whole `STConv` — forward AND every gradient — against the reference's own module file at the
shape the reference's test uses (test/attention_test.py:140-176: [10, 5, 300, 100], hidden 8, K = 2)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close_with_nonfinite, load_golden
from oracle import ref_import as R
from pytorch_geometric_temporal_amd import ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.attention import STConv, TemporalConv


def _torch_temporal_conv(X, convs):
    """stgcn.py:36-44 spelled with torch ops (fp64 when the inputs are)."""
    Xp = X.permute(0, 3, 2, 1)
    P, Q, Rr = (F.conv2d(Xp, w, b) for w, b in convs)
    return F.relu(P * torch.sigmoid(Q) + Rr).permute(0, 3, 2, 1)


@pytest.mark.parametrize("B,T,N,Cin,Cout,k", [
    (2, 5, 7, 1, 3, 2),          # one input channel (speed only), lanes along the rows
    (3, 6, 10, 4, 8, 3),         # float4 loads along k
    (2, 4, 33, 100, 8, 2),       # the reference test's first convolution: K = 200 > one 32-deep tile, ragged rows
    (2, 7, 70, 6, 40, 3),        # two column blocks, odd-ish Cin
    (1, 3, 5, 2, 70, 3),         # T' = 1, three column blocks' worth of channels (grid.y = 2)
    (2, 3, 4, 3, 5, 1),          # k = 1: no shift at all; odd K
    (1, 2, 130, 8, 64, 2),       # more than one row tile
])
def test_temporal_conv_matches_conv2d_chain(backend, B, T, N, Cin, Cout, k):
    torch.manual_seed(B * 100 + Cin)
    m = TemporalConv(Cin, Cout, k)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.5, 0.5)
    X = torch.randn(B, T, N, Cin)
    Wt = torch.randn(B, T - k + 1, N, Cout)
    X64 = X.double().requires_grad_()
    convs64 = [(c.weight.detach().double().requires_grad_(), c.bias.detach().double().requires_grad_())
               for c in (m.conv_1, m.conv_2, m.conv_3)]
    ref = _torch_temporal_conv(X64, convs64)
    (ref * Wt.double()).sum().backward()

    m = m.to(backend.device)
    Xo = backend.t(X).requires_grad_()
    out = m(Xo)
    assert out.shape == (B, T - k + 1, N, Cout) and out.is_contiguous()
    assert_close_with_nonfinite(out, ref, 1e-5, 1e-5, "TemporalConv forward")
    (out * backend.t(Wt)).sum().backward()
    assert_close_with_nonfinite(Xo.grad, X64.grad, 3e-5, 3e-5, "dX")
    for c, (w64, b64), name in zip((m.conv_1, m.conv_2, m.conv_3), convs64, ("conv_1", "conv_2", "conv_3")):
        scale = float(w64.grad.abs().max()) + 1e-12
        assert float((c.weight.grad.cpu().double() - w64.grad).abs().max()) <= 3e-5 * max(scale, 1.0), name + ".weight"
        assert_close_with_nonfinite(c.bias.grad, b64.grad, 3e-5 * max(scale, 1.0), 3e-5, name + ".bias")
    with torch.no_grad():                                # inference: P / sigmoid(Q) are not written
        assert torch.equal(m(backend.t(X)), out.detach())


def test_temporal_conv_input_without_gradient_and_rejections(backend):
    m = TemporalConv(3, 4, 2).to(backend.device)
    X = backend.t(torch.randn(2, 4, 5, 3))
    m(X).sum().backward()                                # the input gradient product is skipped
    assert m.conv_3.weight.grad is not None and X.grad is None
    with pytest.raises(RuntimeError, match="kernel size"):
        m(backend.t(torch.randn(2, 1, 5, 3)))
    with pytest.raises(ValueError, match="batch, time, nodes, channels"):
        m(backend.t(torch.randn(4, 5, 3)))
    Xs = backend.t(torch.randn(2, 5, 4, 3)).transpose(1, 2)      # a strided view is taken as what it denotes
    with torch.no_grad():
        assert torch.equal(m(Xs), m(Xs.contiguous()))


@pytest.mark.parametrize("B,T,N,C", [(3, 2, 5, 4), (2, 3, 7, 10), (4, 1, 3, 1), (2, 2, 6, 66)])
def test_batchnorm_over_nodes_matches_torch(backend, B, T, N, C):
    torch.manual_seed(N)
    bn_ref = torch.nn.BatchNorm2d(N).double()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.uniform_(-0.5, 0.5)
    bn = torch.nn.BatchNorm2d(N)
    bn.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in bn_ref.state_dict().items()})
    bn = bn.to(backend.device)
    Wt = torch.randn(B, T, N, C)
    for step in range(2):                                # two training steps: the running statistics move twice
        X = torch.randn(B, T, N, C) * 2.0 + 3.0
        X64 = X.double().requires_grad_()
        ref = bn_ref(X64.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
        (ref * Wt.double()).sum().backward()
        Xo = backend.t(X).requires_grad_()
        out = ops.batch_norm_nodes(Xo, bn, True)
        assert_close_with_nonfinite(out, ref, 2e-5, 2e-5, "train forward")
        bn.zero_grad()
        (out * backend.t(Wt)).sum().backward()
        assert_close_with_nonfinite(Xo.grad, X64.grad, 1e-4, 1e-4, "train dX")
        assert_close_with_nonfinite(bn.weight.grad, bn_ref.weight.grad, 2e-4, 1e-4, "dgamma")
        assert_close_with_nonfinite(bn.bias.grad, bn_ref.bias.grad, 2e-4, 1e-4, "dbeta")
        bn_ref.zero_grad()
        assert_close_with_nonfinite(bn.running_mean, bn_ref.running_mean, 1e-5, 1e-5, "running_mean")
        assert_close_with_nonfinite(bn.running_var, bn_ref.running_var, 1e-5, 1e-5, "running_var")
        assert int(bn.num_batches_tracked) == step + 1
    bn_ref.eval()
    X = torch.randn(B, T, N, C)
    X64 = X.double().requires_grad_()
    ref = bn_ref(X64.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    (ref * Wt.double()).sum().backward()
    Xo = backend.t(X).requires_grad_()
    out = ops.batch_norm_nodes(Xo, bn, False)
    assert_close_with_nonfinite(out, ref, 2e-5, 2e-5, "eval forward")
    bn.zero_grad()
    (out * backend.t(Wt)).sum().backward()
    assert_close_with_nonfinite(Xo.grad, X64.grad, 1e-4, 1e-4, "eval dX")
    assert_close_with_nonfinite(bn.weight.grad, bn_ref.weight.grad, 2e-4, 1e-4, "eval dgamma")
    assert int(bn.num_batches_tracked) == 2              # evaluation leaves the bookkeeping alone


def test_batchnorm_cumulative_average_and_single_value(backend):
    bn = torch.nn.BatchNorm2d(3, momentum=None).to(backend.device)
    ref = torch.nn.BatchNorm2d(3, momentum=None)
    for _ in range(3):
        X = torch.randn(2, 2, 3, 4) + 1.0
        ops.batch_norm_nodes(backend.t(X), bn, True)
        ref(X.permute(0, 2, 1, 3))
    assert_close_with_nonfinite(bn.running_mean, ref.running_mean, 1e-5, 1e-5, "cumulative running_mean")
    assert_close_with_nonfinite(bn.running_var, ref.running_var, 1e-5, 1e-5, "cumulative running_var")
    with pytest.raises(ValueError, match="more than 1 value"):
        ops.batch_norm_nodes(backend.t(torch.randn(1, 1, 3, 1)), bn, True)


def _reference_test_case(seed=0, B=10, T=5):
    """test/attention_test.py:140-176 (`create_mock_batch`: 300 nodes, 15 edges per node, weighted)."""
    rng = np.random.default_rng(seed)
    n, per = 300, 15
    src = np.repeat(np.arange(n), per)
    dst = rng.integers(0, n, size=n * per)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    ew = torch.from_numpy(rng.uniform(0.1, 1.0, size=n * per).astype(np.float32))
    torch.manual_seed(seed)
    X = torch.rand(B, T, n, 100)
    return X, ei, ew


class _ContiguousInput(torch.nn.Module):
    """torch's CPU BatchNorm2d BACKWARD returns a wrong input gradient for the strides the reference hands it when T' = 1
    (a permuted view of a permuted convolution output with a size-1 axis: `test_torch_cpu_batchnorm_backward_...` below shows
    torch disagreeing with itself).  The reference's test shape has T' = 5 - 2 (3 - 1) = 1, so the reference module's batch
    norm is fed the same values contiguously: no change of meaning, and its state_dict keys stay (`_batch_norm.bn.*`
    is never read back)."""

    def __init__(self, bn):
        super().__init__()
        self.bn = bn

    def forward(self, x):
        return self.bn(x.contiguous())


def test_torch_cpu_batchnorm_backward_disagrees_with_itself_on_the_reference_strides_at_one_output_step():
    """Why the test below feeds the reference's batch norm contiguously when T' = 1 (and only then): documented, not hidden.
    If a later torch fixes this the assertion on T' = 1 flips and the wrapper can go."""
    torch.manual_seed(0)
    diffs = {}
    for Tp in (1, 3):
        bn = torch.nn.BatchNorm2d(30).double().eval()
        base = torch.randn(4, 6, 30, Tp, dtype=torch.float64)                    # Conv2d output [B, C, N, T']
        Wt = torch.randn(4, Tp, 30, 6, dtype=torch.float64)
        a, b = base.clone().requires_grad_(), base.clone().requires_grad_()
        (bn(a.permute(0, 3, 2, 1).permute(0, 2, 1, 3)).permute(0, 2, 1, 3) * Wt).sum().backward()
        (bn(b.permute(0, 3, 2, 1).permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3) * Wt).sum().backward()
        diffs[Tp] = float((a.grad - b.grad).abs().max())
    assert diffs[3] == 0.0
    if diffs[1] == 0.0:
        pytest.skip("this torch build computes the T' = 1 case correctly: _ContiguousInput is no longer needed")


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not mounted")
@pytest.mark.parametrize("B,T,train", [(10, 5, True), (4, 5, False), (3, 7, True)])      # (eval mode on a smaller batch: the CPU double is slow)
def test_stconv_forward_and_every_gradient_at_the_reference_test_shape(emu_backend, B, T, train):
    """STConv(300, 100, 8, 10, kernel_size 3, K = 2) on [10, 5, 300, 100] (and on 7 steps, where torch's batch norm can be
    used as the reference calls it): forward, dX and every parameter gradient against autograd through the reference's OWN
    stgcn.py (fp64 copy of the same parameters), train-mode batch norm included."""
    X, ei, ew = _reference_test_case(B=B, T=T)
    torch.manual_seed(1)
    ref = R.load("nn.attention.stgcn").STConv(300, 100, 8, 10, 3, 2)
    with torch.no_grad():
        for p in ref.parameters():
            p.uniform_(-0.3, 0.3)
        ref._batch_norm.weight.uniform_(0.5, 1.5)
    our = STConv(300, 100, 8, 10, 3, 2)
    our.load_state_dict(ref.state_dict(), strict=True)
    ref = ref.double().train(train)
    our = our.to(emu_backend.device).train(train)
    ref_bn = ref._batch_norm
    if T - 4 == 1:
        ref._batch_norm = _ContiguousInput(ref_bn)
    Wt = torch.randn(B, T - 4, 300, 10)
    Xr = X.double().requires_grad_()
    out_r = ref(Xr, ei, ew.double())
    (out_r * Wt.double()).sum().backward()
    Xo = emu_backend.t(X).requires_grad_()
    out = our(Xo, emu_backend.t(ei), emu_backend.t(ew))
    assert out.shape == (B, T - 4, 300, 10)
    assert_close_with_nonfinite(out, out_r, 1e-4, 1e-4, "forward")
    (out * emu_backend.t(Wt)).sum().backward()
    gscale = float(Xr.grad.abs().max())
    assert_close_with_nonfinite(Xo.grad, Xr.grad, 2e-4 * gscale, 2e-4, "dX")
    refp = {n.replace("_batch_norm.bn.", "_batch_norm."): p for n, p in ref.named_parameters()}
    for name, p in our.named_parameters():
        g = refp[name].grad
        assert p.grad is not None, name
        assert_close_with_nonfinite(p.grad, g, 2e-4 * float(g.abs().max()) + 1e-7, 2e-4, name)
    if train:
        assert_close_with_nonfinite(our._batch_norm.running_mean, ref_bn.running_mean, 1e-5, 1e-5, "running_mean")
        assert_close_with_nonfinite(our._batch_norm.running_var, ref_bn.running_var, 1e-5, 1e-4, "running_var")


def test_stconv_training_step_matches_reference_fixture_with_every_gradient(backend):
    """STConv in training mode against tests/golden/stconv_sensor_grads.npz (the reference's own stgcn.py under torch autograd,
    oracle/make_golden.py): forward, dX, every parameter gradient and the updated batch-norm running statistics — on the
    product library (`-m gpu`) as on the CPU double, no detour through either."""
    g = load_golden("stconv_sensor_grads")
    X, ei, ew, G = (backend.t(g["in"][k]) for k in ("X", "edge_index", "edge_weight", "G"))
    m = STConv(30, 4, 8, 6, kernel_size=3, K=int(g["meta"]["K"]), normalization="sym")
    m.load_state_dict(g["param"], strict=True)
    with torch.no_grad():                                    # the fixture's state_dict was taken AFTER the step: back to a fresh module's
        m._batch_norm.running_mean.zero_()
        m._batch_norm.running_var.fill_(1.0)
        m._batch_norm.num_batches_tracked.zero_()
    m = m.to(backend.device).train()
    Xd = X.clone().requires_grad_()
    out = m(Xd, ei, ew)
    assert_close_with_nonfinite(out, g["out"]["out_train"], 2e-5, 2e-5, "forward (train mode)")
    (out * G).sum().backward()
    assert_close_with_nonfinite(Xd.grad, g["out"]["grad_X"], 1e-4 * float(g["out"]["grad_X"].abs().max()), 1e-4, "dX")
    for name, p in m.named_parameters():
        ref = g["out"]["grad_" + name]
        assert p.grad is not None, name
        assert_close_with_nonfinite(p.grad, ref, 1e-4 * float(ref.abs().max()) + 1e-7, 1e-4, name)
    assert_close_with_nonfinite(m._batch_norm.running_mean, g["out"]["running_mean"], 1e-5, 1e-5, "running_mean")
    assert_close_with_nonfinite(m._batch_norm.running_var, g["out"]["running_var"], 1e-5, 1e-4, "running_var")
    assert int(m._batch_norm.num_batches_tracked) == 1


@pytest.mark.gpu
def test_stconv_at_the_reference_test_shape_on_the_gpu_matches_the_cpu_double():
    """The same block at the same shape on the MI355X against the kernels' CPU double (which the test above pins to the
    reference's module file): forward and every gradient, train mode."""
    from conftest import get_emu_lib
    from pytorch_geometric_temporal_amd import _lib
    X, ei, ew = _reference_test_case()
    torch.manual_seed(1)
    m = STConv(300, 100, 8, 10, 3, 2)
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.3, 0.3)
        m._batch_norm.weight.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    Wt = torch.randn(10, 1, 300, 10)

    def run(device):
        mm = STConv(300, 100, 8, 10, 3, 2)
        mm.load_state_dict(state)
        mm = mm.to(device).train()
        Xd = X.to(device).requires_grad_()
        out = mm(Xd, ei.to(device), ew.to(device))
        (out * Wt.to(device)).sum().backward()
        return out.detach().cpu(), Xd.grad.cpu(), {n: p.grad.cpu() for n, p in mm.named_parameters()}, mm._batch_norm.running_var.cpu()

    ops.GRAPH_CACHE.clear()
    out_g, dx_g, gp_g, rv_g = run("cuda:0")
    ops.GRAPH_CACHE.clear()
    _lib._set_library_for_testing(get_emu_lib())
    try:
        out_c, dx_c, gp_c, rv_c = run("cpu")
    finally:
        _lib._set_library_for_testing(None)
        ops.GRAPH_CACHE.clear()
    assert_close_with_nonfinite(out_g, out_c, 2e-5, 2e-5, "forward")
    assert_close_with_nonfinite(dx_g, dx_c, 1e-4 * float(dx_c.abs().max()), 1e-4, "dX")
    for n in gp_c:
        assert_close_with_nonfinite(gp_g[n], gp_c[n], 1e-4 * float(gp_c[n].abs().max()) + 1e-7, 1e-4, n)
    assert_close_with_nonfinite(rv_g, rv_c, 1e-5, 1e-5, "running_var")


def test_frozen_batch_norm_and_double_modules_follow_the_reference(emu_backend):
    """(a) `model._batch_norm.eval()` inside a training STConv: running statistics are used and left alone, as torch's BatchNorm2d
    does in the reference (round 4's advisor: the block's own flag was consulted).  (b) a `.double()` TemporalConv runs (torch's
    convolutions, stgcn.py:36-44) and agrees with the fp32 kernel path."""
    torch.manual_seed(3)
    ei_np, ew_np = syn.sensor_graph(12, 50, seed=4, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    X = torch.randn(2, 6, 12, 3)
    m = STConv(12, 3, 4, 5, 3, 2)
    with torch.no_grad():
        m._batch_norm.running_mean.uniform_(-0.5, 0.5)
        m._batch_norm.running_var.uniform_(0.5, 1.5)
    rm, rv = m._batch_norm.running_mean.clone(), m._batch_norm.running_var.clone()
    m.train()
    m._batch_norm.eval()
    out_frozen = m(X, ei, ew)
    assert torch.equal(m._batch_norm.running_mean, rm) and torch.equal(m._batch_norm.running_var, rv)
    assert int(m._batch_norm.num_batches_tracked) == 0
    m.eval()
    assert_close_with_nonfinite(out_frozen, m(X, ei, ew), 1e-6, 1e-6, "frozen batch norm = eval-mode statistics")
    m.train()
    out_train = m(X, ei, ew)
    assert not torch.equal(m._batch_norm.running_mean, rm) and float((out_train - out_frozen).detach().abs().max()) > 1e-3
    tc = TemporalConv(3, 4, 3)
    ref32 = tc(X)
    tc64 = TemporalConv(3, 4, 3).double()
    tc64.load_state_dict({k: v.double() for k, v in tc.state_dict().items()})
    out64 = tc64(X.double())
    assert out64.dtype == torch.float64
    assert_close_with_nonfinite(ref32, out64, 1e-5, 1e-5, "double TemporalConv vs the fp32 kernels")
