"""The other lines of bench.py's JSON: the launch-bound regime (the reference's own batch size, whole step as one
hipGraph) and BASELINE.json's configs 1, 3, 4, 5 — each GPU figure with the CPU oracle (op for op the reference,
forward + backward + update) timed beside it on the same host.  Everything here is bounded to a few seconds per
block; bench.py's headline line (config 2 at B = 1024) is unaffected.

    snapshot-edges/s = processed (sample, time-step) graphs x their edge count / wall time      (SURVEY.md section 8 d)
"""
import os
import time

import numpy as np
import torch

from pytorch_geometric_temporal_amd import dp, ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.graphed import GraphedStep


def _time_gpu(fn, reps, warm=3, warm_seconds=0.03):
    """Mean wall time of `reps` calls behind at least `warm` untimed ones AND `warm_seconds` of them (sub-millisecond steps: three
    replays leave the clocks where the previous block left them — bench.py's aggregation block read 2 - 4 % high that way)."""
    t0 = time.perf_counter()
    k = 0
    while k < warm or (time.perf_counter() - t0 < warm_seconds and k < 1000):
        fn()
        k += 1
        if k >= warm:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def _time_cpu(fn, seconds, min_reps=2, max_reps=50):
    fn()
    t0 = time.perf_counter()
    fn()
    per = time.perf_counter() - t0
    reps = int(max(min_reps, min(max_reps, seconds / max(per, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps, reps


def small_batch(device, make_model, masked_mae_loss, series, ei, ew, n_edges, seq, mean, std, cores, hiddens=(2, 64),
                batch=64):
    """The reference's batch size (examples/indexBatching/DCRNN/pems_bay_main.py:130-138: 64 windows): one training
    step is ~250 launches of a few microseconds, so issued one by one from Python the GPU idles; captured once as a
    hipGraph (graphed.GraphedStep) the step is one host call.  Both figures, and the CPU oracle on the same step."""
    from oracle import functional as F
    out = {}
    T_total = series.shape[0]
    ar = torch.arange(seq, device=device)
    for hidden in hiddens:
        torch.manual_seed(0)
        model = make_model(hidden).to(device)
        flat = dp.FlatParameters(model.parameters())
        opt = flat.adam(lr=1e-3)

        def step(xi, yi):
            X, y = series[xi], series[yi]
            pred = model(X, ei, ew)
            loss = masked_mae_loss(pred * std + mean, y * std + mean)
            flat.zero()
            loss.backward()
            opt.step()
            return loss

        rng = np.random.default_rng(7)
        idx = [torch.from_numpy(rng.integers(0, T_total - 2 * seq, size=batch)).to(device) for _ in range(8)]
        pairs = [(i[:, None] + ar[None, :], i[:, None] + seq + ar[None, :]) for i in idx]
        k = [0]

        def eager():
            step(*pairs[k[0] % 8])
            k[0] += 1
        t_eager = _time_gpu(eager, 20)
        graphed = GraphedStep(step, pairs[0])

        def replay():
            graphed(*pairs[k[0] % 8])
            k[0] += 1
        t_graph = _time_gpu(replay, 50)
        edges = batch * seq * n_edges
        # CPU oracle: the same step (forward, loss, backward, Adam) on `cores` threads
        torch.set_num_threads(cores)
        cm = make_model(hidden)
        cparams = {kk[len("rnn."):]: v for kk, v in cm.named_parameters() if kk.startswith("rnn.")}
        copt = torch.optim.Adam(cm.parameters(), lr=1e-3)
        cser, cei, cew = series[:2000].cpu(), ei.cpu(), ew.cpu()
        car = torch.arange(seq)

        def cpu_step():
            i = torch.randint(0, 2000 - 2 * seq, (batch,))
            X, y = cser[i[:, None] + car], cser[i[:, None] + seq + car]
            o = F.batched_dcrnn(X, cei, cew, cparams)
            if cm.head is not None:
                o = torch.nn.functional.linear(o, cm.head.weight, cm.head.bias)
            loss = masked_mae_loss(o * std + mean, y * std + mean)
            copt.zero_grad()
            loss.backward()
            copt.step()
        t_cpu, reps = _time_cpu(cpu_step, 4.0)
        out[f"hidden{hidden}"] = {
            "batch": batch, "eager_ms_per_step": 1e3 * t_eager, "graphed_ms_per_step": 1e3 * t_graph,
            "eager_snapshot_edges_per_s": edges / t_eager, "graphed_snapshot_edges_per_s": edges / t_graph,
            "cpu_oracle_ms_per_step": 1e3 * t_cpu, "cpu_oracle_snapshot_edges_per_s": edges / t_cpu, "cpu_cores": cores,
            "cpu_sample": f"{reps} steps", "epoch_time_s_23974_windows_graphed": 23974.0 / batch * t_graph}
        del graphed, model, flat, opt
    return out


def chickenpox_epoch(device, cores, K=1):
    """BASELINE.json configs[0]: ChickenpoxDatasetLoader + DCRNN(4, 32, K) + Linear(32, 1), the loop of
    examples/recurrent/dcrnn_example.py:38-46 (103 train snapshots, one backward per epoch, Adam): GPU eager, GPU as one
    hipGraph per epoch, and the CPU oracle.  K = 1 is the example's own model (one launch per snapshot each way:
    csrc/small_cell.hip); K = 2, 3 are the reference's test shapes (test/recurrent_test.py:274-315) — the hops, both gate
    products and the blend in ONE workgroup, csrc/seq_small.hip."""
    import torch.nn.functional as TF
    from oracle import functional as F
    from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader
    from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN
    from pytorch_geometric_temporal_amd.signal import temporal_signal_split

    class RecurrentGCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.recurrent = DCRNN(4, 32, K)
            self.linear = torch.nn.Linear(32, 1)

        def forward(self, x, edge_index, edge_weight):
            return self.linear(TF.relu(self.recurrent(x, edge_index, edge_weight)))

    dataset = ChickenpoxDatasetLoader().get_dataset()
    train, _ = temporal_signal_split(dataset, train_ratio=0.2)
    cpu_snaps = [(s.x, s.edge_index, s.edge_attr, s.y) for s in train]
    train = train.to(device)
    snaps = [(s.x, s.edge_index, s.edge_attr, s.y) for s in train]
    torch.manual_seed(0)
    model = RecurrentGCN().to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)

    def epoch():
        cost = 0
        for x, e, w, y in snaps:
            cost = cost + torch.mean((model(x, e, w) - y) ** 2)
        cost = cost / len(snaps)
        for p in model.parameters():
            p.grad.zero_()
        cost.backward()
        opt.step()
        return cost
    t_eager = _time_gpu(epoch, 5, warm=2)
    graphed = GraphedStep(epoch, [])
    t_graph = _time_gpu(lambda: graphed(), 20)
    torch.set_num_threads(cores)
    cm = RecurrentGCN()
    cp = {k[len("recurrent."):]: v for k, v in cm.named_parameters() if k.startswith("recurrent.")}
    copt = torch.optim.Adam(cm.parameters(), lr=0.01)

    def cpu_epoch():
        cost = 0
        for x, e, w, y in cpu_snaps:
            h = F.dcrnn_cell(x, e, w, None, cp)
            cost = cost + torch.mean((cm.linear(TF.relu(h)) - y) ** 2)
        cost = cost / len(cpu_snaps)
        copt.zero_grad()
        cost.backward()
        copt.step()
    t_cpu, reps = _time_cpu(cpu_epoch, 3.0)
    E = int(snaps[0][1].shape[1])
    n = len(snaps)
    return {"what": f"Chickenpox DCRNN(4,32,K={K})+Linear, 1 epoch = 103 snapshots, full-batch backward, Adam",
            "gpu_eager_ms_per_epoch": 1e3 * t_eager, "gpu_graphed_ms_per_epoch": 1e3 * t_graph,
            "cpu_oracle_ms_per_epoch": 1e3 * t_cpu, "cpu_cores": cores, "cpu_sample": f"{reps} epochs",
            "snapshot_edges_per_s_graphed": n * E / t_graph, "snapshot_edges_per_s_cpu": n * E / t_cpu,
            "gpu_wins": bool(t_graph < t_cpu)}


def covid_epoch(device, cores):
    """BASELINE.json configs[4]: the vendored England-Covid mobility graphs (a new edge list every snapshot) through
    EvolveGCNH(129, 8) + Linear, the loop of examples/recurrent/evolvegcnh_example.py:38-50 (one backward per epoch)."""
    import torch.nn.functional as TF
    from oracle import functional as F
    from pytorch_geometric_temporal_amd.dataset import EnglandCovidDatasetLoader
    from pytorch_geometric_temporal_amd.nn.recurrent import EvolveGCNH

    class RecurrentGCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.recurrent = EvolveGCNH(129, 8)
            self.linear = torch.nn.Linear(8, 1)

        def forward(self, x, edge_index, edge_weight):
            return self.linear(TF.relu(self.recurrent(x, edge_index, edge_weight)))

    ds = EnglandCovidDatasetLoader().get_dataset(lags=8)
    cpu_snaps = [(s.x, s.edge_index, s.edge_attr, s.y) for s in ds]
    snaps = [tuple(t.to(device) for t in s) for s in cpu_snaps]
    torch.manual_seed(0)
    model = RecurrentGCN().to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)

    def epoch():
        model.recurrent.reinitialize_weight()
        cost = 0
        for x, e, w, y in snaps:
            cost = cost + torch.mean((model(x, e, w).view(-1) - y) ** 2)
        cost = cost / len(snaps)
        for p in model.parameters():
            p.grad.zero_()
        cost.backward()
        opt.step()
        return cost
    t_eager = _time_gpu(epoch, 5, warm=2)          # edge lists are range-checked once per tensor, nothing else is prepared

    # a STREAM of new graphs (fresh edge tensors every epoch: nothing is remembered between snapshots), through the
    # one-launch small-graph layer and through the prepared-operator path (device sorts + normalisation per new edge list)
    def fresh_epoch():
        nonlocal snaps
        keep = snaps
        snaps = [(x, e.clone(), w.clone(), y) for x, e, w, y in keep]
        try:
            epoch()
        finally:
            snaps = keep
    t_fresh = _time_gpu(fresh_epoch, 3, warm=1)
    fits = ops.gcn_small_fits
    try:
        ops.gcn_small_fits = lambda *a: False
        t_fresh_prepared = _time_gpu(fresh_epoch, 3, warm=1)
    finally:
        ops.gcn_small_fits = fits
    t_graph = None
    try:
        graphed = GraphedStep(epoch, [])
        t_graph = _time_gpu(lambda: graphed(), 20)
    except Exception as e:                          # an auxiliary line must never cost the bench line
        t_graph_err = repr(e)
    # the same epoch captured with the 53 edge lists (index + weight tensors) as GRAPH INPUTS and replayed on fresh copies of
    # them every epoch: what a stream of new graphs of these sizes costs as one hipGraph (the GCN layer reads the RAW edge list
    # inside its kernel, so nothing is prepared on the host per new list).  (The memory access fault of round 4 at this point was
    # the SECOND capture of a process whose first GraphedStep still held its loss and, through it, the parameters' AccumulateGrad
    # nodes on another stream: graphed.py now hands out detached outputs.  scripts/covid_fault_hunt.sh repeats this block.)
    t_graph_fresh = None
    try:
        def epoch_on(*edges):
            nonlocal snaps
            keep = snaps
            snaps = [(x, edges[2 * i], edges[2 * i + 1], y) for i, (x, _, _, y) in enumerate(keep)]
            try:
                return epoch()
            finally:
                snaps = keep
        flat_edges = [t for (_, e, w, _) in snaps for t in (e, w)]
        graphed_e = GraphedStep(epoch_on, flat_edges, warmup=2)
        fresh = [t.clone() for t in flat_edges]
        t_graph_fresh = _time_gpu(lambda: graphed_e(*fresh), 20)
        # (the parameters move with every replay, so there is no eager figure to compare a replayed cost with: finite is the check)
        cost_replay = float(graphed_e(*fresh))
        if not (cost_replay == cost_replay and abs(cost_replay) < 1e30):
            t_graph_fresh = f"non-finite cost from the replay: {cost_replay}"
        del graphed_e, fresh
    except Exception as e:
        t_graph_fresh = repr(e)
    torch.set_num_threads(cores)
    cm = RecurrentGCN()
    cp = {k[len("recurrent."):]: v for k, v in cm.named_parameters() if k.startswith("recurrent.")}
    copt = torch.optim.Adam(cm.parameters(), lr=0.01)

    def cpu_epoch():
        W = cp["initial_weight"][0]
        cost = 0
        for x, e, w, y in cpu_snaps:
            h, W = F.evolvegcnh_step(x, e, w, W, cp)
            cost = cost + torch.mean((cm.linear(TF.relu(h)).view(-1) - y) ** 2)
        cost = cost / len(cpu_snaps)
        copt.zero_grad()
        cost.backward()
        copt.step()
    t_cpu, reps = _time_cpu(cpu_epoch, 3.0)
    edges = sum(int(s[1].shape[1]) for s in cpu_snaps)
    res = {"what": "EvolveGCNH(129,8)+Linear on the vendored england_covid graphs, 1 epoch = 53 snapshots, full-batch backward",
           "gpu_eager_ms_per_epoch": 1e3 * t_eager, "cpu_oracle_ms_per_epoch": 1e3 * t_cpu, "cpu_cores": cores,
           "cpu_sample": f"{reps} epochs", "snapshot_edges_per_s_eager": edges / t_eager,
           "snapshot_edges_per_s_cpu": edges / t_cpu,
           "gpu_eager_ms_per_epoch_new_edge_tensors": 1e3 * t_fresh,
           "gpu_eager_ms_per_epoch_new_edge_tensors_prepared_operator_path": 1e3 * t_fresh_prepared,
           "gpu_graphed_ms_per_epoch_new_edge_tensors": (1e3 * t_graph_fresh if isinstance(t_graph_fresh, float) else t_graph_fresh)}
    if t_graph is not None:
        res.update({"gpu_graphed_ms_per_epoch": 1e3 * t_graph, "snapshot_edges_per_s_graphed": edges / t_graph,
                    "gpu_wins": bool(t_graph < t_cpu)})
    else:
        res.update({"gpu_graphed_error": t_graph_err, "gpu_wins": bool(t_eager < t_cpu)})
    return res


def config3_pemsbay(device, cores, batch=64):
    """BASELINE.json configs[2]: PeMS-BAY-shaped (325 nodes / 2 694 edges) A3TGCN2(2, 32, periods=12), forward + backward."""
    from oracle import functional as F
    from pytorch_geometric_temporal_amd.nn.recurrent import A3TGCN2
    ei_np, ew_np = syn.sensor_graph(325, 2694, seed=0)
    ei, ew = torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device)
    torch.manual_seed(0)
    m = A3TGCN2(2, 32, 12, batch).to(device)
    X = torch.randn(batch, 325, 2, 12, device=device)

    def step():
        m.zero_grad(set_to_none=False)
        m(X, ei, ew).square().mean().backward()
    m(X, ei, ew).square().mean().backward()
    t_eager = _time_gpu(step, 20)
    graphed = GraphedStep(lambda: (step(), X)[1], [])
    t_graph = _time_gpu(lambda: graphed(), 50)
    torch.set_num_threads(cores)
    p = {k: v.detach().cpu().clone().requires_grad_() for k, v in m.state_dict().items()}
    Xc, eic, ewc = X.cpu(), ei.cpu(), ew.cpu()

    def cpu_step():
        for v in p.values():
            v.grad = None
        F.a3tgcn(Xc, eic, ewc, None, p).square().mean().backward()
    t_cpu, reps = _time_cpu(cpu_step, 4.0)
    edges = batch * 12 * 2694
    return {"what": f"A3TGCN2(2,32,periods=12) PeMS-BAY-shaped 325 nodes / 2694 edges, B={batch}, forward+backward",
            "gpu_eager_ms": 1e3 * t_eager, "gpu_graphed_ms": 1e3 * t_graph, "cpu_oracle_ms": 1e3 * t_cpu,
            "cpu_cores": cores, "cpu_sample": f"{reps} steps", "snapshot_edges_per_s_graphed": edges / t_graph,
            "snapshot_edges_per_s_eager": edges / t_eager, "snapshot_edges_per_s_cpu": edges / t_cpu}


def config4_50k(device, cores, bench, batch=8):
    """BASELINE.json configs[3] as SURVEY 8(d) defines it (bench_tgcn.py): the T = 12 BatchedTGCN training step (12 x TGCN2(2, 32)
    -> relu -> Linear(32, 2), masked MAE, Adam) on 50 000 nodes / 400 000 edges at B = 8 per GPU and at the largest B whose step
    fits 10 ms, each with its per-kernel-class roofline, the CPU oracle on the same loop beside them; plus the single cell
    forward + backward the earlier rounds reported (continuity)."""
    import bench_tgcn as BT
    from oracle import functional as F
    from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2
    ei, ew = BT.make_graph(device)
    series = BT.make_series(device, 600)
    out = {"what": "BatchedTGCN = 12 x (TGCN2(2,32) -> relu -> Linear(32,2)), 50 000 nodes / 400 000 edges, training step "
                   "(fwd + bwd + Adam), examples/indexBatching/tgcn/metr_la_main.py:29-47,73-90"}
    r = BT.measure(device, 0, 1, batch, 6, 2, 1, series, ei, ew, bench)
    out.update({"ms_per_step": r["ms_per_step"], "snapshot_edges_per_s": r["snapshot_edges_per_s"], "batch_per_gpu": batch,
                "roofline": r.get("roofline"), "kernels": r.get("kernels")})
    try:                                                   # B = 8 is host-launch bound when issued eagerly: the same step as hipGraphs
        rg = BT.measure(device, 0, 1, batch, 10, 2, 0, series, ei, ew, bench, graph=True)
        out["graphed"] = {"ms_per_step": rg["ms_per_step"], "snapshot_edges_per_s": rg["snapshot_edges_per_s"]}
    except Exception as e:
        out["graphed"] = {"error": repr(e)}
    big = BT.largest_batch_within(device, series, ei, ew, 10.0)
    if big is not None and big != batch:
        rb = BT.measure(device, 0, 1, big, 6, 2, 1, series, ei, ew, bench)
        out["largest_batch_within_10ms"] = {"batch_per_gpu": big, "ms_per_step": rb["ms_per_step"],
                                            "snapshot_edges_per_s": rb["snapshot_edges_per_s"], "roofline": rb.get("roofline"),
                                            "kernels": rb.get("kernels")}
    try:                                                   # the example's own default (metr_la_main.py:18: --batch-size 64), once
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        r64 = BT.measure(device, 0, 1, 64, 3, 1, 1, series, ei, ew, bench)
        out["batch_64_reference_default"] = {"batch_per_gpu": 64, "ms_per_step": r64["ms_per_step"],
                                             "snapshot_edges_per_s": r64["snapshot_edges_per_s"], "roofline": r64.get("roofline"),
                                             "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}
    except Exception as e:
        out["batch_64_reference_default"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    cpu = BT.cpu_oracle(cores, batch=1, seconds=6.0)
    out["cpu_oracle"] = cpu
    out["snapshot_edges_per_s_cpu"] = cpu["value"]
    del series
    torch.cuda.empty_cache()
    # the single cell (forward + backward at B = 8) of rounds 1 - 3
    torch.manual_seed(0)
    m = TGCN2(2, 32, batch).to(device)
    X, H = torch.randn(batch, 50_000, 2, device=device), torch.randn(batch, 50_000, 32, device=device)

    def step():
        m.zero_grad(set_to_none=False)
        m(X, ei, ew, H).square().mean().backward()
    m(X, ei, ew, H).square().mean().backward()
    out["one_cell_fwd_bwd_ms"] = 1e3 * _time_gpu(step, 20)
    return out
