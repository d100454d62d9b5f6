"""BASELINE.json configs[3] as SURVEY.md §8(d) defines it: the index-batched T-GCN training step of the reference's
examples/indexBatching/tgcn/metr_la_main.py on a 50 000-node / 400 000-edge static graph — `BatchedTGCN` (a T = 12 loop of
TGCN2(2, 32) -> relu -> Linear(32, 2), :29-47), masked-MAE on the de-normalised prediction (:49-56, :86-87), backward, one flat
gradient all-reduce, Adam (:73-90) — on B windows per GPU gathered from the HBM-resident series.

    python bench.py --config tgcn50k [--batch B] [--gpus N ...]            (bench.py dispatches here)

`value` = B * 12 * 400 000 * steps / wall snapshot-edges/s (every (sample, step) graph counts its edges once).
The same protocol as the headline (dp.timed_steps: barrier + synchronize on both sides, MAX over ranks), the same live
per-kernel-class roofline (HIP events on the launch stream around every C-ABI call during extra instrumented steps).
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from pytorch_geometric_temporal_amd import dp, ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.conv import Linear
from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2

N_NODES, DEGREE, SEQ = 50_000, 8, 12
MEAN, STD = 54.0, 19.5


class BatchedTGCN(torch.nn.Module):
    """examples/indexBatching/tgcn/metr_la_main.py:29-47, line for line in meaning: x [B, N, F, T]; per step the cell, relu, the
    per-node read-out; the T outputs concatenated to [B, T, N, out].  `dropin` keeps the read-out a torch.nn.Linear (what
    swapping the import alone gives); otherwise it is this package's Linear (same parameters, streaming kernels)."""

    def __init__(self, in_channels, hidden_dim, out_channels, dropin=False):
        super().__init__()
        self.tgnn = TGCN2(in_channels, hidden_dim, 1)
        self.linear = (torch.nn.Linear if dropin else Linear)(hidden_dim, out_channels)

    def forward(self, x, edge_index, edge_weight):
        T = x.shape[-1]
        h = None
        outs = []
        for t in range(T):
            h = self.tgnn(x[..., t], edge_index, edge_weight, h)
            outs.append(self.linear(F.relu(h)).unsqueeze(1))
        return torch.cat(outs, dim=1)


def masked_mae_loss(y_pred, y_true):
    """metr_la_main.py:49-56 (torch plumbing around the path, not part of it)."""
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = torch.abs(y_pred - y_true) * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


def make_graph(device, n=N_NODES, degree=DEGREE, kind="local"):
    gen = syn.local_graph if kind == "local" else syn.uniform_graph
    ei_np, ew_np = gen(n, degree, seed=0)
    return torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device)


def make_series(device, t_total, n=N_NODES, seed=1):
    return torch.from_numpy(syn.traffic_series(t_total, n, seed=seed)).to(device)       # resident [T, N, 2]


def windows(series, idx, seq=SEQ):
    """x [B, N, 2, T] (the example's `x.permute(0, 2, 3, 1)`, :82-84) and y [B, T, N, 2] of the windows starting at idx."""
    ar = torch.arange(seq, device=series.device)
    X = series[idx[:, None] + ar[None, :]]                  # [B, T, N, 2]
    y = series[idx[:, None] + seq + ar[None, :]]
    return X.permute(0, 2, 3, 1), y


def training_step_fn(model, flat, opt, series, ei, ew, world, seq=SEQ, graph=False):
    """forward + loss + backward + ONE flat all-reduce + update on the windows starting at `idx` (a device LongTensor).  `graph`:
    forward + loss + backward as one hipGraph and the update as a second one, the all-reduce between them eager (bench.py --graph)."""
    def forward_backward(idx):
        x, y = windows(series, idx, seq)
        out = model(x, ei, ew)
        loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
        flat.zero()
        loss.backward()
        return loss

    if graph:
        from pytorch_geometric_temporal_amd.graphed import GraphedStep
        snapshot = flat.data.clone()
        g_fb = GraphedStep(forward_backward, [torch.zeros(model.batch_hint, dtype=torch.long, device=series.device)], warmup=1)
        g_opt = GraphedStep(lambda: opt.step(), [], warmup=1)
        flat.data.copy_(snapshot)
        opt.reset()

        def step(idx):
            loss = g_fb(idx)
            flat.all_reduce_mean(world)
            g_opt()
            return loss
        return step

    def step(idx):
        loss = forward_backward(idx)
        flat.all_reduce_mean(world)
        opt.step()
        return loss
    return step


def train_run(device, rank, world, series, ei, ew, batch, steps, warmup, profile_steps=0, hidden=32, dropin=False, graph=False):
    torch.manual_seed(0)
    model = BatchedTGCN(2, hidden, 2, dropin=dropin).to(device)
    model.batch_hint = batch
    flat = dp.FlatParameters(model.parameters())
    opt_kw = {"capturable": True} if graph else {}
    opt = flat.adam(lr=1e-3)
    rng = np.random.default_rng(1000 + rank)
    n_total = warmup + steps + profile_steps
    T_total = series.shape[0]
    batches = [torch.from_numpy(rng.integers(0, T_total - 2 * SEQ, size=batch)).to(device) for _ in range(n_total + 1)]
    run = training_step_fn(model, flat, opt, series, ei, ew, world)
    snapshot = flat.data.clone()
    run(batches[n_total])                                   # initialisation pass (graph preparation, code objects, allocator)
    flat.data.copy_(snapshot)
    opt = flat.adam(lr=1e-3)
    run = training_step_fn(model, flat, opt, series, ei, ew, world, graph=graph)
    step = lambda i: run(batches[i])                        # noqa: E731
    for i in range(warmup):
        step(i)
    dt, loss = dp.timed_steps(step, warmup, steps, device)
    return dt, float(loss.detach()), step, model


def cpu_oracle(cores, batch=1, seconds=10.0):
    """The CPU oracle beside it: the same loop op for op as the reference (oracle/functional.py: three GCNConv with their own
    gcn_norm per gate and step, three Linear, the gate chain; torch autograd; Adam) on `batch` windows, `cores` threads."""
    from oracle import functional as OF
    torch.set_num_threads(cores)
    ei_np, ew_np = syn.local_graph(N_NODES, DEGREE, seed=0)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    series = torch.from_numpy(syn.traffic_series(3 * SEQ, N_NODES, seed=1))
    torch.manual_seed(0)
    m = BatchedTGCN(2, 32, 2, dropin=True)
    p = {k[len("tgnn."):]: v for k, v in m.named_parameters() if k.startswith("tgnn.")}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        idx = torch.randint(0, SEQ, (batch,))
        x, y = windows(series, idx)
        h = torch.zeros(batch, N_NODES, 32)
        outs = []
        for t in range(SEQ):
            h = OF.tgcn_cell(x[..., t], ei, ew, h, p)
            outs.append(m.linear(F.relu(h)).unsqueeze(1))
        loss = masked_mae_loss(torch.cat(outs, 1) * STD + MEAN, y * STD + MEAN)
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    step()
    per = time.perf_counter() - t0
    reps = int(max(1, min(20, seconds / max(per, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t0) / reps
    edges = int(ei.shape[1])
    return {"value": batch * SEQ * edges / dt, "unit": "snapshot-edges/s", "cores": cores, "kind": "port",
            "ms_per_step": 1e3 * dt,
            "sample": f"{reps} training steps of the same T = 12 loop on {batch} window(s) (oracle/functional.py tgcn_cell, fp32, "
                      f"torch.set_num_threads({cores}))"}


def measure(device, rank, world, batch, steps, warmup, profile_steps, series, ei, ew, bench, graph=False, dropin=False):
    """One configuration: timed steps + instrumented steps -> the dict that goes into the JSON line."""
    if graph:
        profile_steps = 0                                  # per-launch events cannot be recorded inside a captured step
    dt, loss, step, model = train_run(device, rank, world, series, ei, ew, batch, steps, warmup, profile_steps, graph=graph,
                                      dropin=dropin)
    edges = int(ei.shape[1])
    res = {"batch_per_gpu": batch, "ms_per_step": 1e3 * dt / steps, "snapshot_edges_per_s": world * batch * SEQ * edges * steps / dt,
           "final_loss": loss}
    if profile_steps > 0:
        if rank == 0:
            ops.KERNEL_TIMER = ops.KernelTimer()
        for i in range(warmup + steps, warmup + steps + profile_steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        if rank == 0:
            kernels = ops.KERNEL_TIMER.summary()
            shapes = ops.KERNEL_TIMER.by_tag()
            ops.KERNEL_TIMER = None
            bench.kernel_class_rooflines(kernels, shapes)
            for v in kernels.values():
                v["ms_per_step"] = v.pop("total_ms") / profile_steps
                v["launches_per_step"] = v.pop("launches") / profile_steps
                v["total_ms"] = v["ms_per_step"] * profile_steps
                v["launches"] = v["launches_per_step"] * profile_steps
            dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
            k = kernels[dom]
            res["roofline"] = {"kernel_class": dom, "bound": "hbm", "achieved": k["achieved_GBs"], "peak": bench.HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": k["hbm_frac"],
                               # the committed PMC passes are of the default batch (8): bytes per launch scale with it
                               **(bench.pmc_traffic(("tgcn_cell_fwd", "tgcn_cell_bwd") if dom == "tgcn_cell" else dom, bench.PMC_FILES_TGCN)
                                  if batch == 8 else {"traffic": None}),
                               "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"], "avg_us_per_launch": k["avg_us"],
                               "share_of_step_ms": k["ms_per_step"],
                               "all_kernel_classes": bench.all_kernel_classes(kernels, profile_steps)}
            bench.set_aside(res["roofline"], kernels)
            res["kernels"] = {kk: {f: v[f] for f in ("launches_per_step", "ms_per_step", "avg_us", "algorithmic_bytes_per_launch",
                                                      "achieved_GBs", "hbm_frac", "set_aside_launches", "set_aside_ms") if f in v} | (
                                   {"by_shape": v["by_shape"]} if "by_shape" in v else {}) for kk, v in kernels.items()}
    del step, model
    torch.cuda.empty_cache()
    return res


def largest_batch_within(device, series, ei, ew, budget_ms=10.0, candidates=(8, 12, 16, 24, 32, 48, 64)):
    """The largest per-GPU batch whose training step stays within `budget_ms` (short probes: 3 steps each)."""
    best = None
    for b in candidates:
        try:
            dt, _, step, model = train_run(device, 0, 1, series, ei, ew, b, 3, 1)
        except RuntimeError:                                 # out of memory: the previous candidate stands
            break
        finally:
            torch.cuda.empty_cache()
        del step, model
        if 1e3 * dt / 3 > budget_ms:
            break
        best = b
    return best


def main(args, rank, local_rank, world, device, bench):
    """`bench.py --config tgcn50k`: the driver-contract JSON line for configs[3]."""
    import json
    ei, ew = make_graph(device)
    series = make_series(device, 600)                       # [600, 50 000, 2] resident: 240 MB
    batch = args.batch if args.batch_given else 8
    scaling = "weak"
    if args.global_batch > 0:
        assert args.global_batch % world == 0
        batch, scaling = args.global_batch // world, "strong"
    res = measure(device, rank, world, batch, args.steps, args.warmup, args.profile_steps, series, ei, ew, bench, graph=args.graph)
    graphed = dropin = dropin_blas = None
    if world == 1 and not args.graph and not args.no_extra:
        graphed = bench._safe(lambda: measure(device, rank, world, batch, args.steps, args.warmup, 0, series, ei, ew, bench, graph=True))
        # import swap only: the example's own torch.nn.Linear read-out behind its relu — routed to the streaming kernels by the states
        # tensor TGCN2 returns (nn/_states.py), and with the routing off (torch's BLAS product for 32 -> 2 over B * 50 000 rows)
        dropin = bench._safe(lambda: measure(device, rank, world, batch, 10, 3, 0, series, ei, ew, bench, graph=True, dropin=True))
        TGCN2.readout_interception = False
        try:
            dropin_blas = bench._safe(lambda: measure(device, rank, world, batch, 10, 3, 0, series, ei, ew, bench, graph=True, dropin=True))
        finally:
            TGCN2.readout_interception = True
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = bench._safe(lambda: cpu_oracle(min(os.cpu_count() or 1, 32)))
    if rank == 0:
        edges = int(ei.shape[1])
        line = {"metric": "snapshot-edges aggregated/sec", "value": res["snapshot_edges_per_s"], "unit": "snapshot-edges/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": "f32 (exact fp32: every product on v_mfma_f32_32x32x2_f32, K = 34 is below the split-bf16 kernels' range)",
                "data": "synthetic",
                "config": {"workload": f"index-batched synthetic {N_NODES}-node / {edges}-edge static graph, BatchedTGCN = 12 x "
                                       "(TGCN2(2,32) -> relu -> Linear(32,2)) training step (fwd+bwd+allreduce+Adam)",
                           "batch_per_gpu": batch, "global_batch": world * batch, "seq_len": SEQ, "parallelism": f"dp{world}",
                           "hidden": 32, "graphed": bool(args.graph)},
                "variants": {"hipgraph_step": None if graphed is None else {
                    "ms_per_step": graphed["ms_per_step"], "snapshot_edges_per_s": graphed["snapshot_edges_per_s"],
                    "what": "--graph: forward+backward and the update as two hipGraphs (B = 8 is host-launch bound when issued eagerly)"},
                    "dropin_torch_linear_graphed": None if dropin is None else {
                        "ms_per_step": dropin["ms_per_step"], "what": "import swap only: the example's torch.nn.Linear(32, 2) behind its relu, routed "
                        "to the streaming kernels by the states tensor TGCN2 returns; as hipGraphs"},
                    "dropin_blas_readout_graphed": None if dropin_blas is None else {
                        "ms_per_step": dropin_blas["ms_per_step"], "what": "the same with TGCN2.readout_interception = False (torch's BLAS product)"}},
                "final_loss": res["final_loss"], "roofline": res.get("roofline"), "kernels": res.get("kernels"), "cpu_baseline": cpu}
        import bench_line
        partial = args.no_extra or args.no_cpu_baseline or args.profile_steps == 0      # (a profiler pass keeps its own file)
        bench_line.emit(line, "bench_partial_tgcn50k.json" if partial else "bench_full_tgcn50k.json")
