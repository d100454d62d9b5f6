"""Round-4 probe: the fused T-GCN cell kernels (forward / adjoint separately, HIP events), the one-workgroup DCRNN sequence
kernels at the reference's batch size, both against the paths they replace.  One JSON line per measurement."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd import dp, ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.graphed import GraphedStep  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2, BatchedDCRNN  # noqa: E402


def ev_time(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def tgcn_cell(dev):
    ei_np, ew_np = syn.local_graph(50_000, 8, seed=0)
    ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
    for B in (8, 32):
        for fused in (True,):
            ops.USE_TGCN_FUSED = fused
            torch.manual_seed(0)
            m = TGCN2(2, 32, B).to(dev)
            X, H = torch.randn(B, 50_000, 2, device=dev), torch.randn(B, 50_000, 32, device=dev).requires_grad_()
            w = torch.randn(B, 50_000, 32, device=dev)
            with torch.no_grad():
                f_inf = ev_time(lambda: m(X, ei, ew, H))
            f_train = ev_time(lambda: m(X, ei, ew, H))
            out = m(X, ei, ew, H)

            def bwd():
                m.zero_grad(set_to_none=True)
                H.grad = None
                out.backward(w, retain_graph=True)
            b = ev_time(bwd)
            print(json.dumps({"tgcn_cell": {"B": B, "rows": B * 50_000, "fused": fused, "fwd_inference_us": f_inf, "fwd_training_us": f_train,
                                            "bwd_us": b}}), flush=True)
    ops.USE_TGCN_FUSED = True


def small_batch(dev):
    import bench
    ei_np, ew_np = syn.sensor_graph(207, 1515, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
    series = torch.from_numpy(syn.traffic_series(4000, 207, seed=1)).to(dev)
    ar = torch.arange(12, device=dev)
    for hidden in (2, 8):
        for small in (True,):
            ops.USE_SEQ_SMALL = small
            torch.manual_seed(0)
            model = bench.Model(hidden).to(dev) if hidden == 2 else bench.Model(hidden).to(dev)
            flat = dp.FlatParameters(model.parameters())
            opt = flat.optimizer(torch.optim.Adam, lr=1e-3, capturable=True)

            def step(xi, yi):
                X, y = series[xi], series[yi]
                pred = model(X, ei, ew)
                loss = bench.masked_mae_loss(pred * bench.STD + bench.MEAN, y * bench.STD + bench.MEAN)
                flat.zero()
                loss.backward()
                opt.step()
                return loss
            idx = torch.randint(0, 3000, (64,), device=dev)
            pair = (idx[:, None] + ar[None, :], idx[:, None] + 12 + ar[None, :])
            t_eager = ev_time(lambda: step(*pair), reps=20)
            g = GraphedStep(step, pair)
            t_graph = ev_time(lambda: g(*pair), reps=50)
            print(json.dumps({"small_batch": {"hidden": hidden, "B": 64, "seq_small": small, "eager_us": t_eager, "graphed_us": t_graph}}),
                  flush=True)
            del g, model, flat, opt
    ops.USE_SEQ_SMALL = True


def chickenpox(dev):
    import bench_configs as BC
    for K in (1, 2, 3):
        r = BC.chickenpox_epoch(dev, 8, K=K)
        print(json.dumps({"chickenpox": {"K": K, "eager_ms": r["gpu_eager_ms_per_epoch"], "graphed_ms": r["gpu_graphed_ms_per_epoch"],
                                          "cpu_ms": r["cpu_oracle_ms_per_epoch"]}}), flush=True)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["tgcn_cell", "small_batch", "chickenpox"]
    if "tgcn_cell" in which:
        tgcn_cell(dev)
    if "small_batch" in which:
        small_batch(dev)
    if "chickenpox" in which:
        chickenpox(dev)
