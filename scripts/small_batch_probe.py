"""The reference's batch size (B = 64, hidden 64) as one hipGraph per step — what bench_configs.small_batch times; run
under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from pytorch_geometric_temporal_amd import dp
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.graphed import GraphedStep

dev = torch.device("cuda:0")
for kv in [a for a in sys.argv[1:] if "=" in a]:          # pgt_tune switches: key=value (after the positional arguments)
    from pytorch_geometric_temporal_amd import _lib
    _lib.get_lib().tune(kv.split("=")[0], int(kv.split("=")[1]))
sys.argv = [a for a in sys.argv if "=" not in a]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ei_np, ew_np = syn.sensor_graph(207, 1515, seed=0, symmetric=False)
ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
series = torch.randn(4000, 207, 2, device=dev)
torch.manual_seed(0)
model = bench.Model(hidden).to(dev)
flat = dp.FlatParameters(model.parameters())
opt = flat.adam(lr=1e-3)
ar = torch.arange(12, device=dev)


def step(xi, yi):
    X, y = series[xi], series[yi]
    pred = model(X, ei, ew)
    loss = bench.masked_mae_loss(pred, y)
    flat.zero()
    loss.backward()
    opt.step()
    return loss


i = torch.from_numpy(np.random.default_rng(7).integers(0, 3900, size=B)).to(dev)
pair = (i[:, None] + ar[None, :], i[:, None] + 12 + ar[None, :])
for _ in range(3):
    step(*pair)
if len(sys.argv) > 4 and sys.argv[4] == "eager":      # for rocprofv3 --kernel-trace: the same kernels, launched one by one
    for _ in range(reps):
        step(*pair)
    torch.cuda.synchronize()
    sys.exit(0)
g = GraphedStep(step, pair)
for _ in range(5):
    g(*pair)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g(*pair)
torch.cuda.synchronize()
print(f"B = {B}, hidden {hidden}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per graphed step")
