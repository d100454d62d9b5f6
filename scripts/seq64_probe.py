"""The one-launch hidden-64 sequence kernels (csrc/seq64.hip) against the per-step launches: outputs, gradients, and the time of
forward and forward + backward at several batch sizes.  usage: seq64_probe.py [B ...]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd import ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN

dev = torch.device("cuda:0")
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 128, 256, 1024]
edges = 1722 if "e1722" in sys.argv else 1515
ei_np, ew_np = syn.sensor_graph(207, edges, seed=0, symmetric=False)
ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
torch.manual_seed(0)
model = BatchedDCRNN(2, 64, 3).to(dev)
with torch.no_grad():
    for p in model.parameters():
        p.mul_(0.5)


def run(flag, X, w, reps, backward):
    ops.USE_SEQ64 = flag
    for _ in range(2):
        model.zero_grad()
        out = model(X, ei, ew)
        if backward:
            (out * w).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.zero_grad()
        out = model(X, ei, ew)
        if backward:
            (out * w).sum().backward()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out.detach(), [p.grad.clone() for p in model.parameters()] if backward else None


for B in batches:
    X = torch.randn(B, 12, 207, 2, device=dev)
    w = torch.randn(B, 12, 207, 64, device=dev)
    reps = 20 if B <= 256 else 5
    t_old, o_old, g_old = run(False, X, w, reps, True)
    t_new, o_new, g_new = run(True, X, w, reps, True)
    f_old = run(False, X, w, reps, False)[0]
    f_new = run(True, X, w, reps, False)[0]
    gerr = max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(g_old, g_new))
    print(f"B = {B} ({edges} edges): forward {f_old:.3f} -> {f_new:.3f} ms, forward + backward {t_old:.3f} -> {t_new:.3f} ms; "
          f"max |out diff| {float((o_old - o_new).abs().max()):.2e} (scale {float(o_old.abs().max()):.2f}), worst relative gradient diff {gerr:.2e}",
          flush=True)
