#!/usr/bin/env python
"""A handful of aggregation launches and nothing else — the target of the rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS per dispatch).  usage: spmm_only.py {ns-local|ns-uniform|metrla} [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "ns-local"
if which.startswith("ns"):
    n, F = 200_000, 64
    ei, ew = (syn.local_graph if which == "ns-local" else syn.uniform_graph)(n, 8, seed=0)
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n, F = 207, B * 66
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
pairs = 6
Xs = [torch.randn(n, F, device=dev) for _ in range(pairs)]
Ys = [torch.empty(n, F, device=dev) for _ in range(pairs)]
for i in range(18):
    ops.spmm(g.fwd_o, Xs[i % pairs], Ys[i % pairs])
torch.cuda.synchronize()
print(which, "algorithmic bytes per launch:", ops.spmm_algorithmic_bytes(n, g.E, F, False))
