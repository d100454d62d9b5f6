#!/usr/bin/env python
"""Lab probe: ELLW column chunks vs the wide CSR kernel on locality-ordered graphs with rows wider than 64 floats."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for n in (50_000, 200_000):
    ei, ew = syn.local_graph(n, 8, seed=0)
    g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
    for F in (128, 256, 1024):
        if n * F * 4 * 2 > 6e9: continue
        X, Y = torch.randn(n, F, device=dev), torch.empty(n, F, device=dev)
        nb = ops.spmm_algorithmic_bytes(n, g.E, F, False)
        for csr, name in ((g.fwd_o, "P_o (scale mode)"), (g.fwd_i, "P_i")):
            a = timeit(lambda: ops.spmm(csr, X, Y, ellw=False))
            b = timeit(lambda: ops.spmm(csr, X, Y, ellw=True))
            e = csr.ellw
            print(f"N={n} F={F} {name}: CSR {a:.1f} us ({nb / a / 1e3 / 8000:.3f} of 8 TB/s)   ELLW chunks {b:.1f} us ({nb / b / 1e3 / 8000:.3f})"
                  f"   [{'scale' if e.scale is not None else 'vals'}, {e.n_tiles} tiles x {F // 64} chunks]")
