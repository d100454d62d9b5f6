#!/usr/bin/env python
"""Lab probe: aggregation on a skewed graph (a few very long rows) at the north-star size."""
import os, sys, numpy as np, torch
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
n = 200_000
ei, ew = syn.local_graph(n, 8, seed=0)
rng = np.random.default_rng(0)
for hubs, hub_deg in ((0, 0), (20, 2000), (20, 20000), (200, 2000)):
    if hubs:
        hub_rows = rng.choice(n, hubs, replace=False)
        src = np.concatenate([rng.choice(n, hub_deg, replace=False) for _ in hub_rows])
        dst = np.repeat(hub_rows, hub_deg)
        e2 = np.concatenate([ei, np.stack([src, dst])], axis=1)
        w2 = np.concatenate([ew, np.ones(src.size, dtype=np.float32)])
        key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
        e2, w2 = e2[:, key], w2[key]
    else:
        e2, w2 = ei, ew
    g = ops.DConvGraph(torch.from_numpy(e2).to(dev), torch.from_numpy(w2).to(dev), n)
    X, Y = torch.randn(n, 64, device=dev), torch.empty(n, 64, device=dev)
    nb = ops.spmm_algorithmic_bytes(n, g.E, 64, False)
    t = timeit(lambda: ops.spmm(g.fwd_o, X, Y))
    ref = torch.zeros(n, 64, dtype=torch.float64, device=dev)
    csr = g.fwd_o
    nnz = int(csr.rowptr[-1])
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
    ref.index_add_(0, rows, X.double()[csr.col[:nnz].long()] * csr.val[:nnz].double()[:, None])
    err = float((Y.double() - ref).abs().max())
    print(f"hubs={hubs} x {hub_deg}: E={g.E} longest row {csr.max_len} ellw={'yes' if csr.ellw is not None else 'no'}: {t:.1f} us "
          f"({nb / t / 1e3 / 8000:.3f} of 8 TB/s), max err {err:.2e}")

# locality-ordered graph with a fraction of long-range edges (the ELLW layout gives them LDS rows behind the window; what does not fit comes through the CSR)
for frac in (0.0, 0.01, 0.03, 0.045):
    e2, w2 = ei.copy(), ew.copy()
    k = int(frac * e2.shape[1])
    if k:
        pick = rng.choice(e2.shape[1], k, replace=False)
        e2[0, pick] = rng.integers(0, n, k)
        key = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
        e2, w2 = e2[:, key], w2[key]
    g = ops.DConvGraph(torch.from_numpy(e2).to(dev), torch.from_numpy(w2).to(dev), n)
    X, Y = torch.randn(n, 64, device=dev), torch.empty(n, 64, device=dev)
    nb = ops.spmm_algorithmic_bytes(n, g.E, 64, False)
    t_e = timeit(lambda: ops.spmm(g.fwd_o, X, Y))
    e = g.fwd_o.ellw
    far, far_csr = (e.far, e.far_csr) if e is not None else (-1, -1)
    t_c = timeit(lambda: ops.spmm(g.fwd_o, X, Y, ellw=False))
    t_i = timeit(lambda: ops.spmm(g.fwd_i, X, Y))
    print(f"long-range fraction {frac}: halo {g.fwd_o.halo}, out-of-window slots {far} ({far_csr} via CSR): P_o auto {t_e:.1f} us, "
          f"CSR tiles {t_c:.1f} us; P_i auto {t_i:.1f} us")
