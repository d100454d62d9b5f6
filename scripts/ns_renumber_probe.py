#!/usr/bin/env python
"""GPU probe of the renumbered ELLW layout (csrc/tile_order.hip + spmm_ellw64_kernel<.., EllwCfgC, true>) at the north-star
shape: the 447 x 447 mesh numbered row by row and at random, a shuffled band graph (in-degree 8 and 16) — the renumbered
window kernel against the CSR row tiles on the caller's numbering, and against the same mesh numbered along a Hilbert curve by
the caller.  Plain eager launches over six rotating (X, Y) pairs, so that rocprofv3 --kernel-trace / --pmc see single kernels.
    python scripts/ns_renumber_probe.py [launches]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    pairs = 6
    out = {}
    for name in ("grid2d_hilbert", "grid2d_rowmajor", "grid2d_shuffled", "local_shuffled", "local_deg16_shuffled"):
        if name.startswith("grid2d"):
            n = 447 * 447
            ei, ew = syn.grid2d_graph(447, name.split("_")[1], 0)
        else:
            n = 200_000
            ei, ew = syn.local_graph(n, 16 if "deg16" in name else 8, seed=0)
            ei = np.random.default_rng(5).permutation(n)[ei]
        g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
        Xs = [torch.randn(n, 64, device=dev) for _ in range(pairs)]
        Ys = [torch.empty(n, 64, device=dev) for _ in range(pairs)]
        t0 = time.perf_counter()
        ops.spmm(g.fwd_o, Xs[0], Ys[0])
        torch.cuda.synchronize()
        prep = time.perf_counter() - t0
        e = g.fwd_o.ellw or None
        rec = {"layout": None if e is None else ("renumbered" if e.order is not None else "ellw"), "prep_s": prep}
        if e is not None:
            rec.update(tile_rows=e.tile_rows, width=e.width, n_tiles=e.n_tiles, far=e.far, far_csr=e.far_csr,
                       mode="scale" if e.scale is not None else "vals")
        nbytes = ops.spmm_algorithmic_bytes(n, g.E, 64, False)
        for label, flag in (("shipped", None), ("csr_row_tiles", False)):
            for i in range(pairs):
                ops.spmm(g.fwd_o, Xs[i], Ys[i], ellw=flag)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side = torch.cuda.Stream(device=dev)
            graph = torch.cuda.CUDAGraph()
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(launches):
                        ops.spmm(g.fwd_o, Xs[i % pairs], Ys[i % pairs], ellw=flag)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / launches
            rec[label] = {"us": us, "frac": nbytes / us / 1e3 / 8000.0}
            del graph
        # the per-slot operator P_i through its layout too
        ops.spmm(g.fwd_i, Xs[0], Ys[0])
        ei_ = g.fwd_i.ellw or None
        rec["P_i"] = None if ei_ is None else {"renumbered": ei_.order is not None, "mode": "scale" if ei_.scale is not None else "vals",
                                               "far_csr": ei_.far_csr, "far": ei_.far}
        out[name] = rec
        print(json.dumps({name: rec}), flush=True)
        del g, Xs, Ys


if __name__ == "__main__":
    main()
