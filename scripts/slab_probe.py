#!/usr/bin/env python
"""GPU probe of the LDS-resident diffusion-stack kernels (csrc/dconv_slab.hip) at the benchmark shapes: whole-sample
kernels against the column-split ones for every (windows, workgroups per CU, threads) the library has shapes for.
Each figure: 20 launches captured as one hipGraph, mean of 3 replays; bytes = algorithmic (forward: read 1 + write 4
blocks, backward: read 5 + write 1).   python scripts/slab_probe.py [B ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402


def timed(fn, launches=20):
    dev = torch.device("cuda:0")
    for _ in range(3):
        fn()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(launches):
                fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / launches)
    return sum(ts) / len(ts)


def main():
    lib = _lib.get_lib()
    dev = torch.device("cuda:0")
    whole_only = "whole" in sys.argv[1:]
    Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 64]
    n, E, K = 207, 1515, 3
    ei, ew = syn.sensor_graph(n, E, seed=0, symmetric=False)
    g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
    configs = [("whole_pair", 0, 0, 0), ("whole", 0, 0, 0), ("whole_gu4", 0, 0, 0), ("whole_again", 0, 0, 0), ("auto", 1, 0, 0)]
    for ns in (() if whole_only else (2, 3, 4, 6, 8)):
        for wpc in (1, 2, 3):
            for th in (0, 512, 640, 1024):
                configs.append((f"split{ns}_wpc{wpc}_t{th}", ns, wpc, th))
    for B in Bs:
        for name, C, bwd in (("fwd_C66", 66, False), ("bwd_C64", 64, True), ("bwd_C66", 66, True)):
            TS = torch.randn(5, B * n, C, device=dev)
            seg = B * n * C
            nbytes = (6 if bwd else 5) * 4 * B * n * C
            seen = {}
            for cname, split, wpc, th in configs:
                lib.tune("slab_split", split)
                lib.tune("slab_wpc", wpc)
                lib.tune("slab_threads", th)
                lib.tune("slab_quad", 0 if cname == "whole_pair" else 1)
                lib.tune("slab_gu", 4 if cname == "whole_gu4" else 2)
                try:
                    if bwd:
                        us = timed(lambda: ops._slab_bwd(g, TS[0], seg, B, C, K, True))
                    else:
                        us = timed(lambda: ops._slab_fwd(g, TS[0], seg, B, C, K))
                except Exception as e:   # noqa: BLE001
                    print(json.dumps({"B": B, "case": name, "config": cname, "error": repr(e)[:200]}))
                    continue
                finally:
                    lib.tune("slab_split", 1)
                    lib.tune("slab_wpc", 0)
                    lib.tune("slab_threads", 0)
                    lib.tune("slab_quad", 1)
                    lib.tune("slab_gu", 2)
                lib.tune("slab_split", split); lib.tune("slab_wpc", wpc); lib.tune("slab_threads", th)
                plan = ops.slab_plan(g, C, K, B) + (cname,) if cname.startswith("whole") else ops.slab_plan(g, C, K, B)
                lib.tune("slab_split", 1); lib.tune("slab_wpc", 0); lib.tune("slab_threads", 0)
                if plan in seen:
                    continue
                seen[plan] = us
                print(json.dumps({"B": B, "case": name, "config": cname, "plan": plan, "us": round(us, 2),
                                  "GBs": round(nbytes / us / 1e3, 1), "hbm_frac": round(nbytes / us / 1e3 / 8000, 3)}), flush=True)
            del TS


if __name__ == "__main__":
    main()
