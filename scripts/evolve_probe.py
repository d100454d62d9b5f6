#!/usr/bin/env python
"""GPU probe: the one-launch EvolveGCN-H weight evolution against the module chain it replaces (TopKPooling + torch.nn.GRU),
forward + backward of one snapshot, eager and as a hipGraph of 50 snapshots."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import ops  # noqa: E402
from pytorch_geometric_temporal_amd.nn.conv import TopKPooling  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, F_ = 129, 8
    torch.manual_seed(0)
    gru = torch.nn.GRU(F_, F_, 1).to(dev)
    pool = TopKPooling(F_, F_ / n).to(dev)
    X = torch.randn(n, F_, device=dev, requires_grad=True)
    W0 = torch.randn(1, F_, F_, device=dev, requires_grad=True)

    def fused():
        w = ops.EvolveWeightFunction.apply(X, pool.select.weight, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0, W0, F_)
        w.sum().backward()

    def modules():
        xt = pool(X)[0][None]
        _, w = gru(xt, W0)
        w.sum().backward()

    for name, fn in (("fused", fused), ("modules", modules)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200 * 1e6
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(50):
                    fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:8s} eager {eager:8.1f} us per snapshot (fwd + bwd);  graphed {1e3 * e0.elapsed_time(e1) / 200:7.1f} us")


if __name__ == "__main__":
    main()
