"""ST-Conv block on the GPU: the gated temporal convolution (csrc/tconv.hip) against the three-Conv2d + permutes chain it
replaces (torch / MIOpen), forward and forward + backward, at STGCN-typical shapes; algorithmic bytes / time for the kernel.
One JSON line per shape."""
import json
import sys
import time

import torch
import torch.nn.functional as F

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd.nn.attention import STConv, TemporalConv  # noqa: E402


def t_gpu(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def chain(m, X):
    Xp = X.permute(0, 3, 2, 1)
    P = m.conv_1(Xp)
    Q = torch.sigmoid(m.conv_2(Xp))
    return F.relu(P * Q + m.conv_3(Xp)).permute(0, 3, 2, 1)


def main():
    dev = torch.device("cuda:0")
    for (B, T, N, Cin, Cout, k) in ((50, 12, 228, 1, 64, 3), (50, 10, 228, 16, 64, 3), (50, 12, 228, 64, 64, 3), (10, 5, 300, 100, 8, 3),
                                    (64, 12, 207, 2, 32, 3), (32, 12, 883, 64, 64, 3)):
        torch.manual_seed(0)
        m = TemporalConv(Cin, Cout, k).to(dev)
        X = torch.randn(B, T, N, Cin, device=dev, requires_grad=True)
        Tp = T - k + 1
        with torch.no_grad():
            ours_f = t_gpu(lambda: m(X))
            ref_f = t_gpu(lambda: chain(m, X))
            err = float((m(X) - chain(m, X)).abs().max())

        def fb(f):
            m.zero_grad(set_to_none=True)
            X.grad = None
            f(m, X).sum().backward() if f is chain else f(X).sum().backward()
        ours_fb = t_gpu(lambda: fb(m))
        ref_fb = t_gpu(lambda: fb(chain))
        bytes_f = 4.0 * (B * T * N * Cin + B * Tp * N * Cout)
        print(json.dumps({"shape": [B, T, N, Cin, Cout, k], "fwd_us": ours_f, "torch_chain_fwd_us": ref_f, "fwd_bwd_us": ours_fb,
                          "torch_chain_fwd_bwd_us": ref_fb, "fwd_algorithmic_GBs": bytes_f / ours_f / 1e3,
                          "fwd_fp32_TFLOPs": 2.0 * B * Tp * N * k * Cin * 3 * Cout / ours_f / 1e6, "max_abs_diff_vs_chain": err}), flush=True)
    # the whole block at the reference's test shape and at a PeMSD7-like training shape
    from pytorch_geometric_temporal_amd.dataset import synthetic as syn
    for (B, T, N, Cin, hid, Cout, K) in ((10, 5, 300, 100, 8, 10, 2), (50, 12, 228, 1, 16, 64, 3)):
        ei_np, ew_np = syn.sensor_graph(N, 8 * N, seed=0, symmetric=True)
        ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
        m = STConv(N, Cin, hid, Cout, 3, K).to(dev)
        X = torch.randn(B, T, N, Cin, device=dev)

        def step():
            m.zero_grad(set_to_none=True)
            m(X, ei, ew).square().mean().backward()
        us = t_gpu(step)
        print(json.dumps({"stconv": [B, T, N, Cin, hid, Cout, K], "fwd_bwd_us": us}), flush=True)


if __name__ == "__main__":
    main()
