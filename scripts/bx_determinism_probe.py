#!/usr/bin/env python
"""GPU diagnostic: every split-bf16 kernel of csrc/gemm_bx.hip at the benchmark size, the same launch repeated on the same
operands — kernels without atomics must reproduce their output bit for bit (a difference = a race / a missing wait)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402


def report(name, outs, ref=None):
    base = outs[0]
    worst, rows = 0.0, 0
    for o in outs[1:]:
        d = (o - base).abs()
        worst = max(worst, float(d.max()))
        rows = max(rows, int((d.reshape(d.shape[0], -1).max(dim=1).values > 0).sum()))
    msg = f"{name:46s} repeats {len(outs)}  max run-to-run diff {worst:.3e}  differing rows {rows}"
    if ref is not None:
        msg += f"  max err vs fp64 {float((base.double() - ref).abs().max()):.3e}"
    print(msg, flush=True)


def main():
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 211968
    reps = 6
    g = torch.Generator(device="cpu").manual_seed(0)
    S, C, O = 5, 66, 64
    K = S * C
    A = torch.randn(S, M, C, generator=g).to(dev)
    Wzr, bzr = (torch.randn(K, 2 * O, generator=g) / K ** 0.5).to(dev), torch.randn(2 * O, generator=g).to(dev)
    Wh, bh = (torch.randn(K, O, generator=g) / K ** 0.5).to(dev), torch.randn(O, generator=g).to(dev)
    H = torch.randn(M, O, generator=g).to(dev)
    # forward products
    outs = []
    for _ in range(reps):
        Cc = torch.empty(M, 2 * O, device=dev)
        ops.gemm(A, C, M * C, S, C, Wzr, 2 * O, 1, Cc, 2 * O, 0, 2 * O, bzr, M, 2 * O)
        outs.append(Cc)
    report("NN 330->128 plain", outs)
    outs, outs2 = [], []
    for _ in range(reps):
        zr, xhr = torch.empty(M, 2 * O, device=dev), torch.zeros(M, C, device=dev)
        ops.gemm_gru_zr(A, C, M * C, S, C, Wzr, 2 * O, 1, bzr, zr, H, xhr, 2)
        outs.append(zr); outs2.append(xhr)
    report("NN 330->128 + z|r gates: zr", outs)
    report("NN 330->128 + z|r gates: H*r", outs2)
    zr = outs[0]
    outs, outs2 = [], []
    for _ in range(reps):
        ht, o0, o1 = torch.empty(M, O, device=dev), torch.empty(M, O, device=dev), torch.zeros(M, C, device=dev)
        ops.gemm_gru_h(A, C, M * C, S, C, Wh, O, 1, bh, ht, zr, H, o0, o1[:, 2:])
        outs.append(ht); outs2.append(o0)
    report("NN 330->64 + candidate gate: ht", outs)
    report("NN 330->64 + candidate gate: state", outs2)
    # feature gradients (symmetric short-K kernel)
    for Kd in (128, 64):
        dP = torch.randn(M, Kd, generator=g).to(dev)
        for N in (320, 256):
            WH = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
            outs = []
            for _ in range(reps):
                G = torch.empty(N // O, M, O, device=dev)
                ops.gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, O, M * O, O, None, M, N)
                outs.append(G.permute(1, 0, 2).reshape(M, N))
            idx = torch.arange(0, M, max(M // 512, 1), device=dev)
            ref = dP[idx].double() @ WH.double().t()
            report(f"NT {Kd}->{N} feature gradient", outs)
            print(f"      sampled rows vs fp64: {float((outs[0][idx].double() - ref).abs().max()):.3e}; "
                  f"rows of the repeats vs fp64: {[float((o[idx].double() - ref).abs().max()) for o in outs[1:]]}")
    # weight gradient, deterministic mode (no atomics) and atomics
    T = 12 if M <= 300000 else 1
    At = torch.randn(S, T * M, C, generator=g).to(dev) if T > 1 else A
    for N in (128, 64):
        Gm = torch.randn(T * M, N, generator=g).to(dev)
        for det in (True, False):
            ops.DETERMINISTIC_WEIGHT_GRADIENTS = det
            outs = []
            for _ in range(reps):
                dW, db = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
                ops.gemm_tn_acc(At, C, T * M * C, S, C, Gm, N, dW, N, db, T * M, N)
                outs.append(torch.cat([dW, db[None]], 0))
            ops.DETERMINISTIC_WEIGHT_GRADIENTS = False
            report(f"TN 330x{N} weight gradient M={T * M} {'det' if det else 'atomics'}", outs)
        lib.tune("gemm_bx", 0)
        dW, db = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
        ops.gemm_tn_acc(At, C, T * M * C, S, C, Gm, N, dW, N, db, T * M, N)
        lib.tune("gemm_bx", 1)
        print(f"      split-bf16 vs exact-fp32 kernels: max diff {float((outs[0][:K] - dW).abs().max()):.3e} (scale {float(dW.abs().max()):.1f})")


if __name__ == "__main__":
    main()
