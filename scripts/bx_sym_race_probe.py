#!/usr/bin/env python
"""GPU diagnostic for the symmetric short-K split-bf16 kernel (64 -> 320 columns): where do repeated launches differ?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    O = 64
    for M in (211968, 40000, 211968 + 13):
        for Kd, N in ((64, 320), (64, 288), (32, 320), (128, 320)):
            g = torch.Generator(device="cpu").manual_seed(M + Kd)
            dP = torch.randn(M, Kd, generator=g).to(dev)
            WH = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
            ref = None
            bad_total = {}
            for rep in range(12):
                if N % O == 0:
                    G = torch.full((N // O, M, O), float("nan"), device=dev)
                    ops.gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, O, M * O, O, None, M, N)
                    out = G.permute(1, 0, 2).reshape(M, N)
                else:
                    G = torch.full((N // 32, M, 32), float("nan"), device=dev)
                    ops.gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, 32, M * 32, 32, None, M, N)
                    out = G.permute(1, 0, 2).reshape(M, N)
                if ref is None:
                    lib.tune("gemm_bx", 0)
                    G2 = torch.empty_like(G)
                    w = G.shape[2]
                    ops.gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G2, w, M * w, w, None, M, N)
                    lib.tune("gemm_bx", 1)
                    ref = G2.permute(1, 0, 2).reshape(M, N)
                d = (out - ref).abs()
                d = torch.where(torch.isnan(d), torch.full_like(d, 1e9), d)
                bad = (d > 1e-4).nonzero()
                for r, c in bad.tolist()[:4000]:
                    bad_total.setdefault(rep, []).append((r, c))
            if not bad_total:
                print(f"M={M} K={Kd} N={N}: all 12 repeats agree with the exact-fp32 kernels", flush=True)
                continue
            for rep, lst in bad_total.items():
                rows = sorted({r for r, _ in lst})
                cols = sorted({c for _, c in lst})
                print(f"M={M} K={Kd} N={N} repeat {rep}: {len(lst)} bad elements; rows {rows[:12]}{'...' if len(rows) > 12 else ''} "
                      f"(blocks {sorted({r // 32 for r in rows})[:8]}, block % 256 = {sorted({(r // 32) % 256 for r in rows})[:8]}); "
                      f"cols {cols[0]}..{cols[-1]} ({len(cols)} distinct)", flush=True)


if __name__ == "__main__":
    main()
