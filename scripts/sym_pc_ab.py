#!/usr/bin/env python
"""A/B of the short-K split-bf16 product — all wavefronts alike (gemm_bx_sym_kernel) against ten consumers + two producers
(gemm_bx_sym_pc_kernel, pgt_tune("gemm_bx_sym_pc")) — at the training step's feature-gradient shapes: M = 211 968 rows,
K = 128 / 64 -> 320 columns in five 64-wide output segments; alternating, 20 launches as one hipGraph each; then the bits compared."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402
from slab_probe import timed  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
M, N = 211968, 320
for K in (128, 64):
    g = torch.Generator().manual_seed(K)
    A = torch.randn(1, M, K, generator=g).to(dev)
    Bt = torch.randn(N, K, generator=g).to(dev)
    Cs = torch.empty(N // 64, M, 64, device=dev)
    nbytes = 4.0 * M * (K + N)
    run = lambda: ops.gemm(A, K, M * K, 1, K, Bt, 1, K, Cs, 64, M * 64, 64, None, M, N)   # noqa: E731
    out = {1: [], 0: []}
    for rep in range(3):
        for pc in (1, 0):
            lib.tune("gemm_bx_sym_pc", pc)
            out[pc].append(round(timed(run), 2))
    res = {}
    for pc in (1, 0):
        lib.tune("gemm_bx_sym_pc", pc)
        Cs.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        res[pc] = Cs.clone()
    lib.tune("gemm_bx_sym_pc", 0)
    ref = (A[0, :4096].double() @ Bt.double().t())
    got = res[1].permute(1, 0, 2).reshape(M, N)[:4096].double()
    print(json.dumps({"K": K, "producers_consumers_us": out[1], "all_alike_us": out[0], "pc_frac": round(nbytes / min(out[1]) / 1e3 / 8000, 3),
                      "all_alike_frac": round(nbytes / min(out[0]) / 1e3 / 8000, 3), "bit_identical": bool(torch.equal(res[1], res[0])),
                      "max_err_vs_fp64_first_4096_rows": float((got - ref).abs().max())}), flush=True)
