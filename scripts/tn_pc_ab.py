#!/usr/bin/env python
"""A/B of the split-bf16 weight-gradient kernels — all wavefronts alike (gemm_bx_tn_kernel) against producers / consumers
(gemm_bx_tn_pc_kernel, pgt_tune("gemm_bx_tn_pc")) — at the training step's shapes: M = 12 x 211 968 rows, K = 5 x 66, N = 128 / 64;
alternating, HIP events around 10 calls of pgt_gemm_tn_acc_f32 (two kernel launches each); then the two forms' sums compared."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
M, segs, segk = 12 * 211968, 5, 66
A = torch.randn(segs, M, segk, device=dev)
for N in (128, 64):
    G = torch.randn(M, N, device=dev)
    dW, db = torch.zeros(segs * segk, N, device=dev), torch.zeros(N, device=dev)
    nbytes = 4.0 * M * (segs * segk + N)
    out = {2: [], 1: [], 0: []}
    for rep in range(3):
        for pc in (2, 1, 0):
            lib.tune("gemm_bx_tn_pc", pc)
            for _ in range(2):
                ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dW, N, db, M, N)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dW, N, db, M, N)
            e1.record()
            torch.cuda.synchronize()
            out[pc].append(round(1e3 * e0.elapsed_time(e1) / 10, 1))
    res = {}
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = True
    for pc in (2, 1, 0):
        lib.tune("gemm_bx_tn_pc", pc)
        dW, db = torch.zeros(segs * segk, N, device=dev), torch.zeros(N, device=dev)
        ops.gemm_tn_acc(A, segk, M * segk, segs, segk, G, N, dW, N, db, M, N)
        torch.cuda.synchronize()
        res[pc] = (dW, db)
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = False
    lib.tune("gemm_bx_tn_pc", 1)
    print(json.dumps({"N": N, "twelve_wavefronts_us": out[2], "four_plus_four_us": out[1], "all_alike_us": out[0],
                      "twelve_frac": round(nbytes / min(out[2]) / 1e3 / 8000, 3), "four_plus_four_frac": round(nbytes / min(out[1]) / 1e3 / 8000, 3),
                      "all_alike_frac": round(nbytes / min(out[0]) / 1e3 / 8000, 3),
                      "bit_identical": bool(all(torch.equal(res[k][0], res[0][0]) and torch.equal(res[k][1], res[0][1]) for k in (1, 2)))}), flush=True)
    del G
