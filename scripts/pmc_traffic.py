#!/usr/bin/env python
"""Condense the rocprofv3 passes of scripts/pmc_bench.sh into profiles/r03_pmc_traffic.json (what bench.py reports as
`roofline.traffic`) and profiles/r03_bench_kernel_stats.csv.

Per kernel class (gemm / gemm_tn / stack / ns_spmm ...): mean over the dispatches of FETCH_SIZE x 2 (the gfx950
correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE reports half the bytes of wide coalesced reads) and of
WRITE_SIZE, both reported by rocprofv3 in KB; bytes_per_launch = their sum."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


TAG = sys.argv[2] if len(sys.argv) > 2 else "r04"


def klass(name):
    if "tgcn_cell_fwd" in name:
        return "tgcn_cell_fwd"
    if "tgcn_cell_bwd" in name:
        return "tgcn_cell_bwd"
    if "relu_linear" in name:
        return "readout"
    if "tconv_glu" in name:
        return "tconv"
    if "seq64" in name:
        return "seq64"
    if "seq_small" in name:
        return "seq_small"
    if "gemm_tn" in name or "gemm_bx_tn" in name:
        return "gemm_tn"
    if "gemm" in name:
        return "gemm"
    if "dconv_slab" in name:
        return "stack"
    if "spmm_ellw64" in name:
        return "ns_spmm_ellw"
    if "spmm_tile" in name:
        return "ns_spmm_csr_tiles"
    if "spmm" in name:
        return "spmm"
    if "gru_" in name or "lstm" in name:
        return "gates"
    return None


def reduce(path, wanted):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    per_kernel = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if path is None:
        return acc, per_kernel
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] not in wanted:
                continue
            k = klass(r["Kernel_Name"])
            if k is None:
                continue
            for store, key in ((acc, k), (per_kernel, r["Kernel_Name"].split("(")[0][-70:])):
                a = store[key][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc, per_kernel


fetch, fetch_k = reduce(newest(os.path.join(src, TAG + "_pmc_fetch", "**", "*counter_collection.csv")), {"FETCH_SIZE"})
write, write_k = reduce(newest(os.path.join(src, TAG + "_pmc_write", "**", "*counter_collection.csv")), {"WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"})
out = {"command": sys.argv[1] if len(sys.argv) > 1 else "", "units": "bytes per launch; FETCH_SIZE (KB) x 2 x 1000, WRITE_SIZE (KB) x 1000",
       "kernels": {}, "per_kernel": {}}
for store_f, store_w, dest in ((fetch, write, out["kernels"]), (fetch_k, write_k, out["per_kernel"])):
    for k in sorted(set(store_f) | set(store_w)):
        f = store_f.get(k, {}).get("FETCH_SIZE", [0.0, 0])
        w = store_w.get(k, {}).get("WRITE_SIZE", [0.0, 0])
        h = store_w.get(k, {}).get("TCC_HIT_sum", [0.0, 0])
        m = store_w.get(k, {}).get("TCC_MISS_sum", [0.0, 0])
        fb = 2e3 * f[0] / f[1] if f[1] else None
        wb = 1e3 * w[0] / w[1] if w[1] else None
        dest[k] = {"dispatches": f[1] or w[1], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                   "bytes_per_launch": (fb or 0) + (wb or 0) if (fb is not None and wb is not None) else None,
                   "l2_hit_rate": (h[0] / (h[0] + m[0])) if (h[0] + m[0]) > 0 else None}
os.makedirs(dst, exist_ok=True)
for d in (dst, src):      # profiles/ when run in the build container; gpurun_out/prof travels back from the GPU box
    with open(os.path.join(d, TAG + "_pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
print(json.dumps(out["kernels"], indent=1))
st = newest(os.path.join(src, TAG + "_stats", "**", "*kernel_stats.csv"))
if st:
    shutil.copy(st, os.path.join(dst, TAG + "_bench_kernel_stats.csv"))
    shutil.copy(st, os.path.join(src, TAG + "_bench_kernel_stats.csv"))
    print("copied", st)
