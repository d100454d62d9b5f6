#!/usr/bin/env python
"""Condense gpurun_out/prof (rocprofv3 output) into small tracked files under profiles/.

    python scripts/summarise_profiles.py r01a        # tag = round / session label

kernel_stats.csv is copied as is (it is the `--kernel-trace --stats` summary); the PMC passes are reduced to one
row per (kernel, counter): mean value over the dispatches plus the mean dispatch duration of that pass.
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; the gfx950 correction of MI355X_MICROARCH.md §HBM
(FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read) is applied in the `MB_corrected` column.
"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)



def newest(pattern):
    """rocprofv3 names its files <pid>_*.csv and gpurun MERGES gpurun_out/ across calls, so a pass directory can hold
    the files of several runs: only the most recent one belongs to this round's numbers."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


for d in sorted(glob.glob(os.path.join(src, "stats*"))):
    if not os.path.isdir(d):
        continue
    for f in newest(os.path.join(d, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(d)}_kernel_stats.csv"))
        print("copied", f)

rows = []


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:80]


for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in newest(os.path.join(d, "*", "*counter_collection.csv")):
        acc = defaultdict(lambda: [0.0, 0.0, 0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                a = acc[k]
                a[0] += float(r["Counter_Value"])
                a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a[2] += 1
        for (kern, ctr), (v, ns, n) in sorted(acc.items()):
            if "spmm" not in kern and "gemm" not in kern and "cell" not in kern:
                continue
            mean = v / n
            corr = ""
            if ctr == "FETCH_SIZE":
                corr = f"{2 * mean / 1e3:.2f}"
            elif ctr == "WRITE_SIZE":
                corr = f"{mean / 1e3:.2f}"
            rows.append((os.path.basename(d), kern, ctr, n, f"{mean:.1f}", corr, f"{ns / n / 1e3:.2f}"))
out = os.path.join(dst, f"{tag}_pmc_summary.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["pass", "kernel", "counter", "dispatches", "mean_value", "MB_corrected", "mean_dispatch_us_profiled"])
    w.writerows(rows)
print("wrote", out, len(rows), "rows")
for f in ("bench.json", "probe.jsonl"):
    p = os.path.join(ROOT, "gpurun_out", f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
