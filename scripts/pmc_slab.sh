#!/bin/bash
# PMC pass over the LDS-resident diffusion stack kernels (SQ LDS counters; no trace domains besides --kernel-trace)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_slab -- python $OLDPWD/scripts/gpu_probe.py stack) > $O/pmc_slab.log 2>&1
echo "rc=$?"
tail -2 $O/pmc_slab.log
python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc_slab/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "slab" in k:
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
