#!/bin/bash
# LDS counters of the diffusion-stack kernels at the benchmark shape (B = 1024, whole-sample quad kernels): one rocprofv3 --pmc
# pass over scripts/slab_probe.py (counters in their own run, --kernel-trace only beside them); summary -> gpurun_out/prof/r03_pmc_slab.csv
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/r03_pmc_slab -- python $OLDPWD/scripts/slab_probe.py 1024 whole) > $O/r03_pmc_slab.log 2>&1
echo "rc=$?"
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
files = glob.glob(os.path.join(O, "r03_pmc_slab", "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "dconv_slab" not in k:
            continue
        import re as _re
        m_ = _re.search(r"(dconv_slab_\w+?_kernel<[^>]*>)", k)
        name = m_.group(1) if m_ else k[:60]
        key = (name, row["Counter_Name"])
        acc[key][0] += 1
        acc[key][1] += float(row["Counter_Value"])
with open(os.path.join(O, "r03_pmc_slab.csv"), "w") as out:
    out.write("kernel,counter,dispatches,mean_value\n")
    for (name, c), (n, v) in sorted(acc.items()):
        out.write(f"{name},{c},{n},{v / n:.1f}\n")
print(open(os.path.join(O, "r03_pmc_slab.csv")).read())
PY
