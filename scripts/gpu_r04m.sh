#!/bin/bash
# Round 4, trip m: the one-workgroup sequence kernels after the lock-step row gathers and the run-split weight-gradient sums
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=$PWD/gpurun_out
(timeout 300 python -m pytest tests/test_dcrnn.py tests/test_baseline_shapes.py -m gpu -q -k "one_workgroup or small_graph or hops or chickenpox or config1" 2>&1 | tail -4) > $O/pytest_gpu_sel.log; tail -2 $O/pytest_gpu_sel.log
for B in 64 256 1024; do timeout 100 python scripts/small_batch_probe.py $B 2 100 2>&1 | tail -1; done
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/r04m_seq -- python $OLDPWD/scripts/small_batch_probe.py 64 2 50 eager) > $O/prof/r04m_seq.log 2>&1; echo "stats rc=$?"
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.getcwd(), "gpurun_out", "prof", "r04m_seq", "**", "*kernel_stats.csv"), recursive=True):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 4: print(row["Name"][:80], row["Calls"], row["AverageNs"], row["Percentage"])
PY
