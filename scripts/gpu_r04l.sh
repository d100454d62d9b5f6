#!/bin/bash
# Round 4, trip l: the one-workgroup sequence kernels with LDS-only barriers, reads a phase ahead and register sums (seq_ahead) — A/B
# at the reference's own model (hidden 2) for B = 64 / 128 / 256 / 1024, plus the parity tests of those kernels, plus rocprofv3 stats.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=$PWD/gpurun_out
(timeout 300 python -m pytest tests/test_dcrnn.py -m gpu -q -k "one_workgroup or small_graph or hops" 2>&1 | tail -4) > $O/pytest_gpu_sel.log; tail -2 $O/pytest_gpu_sel.log
for B in 64 256 1024; do
  for A in 0 1; do
    echo -n "seq_ahead=$A "; PGT_TUNE=seq_ahead=$A timeout 100 python scripts/small_batch_probe.py $B 2 100 2>&1 | tail -1
  done
done
echo -n "default "; timeout 100 python scripts/small_batch_probe.py 64 2 100 2>&1 | tail -1
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/r04l_seq -- python $OLDPWD/scripts/small_batch_probe.py 64 2 50 eager) > $O/prof/r04l_seq.log 2>&1; echo "stats rc=$?"
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.getcwd(), "gpurun_out", "prof", "r04l_seq", "**", "*kernel_stats.csv"), recursive=True):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8: print(row["Name"][:80], row["Calls"], row["AverageNs"], row["Percentage"])
PY
