#!/bin/bash
# Round 4: A/B of the slot-count row order in the stack kernels (same box), then the full default bench, config 4.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
for v in 0 1; do
  (PGT_TUNE="slab_sort=$v" timeout 200 python bench.py --no-cpu-baseline --no-ns --no-extra --steps 20 --warmup 5) > $O/bench_sort$v.json 2> $O/bench_sort$v.err
  python - <<PY
import json
d = json.load(open("$O/bench_sort$v.json"))
k = d["kernels"]
print("slab_sort=$v", round(d["ms_per_step"], 3), "ms/step; stack", round(k["stack"]["total_ms"] / 2, 3), "ms", round(k["stack"]["hbm_frac"], 3), "avg us", round(k["stack"]["avg_us"], 1))
PY
done
(timeout 200 python -m pytest tests -m gpu -q -x -k "slab or stack or dcrnn or config2" 2>&1 | tail -3) > $O/pytest_gpu_sel.log
cat $O/pytest_gpu_sel.log
