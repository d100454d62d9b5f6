#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
(timeout 300 python scripts/r04_probe.py) > $O/r04_probe.jsonl 2> $O/r04_probe.err
echo "probe rc=$?"; cat $O/r04_probe.jsonl; tail -3 $O/r04_probe.err
(timeout 200 python -m pytest tests -m gpu -q -x -k "one_workgroup or one_launch or fused_tgcn or config4 or config3 or fixture" 2>&1 | tail -4) > $O/pytest_gpu_sel.log
cat $O/pytest_gpu_sel.log
