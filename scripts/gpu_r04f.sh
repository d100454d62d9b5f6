#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
(timeout 200 python scripts/r04_probe.py tgcn_cell) > $O/r04_probe_f.jsonl 2> $O/r04_probe_f.err
echo "probe rc=$?"; cat $O/r04_probe_f.jsonl; tail -2 $O/r04_probe_f.err
(timeout 300 python -m pytest tests -m gpu -q -x -k "fused_tgcn or config4 or config3 or reproduce_their_output or dynamic_graph_epoch or weight_recurrence or tgcn" 2>&1 | tail -4) > $O/pytest_gpu_sel.log
cat $O/pytest_gpu_sel.log
