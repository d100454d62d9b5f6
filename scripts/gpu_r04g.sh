#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof/tgcn_cell -- python $OLDPWD/scripts/r04_probe.py tgcn_cell) > $O/prof/tgcn_cell.log 2>&1
echo "rocprof rc=$?"
find $O/prof/tgcn_cell -name "*kernel_trace.csv" -size +8M -delete
f=$(find $O/prof/tgcn_cell -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-160
