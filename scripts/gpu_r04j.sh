#!/bin/bash
# Round 4, trip j (diagnostic): where the headline's set-up spent 190 s in trip i.  Python stacks every 20 s, with and without the NS block.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=$PWD/gpurun_out
SECONDS=0
(PGT_BENCH_STACKS=20 PGT_BENCH_STEP_TIMES=1 timeout 150 python bench.py --no-extra --no-cpu-baseline --profile-steps 0) > $O/diag_ns.json 2> $O/diag_ns.err
echo "with NS rc=$? wall=${SECONDS}s"; grep "^\[bench" $O/diag_ns.err | head -20; grep -c "most recent call first" $O/diag_ns.err
SECONDS=0
(PGT_BENCH_STACKS=20 PGT_BENCH_STEP_TIMES=1 timeout 150 python bench.py --no-extra --no-cpu-baseline --profile-steps 0 --no-ns) > $O/diag_nons.json 2> $O/diag_nons.err
echo "without NS rc=$? wall=${SECONDS}s"; grep "^\[bench" $O/diag_nons.err | head -20; grep -c "most recent call first" $O/diag_nons.err
grep -A 25 "most recent call first" $O/diag_ns.err | head -80
