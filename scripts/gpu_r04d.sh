#!/bin/bash
# Round 4, fourth GPU trip: the default bench under a clock with stage marks on stderr; config 4 with its hipGraph variant.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "^\[bench" $O/bench.err | tail -40
SECONDS=0
(timeout 200 python bench.py --config tgcn50k --no-cpu-baseline) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$? wall=${SECONDS}s"; head -c 300 $O/bench_tgcn.json; tail -2 $O/bench_tgcn.err
