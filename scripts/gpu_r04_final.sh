#!/bin/bash
# Round 4, closing trip: full GPU parity suite, smoke(), the default bench (clocked), config 4, rocprofv3 stats + PMC passes of both.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 200 python __graft_entry__.py --smoke) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "^\[bench" $O/bench.err | tail -30
(timeout 200 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 260 $O/bench_tgcn.json; echo
TAG=r04 bash scripts/pmc_bench.sh > $O/pmc_r04.log 2>&1; tail -3 $O/pmc_r04.log
TAG=r04_tgcn50k BENCH_ARGS="--config tgcn50k" bash scripts/pmc_bench.sh > $O/pmc_r04_tgcn.log 2>&1; tail -3 $O/pmc_r04_tgcn.log
ls $O/prof | head -30
