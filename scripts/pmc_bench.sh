#!/bin/bash
# rocprofv3 passes over the bench command itself: --kernel-trace --stats (per-kernel time) and two separate --pmc passes
# (FETCH_SIZE; WRITE_SIZE + L2 hit/miss), as MI355X_MICROARCH.md prescribes (counters in their own runs, no tracing
# domains beside --kernel-trace).  Outputs under gpurun_out/prof/${TAG}_*; scripts/pmc_traffic.py condenses them.
#   TAG=r04 scripts/pmc_bench.sh          TAG=r04_tgcn50k BENCH_ARGS="--config tgcn50k" scripts/pmc_bench.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
TAG=${TAG:-r04}
mkdir -p $O
ARGS="--steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python $OLDPWD/bench.py $ARGS) > $O/${TAG}_stats.log 2>&1
echo "stats rc=$?"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python $OLDPWD/bench.py $ARGS) > $O/${TAG}_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_pmc_write -- python $OLDPWD/bench.py $ARGS) > $O/${TAG}_pmc_write.log 2>&1
echo "pmc write rc=$?"
find $O -name "*kernel_trace.csv" -size +8M -delete
python scripts/pmc_traffic.py "bench.py $ARGS" $TAG
