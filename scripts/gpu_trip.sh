#!/bin/bash
# One gpurun call, stages chosen on the command line (in the order given); everything lands under gpurun_out/.
# tests      pytest -m gpu (tail of the log)                smoke    __graft_entry__.smoke()
# bench      python bench.py (default run, clocked)          tgcn     python bench.py --config tgcn50k
# stats/pmc  rocprofv3 --stats / --pmc passes of the headline command (TAG=${TAG:-r06}); *_tgcn: of --config tgcn50k
# x:<cmd>    any other command, quoted
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out
mkdir -p $O/prof
TAG=${TAG:-r06}
for stage in "$@"; do
  SECONDS=0
  case "$stage" in
    tests) (timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} 2>&1 | tail -40) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log ;;
    smoke) (timeout 300 python __graft_entry__.py --smoke) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
    bench) (timeout 600 python bench.py ${BENCH_ARGS:-}) > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes=$(wc -c < $O/bench.json)"
           cat $O/bench.json; grep "^\[bench \|(aux)" $O/bench.err | tail -40 ;;
    tgcn)  (timeout 400 python bench.py --config tgcn50k ${TGCN_ARGS:-}) > $O/bench_tgcn.json 2> $O/bench_tgcn.err; echo "tgcn rc=$?"; cat $O/bench_tgcn.json ;;
    covid:*) bash scripts/covid_fault_hunt.sh "${stage#covid:}" ${COVID_ENV:-} ;;
    stats) (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof/${TAG}_stats -- python $OLDPWD/bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra) > $O/prof/${TAG}_stats.log 2>&1; echo "stats rc=$?"
           find $O/prof -name "*kernel_trace.csv" -size +8M -delete ;;
    pmc)   TAG=$TAG bash scripts/pmc_bench.sh 2>&1 | tail -5 ;;
    stats_tgcn) (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof/${TAG}_tgcn50k_stats -- python $OLDPWD/bench.py --config tgcn50k --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra) > $O/prof/${TAG}_tgcn50k_stats.log 2>&1; echo "stats_tgcn rc=$?"
           find $O/prof -name "*kernel_trace.csv" -size +8M -delete ;;
    pmc_tgcn) TAG=${TAG}_tgcn50k BENCH_ARGS="--config tgcn50k" bash scripts/pmc_bench.sh 2>&1 | tail -5 ;;
    x:*)   bash -c "${stage#x:}" ;;
    *)     echo "unknown stage $stage" ;;
  esac
  echo "== stage $stage: ${SECONDS}s"
done
