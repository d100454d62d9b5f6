#!/bin/bash
# One gpurun call: GPU parity tests, probes, bench, rocprofv3 stats + PMC passes.  Outputs under gpurun_out/.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 700 python scripts/gpu_probe.py ${PROBES:-spmm_ns spmm_batched stack gemm step models}) > $O/probe.jsonl 2> $O/probe.err
echo "probe rc=$?"
(timeout 500 python bench.py ${BENCH_ARGS:-}) > $O/bench.json 2> $O/bench.err
(timeout 300 python examples/dcrnn_chickenpox.py 20) > $O/example_chickenpox.log 2>&1
echo "bench rc=$?"
if [ -n "${AB_TUNE:-}" ]; then (PGT_TUNE="$AB_TUNE" timeout 200 python bench.py --no-cpu-baseline --no-ns --profile-steps 0 ${BENCH_ARGS:-}) > $O/bench_ab.json 2> $O/bench_ab.err; tail -c 300 $O/bench_ab.json; fi
tail -c 400 $O/bench.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof/stats -- \
      python $OLDPWD/bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-ns ${BENCH_ARGS:-}) > $O/prof/stats.log 2>&1
  echo "rocprof stats rc=$?"
  for cfg in "ns-local" "metrla 4096"; do
    tag=$(echo $cfg | tr ' ' '_')
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/prof/pmc_fetch_$tag -- \
        python $OLDPWD/scripts/spmm_only.py $cfg) > $O/prof/pmc_fetch_$tag.log 2>&1
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OLDPWD/$O/prof/pmc_write_$tag -- \
        python $OLDPWD/scripts/spmm_only.py $cfg) > $O/prof/pmc_write_$tag.log 2>&1
  done
  echo "pmc done"
  find $O/prof -name "*.csv" | head -30
  # keep the merge small: drop the raw per-dispatch traces of the bench run (hundreds of thousands of rows)
  find $O/prof/stats -name "*kernel_trace.csv" -size +8M -delete
fi
