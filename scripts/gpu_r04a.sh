#!/bin/bash
# Round 4, first GPU trip: full GPU parity suite, the headline bench, config 4 as SURVEY 8(d) (bench.py --config tgcn50k) with its
# rocprofv3 kernel stats, the ST-Conv probe.  Outputs under gpurun_out/.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
(timeout 300 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; tail -c 600 $O/bench_tgcn.json
(timeout 200 python scripts/stconv_probe.py) > $O/stconv_probe.jsonl 2> $O/stconv_probe.err
echo "stconv probe rc=$?"
(timeout 600 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 600 $O/bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof/tgcn_stats -- \
    python $OLDPWD/bench.py --config tgcn50k --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline) > $O/prof/tgcn_stats.log 2>&1
echo "rocprof rc=$?"
find $O/prof/tgcn_stats -name "*kernel_trace.csv" -size +8M -delete
find $O/prof -name "*.csv" | head
