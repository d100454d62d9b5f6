#!/bin/bash
# Round 4, second GPU trip: full GPU suite on the new kernels (seq_small, tgcn_cell), headline + config-4 benches.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
(timeout 300 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 400 $O/bench_tgcn.json; tail -3 $O/bench_tgcn.err
(timeout 700 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 300 $O/bench.json; tail -3 $O/bench.err
(timeout 120 ./lab/store_lab) > $O/store_lab.jsonl 2>&1
echo "store lab rc=$?"; cat $O/store_lab.jsonl
