#!/bin/bash
# Round 4, third GPU trip: the fused T-GCN cell after the load-batching fix, the one-workgroup sequence kernels at B = 64, the
# full default bench under a clock (it has to finish within minutes), selected GPU tests.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=gpurun_out
(timeout 300 python scripts/r04_probe.py) > $O/r04_probe.jsonl 2> $O/r04_probe.err
echo "probe rc=$?"; cat $O/r04_probe.jsonl; tail -3 $O/r04_probe.err
(timeout 400 python -m pytest tests -m gpu -q -x -k "tgcn or stconv or batchnorm or a3tgcn or config3 or config4 or one_workgroup or one_launch" 2>&1 | tail -6) > $O/pytest_gpu_sel.log
cat $O/pytest_gpu_sel.log
SECONDS=0
(timeout 500 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 300 $O/bench.json; tail -3 $O/bench.err
(timeout 200 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 300 $O/bench_tgcn.json
