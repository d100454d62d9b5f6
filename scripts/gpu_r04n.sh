#!/bin/bash
# Round 4, trip n (closing, after the one-workgroup sequence kernels changed): every GPU test that reaches them, then the default bench.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=$PWD/gpurun_out
(timeout 400 python -m pytest tests/test_dcrnn.py tests/test_graphed.py tests/test_baseline_shapes.py tests/test_models.py tests/test_distributed.py tests/test_edge_cases.py -m gpu -q 2>&1 | tail -6) > $O/pytest_gpu_sel.log; tail -3 $O/pytest_gpu_sel.log
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "^\[bench\|(aux)" $O/bench.err | grep -v "model, optimizer\|initialisation pass" | tail -24
