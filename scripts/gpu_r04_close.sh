#!/bin/bash
# Round 4, closing trip (after the aux blocks moved into a child process): the default bench under a clock, config 4, the new
# example, and LAST the diagnostic of the withdrawn graphed-edge-inputs block.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "bench" $O/bench.err | tail -34
(timeout 200 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 260 $O/bench_tgcn.json; echo
(timeout 120 python examples/tgcn_index_batched_synthetic.py --nodes 5000 --windows 32) > $O/example_tgcn.log 2>&1; echo "example rc=$?"; tail -2 $O/example_tgcn.log
for args in "6 1" "53 0" "53 1"; do
  (timeout 120 python scripts/covid_graph_inputs_repro.py $args) > $O/covid_repro_$(echo $args | tr ' ' '_').log 2>&1
  echo "repro $args rc=$?"; tail -2 $O/covid_repro_$(echo $args | tr ' ' '_').log
done
