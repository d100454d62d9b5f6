#!/bin/bash
# Round 4, trip k (closing): full GPU parity suite, smoke(), the default bench under a clock, config 4, and the multi-rank bench
# protocol on the one GPU of the box (two ranks sharing cuda:0 through gloo — PGT_BENCH_BACKEND — so that barrier, flat-gradient
# all-reduce, MAX-over-ranks timing and rank-0 printing run on hardware; RCCL itself needs > 1 device).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=$PWD/gpurun_out
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 200 python __graft_entry__.py --smoke) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "^\[bench\|(aux)" $O/bench.err | tail -40
(timeout 200 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 260 $O/bench_tgcn.json; echo
for cfg in dcrnn_metrla tgcn50k; do
  (PGT_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
     bench.py --gpus 2 --steps 5 --warmup 2 --config $cfg --no-extra --no-cpu-baseline --no-ns --profile-steps 0) > $O/bench_2ranks_$cfg.json 2> $O/bench_2ranks_$cfg.err
  echo "2 ranks on one GPU ($cfg) rc=$?"; tail -1 $O/bench_2ranks_$cfg.json | head -c 420; echo
done
