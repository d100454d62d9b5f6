#!/bin/bash
# A/B of a pgt_tune switch inside the headline step: alternating runs on one box, per-shape product times.
#   scripts/bx_ab.sh gemm_bx_sym2 "1 0 1 0"
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
KEY=${1:-gemm_bx_sym2}
for v in ${2:-1 0 1 0}; do
  echo "$KEY=$v"
  PGT_TUNE=$KEY=$v python bench.py --no-extra --no-cpu-baseline --no-ns 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], d['roofline']['all_kernel_classes']['classes_ms_frac'])"
  python - <<'P'
import json
d=json.load(open('gpurun_out/bench_full.json'))
for s in d['kernels']['gemm']['by_shape']: print('   ', s['shape'], round(s['avg_us'],1), round(s['hbm_frac'],3))
P
done
