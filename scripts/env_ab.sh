# same-box A/B of environment switches inside the bench: ENV_AB="PGT_FOLD_DH=1 PGT_FOLD_DH=0 PGT_FOLD_DH=1" scripts/env_ab.sh
for t in ${ENV_AB}; do
  env "$t" python bench.py --no-extra --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ab.json 2>gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));k=d['kernels']
print('$t', round(d['ms_per_step'],3), 'gemm', round(k['gemm']['total_ms'],2), 'tn', round(k['gemm_tn']['total_ms'],2), 'stack', round(k['stack']['total_ms'],2), 'timed', round(d['roofline']['all_kernel_classes']['ms_per_step_in_timed_kernels'],2))"
done
