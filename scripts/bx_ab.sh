# same-box A/B of the split-bf16 kernels inside the bench (PGT_TUNE switches, see include/pgt_hip.h: pgt_tune)
for t in ${BX_AB:-"gemm_bx=1" "gemm_bx=0" "gemm_bx=1"}; do
  PGT_TUNE="$t" python bench.py --no-extra --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));k=d['kernels']
print('$t', round(d['ms_per_step'],3), 'gemm', round(k['gemm']['total_ms'],2), 'tn', round(k['gemm_tn']['total_ms'],2), 'stack', round(k['stack']['total_ms'],2), ' '.join(s['shape'][0]+str(s['shape'][2])+'/'+str(s['shape'][4])+':'+str(round(s['avg_us'],1)) for s in k['gemm']['by_shape'][:6]))"
done
