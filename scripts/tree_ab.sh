# same-box A/B of two source trees (this checkout against a copy of an older commit under _ab_old/): bench.py --no-extra in each
R=$PWD
for t in ${TREES:-. _ab_old . _ab_old}; do
  (cd $R/$t && python bench.py --no-extra --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/ab.json 2>$R/gpurun_out/ab.err) || tail -3 $R/gpurun_out/ab.err
  python -c "
import json;d=json.load(open('$R/gpurun_out/ab.json'));k=d['kernels']
print('$t', round(d['ms_per_step'],3), 'gemm', round(k['gemm']['total_ms'],2), 'tn', round(k['gemm_tn']['total_ms'],2), 'stack', round(k['stack']['total_ms'],2), ' '.join(s['shape'][0]+str(s['shape'][2])+'/'+str(s['shape'][4])+':'+str(round(s['avg_us'],1)) for s in k['gemm']['by_shape'][:6]))"
done
