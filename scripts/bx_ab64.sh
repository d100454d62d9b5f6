for t in "gemm_bx_sym=1" "gemm_bx_sym=0" "gemm_bx=0" "gemm_bx_sym=1"; do
  PGT_TUNE="$t" python bench.py --global-batch 64 --steps 60 --warmup 10 --profile-steps 0 --no-extra --no-cpu-baseline > gpurun_out/ab64.json 2>gpurun_out/ab64.err || tail -3 gpurun_out/ab64.err
  python -c "
import json;d=json.load(open('gpurun_out/ab64.json'));print('$t', round(d['ms_per_step'],3))"
done
