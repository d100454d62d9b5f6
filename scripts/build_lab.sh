#!/bin/bash
# builds the kernel lab harnesses under lab/ (cross-compiles without a GPU; the binaries are git-ignored and travel to the
# GPU box with gpurun)
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I pytorch_geometric_temporal_amd/csrc"
for h in ellw_lab ellw_prod_lab gemm_lab gemm_bx_lab gemm_bx_tn_lab slab_lab seq_small_lab; do
  /opt/rocm/bin/hipcc $FLAGS lab/$h.hip -o lab/$h || exit 1
done
# the shipped K-split split-bf16 kernel, taken apart at compile time (BX_SKIP mask: 1 no MFMAs, 4 no epilogue stores, 8 no loads
# of A, 16 no gate-operand loads, 32 no wait for them); mask 0 also prints the per-wavefront timeline
for m in 0 1 4 8 16 20 21 28 9 29 32; do
  /opt/rocm/bin/hipcc $FLAGS -DBX_SKIP=$m lab/gemm_bx_trace_lab.hip -o lab/gemm_bx_trace_lab_$m || exit 1
done
# csrc/seq64.hip with the phase timeline of workgroup 0 (scripts/seq64_trace.py); "_stagger": the withdrawn overlapped forward;
# "_skipN": parts switched off (SQ_LAB_SKIP bits in csrc/seq64.hip)
/opt/rocm/bin/hipcc $FLAGS -shared -fPIC lab/seq64_lab.hip -o lab/libseq64_lab.so || exit 1
/opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DSQ_STAGGER_RECORD lab/seq64_lab.hip -o lab/libseq64_lab_stagger.so || exit 1
for m in ${SQ_SKIPS:-}; do /opt/rocm/bin/hipcc $FLAGS -shared -fPIC -DSQ_SKIP=$m lab/seq64_lab.hip -o lab/libseq64_lab_skip$m.so || exit 1; done
