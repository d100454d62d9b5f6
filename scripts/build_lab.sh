#!/bin/bash
# builds the kernel lab harnesses under lab/ (cross-compiles without a GPU; the binaries are git-ignored and travel to the
# GPU box with gpurun)
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I pytorch_geometric_temporal_amd/csrc"
for h in ellw_lab ellw_prod_lab gemm_lab gemm_bx_lab gemm_bx_tn_lab slab_lab; do
  /opt/rocm/bin/hipcc $FLAGS lab/$h.hip -o lab/$h || exit 1
done
