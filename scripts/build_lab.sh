#!/bin/bash
# builds lab/ns_lab (kernel lab harness; cross-compiles without a GPU)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I pytorch_geometric_temporal_amd/csrc lab/ns_lab.hip -o lab/ns_lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I pytorch_geometric_temporal_amd/csrc lab/gemm_lab.hip -o lab/gemm_lab
