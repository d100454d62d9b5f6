#!/bin/bash
# Out-of-bounds audit of the kernels: the SAME .hip sources compiled for the CPU test double (tests/hipemu) with
# AddressSanitizer, then the kernel / model parity tests run against that build.  Every tensor the tests hand to a kernel is
# a separate heap allocation with red zones, so a clamped-load bug (a lane reading past the end of an operand, a tile
# writing past M or N) that a GPU would silently tolerate -- or fault on, once in a while -- stops the run here.
#   scripts/asan_audit.sh [pytest args]        default: the kernel, DCRNN, model and edge-case suites
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/pgt_asan_emu
mkdir -p "$OUT"
g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -shared -x c++ -DPGT_EMU \
    -I tests/hipemu -I include -I pytorch_geometric_temporal_amd/csrc pytorch_geometric_temporal_amd/csrc/*.hip \
    -o "$OUT/libpgt_emu.so"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
PGT_EMU_LIB="$OUT/libpgt_emu.so" \
python -m pytest ${@:-tests/test_kernels.py tests/test_dcrnn.py tests/test_models.py tests/test_edge_cases.py} -x -q -m "not gpu"
