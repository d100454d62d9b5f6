#!/bin/bash
# Memory-safety audit of the kernels: the SAME .hip sources compiled for the CPU test double (tests/hipemu) with a
# sanitizer, then the kernel / model parity tests (fuzz tests included) run against that build.
#   scripts/asan_audit.sh [asan|ubsan|order] [pytest args]     default: asan; kernel, DCRNN, model and edge-case suites
# asan : every tensor the tests hand to a kernel is a separate heap allocation with red zones, so a clamped-load bug
#        (a lane reading past the end of an operand, a tile writing past M or N) that a GPU would silently tolerate --
#        or fault on, once in a while -- stops the run.
# ubsan: float2 / float4 are declared naturally aligned (PGT_EMU_STRICT_ALIGN), so a vector access the host-side
#        dispatch should not have allowed (8- / 16-byte alignment of operands, strides and segment offsets) is
#        reported, together with signed overflow in the index arithmetic, invalid shifts and bounds of fixed arrays.
# order: no sanitizer; the test double hands out turns to the wavefronts of a workgroup in reverse and in rotating order
#        (PGT_EMU_ORDER): a result that changes with the order is a missing __syncthreads() between wavefronts.
set -e
cd "$(dirname "$0")/.."
MODE=${1:-asan}
case "$MODE" in asan|ubsan|order) shift || true ;; *) MODE=asan ;; esac
if [ $# -gt 0 ]; then ARGS=("$@"); else ARGS=(tests/test_kernels.py tests/test_dcrnn.py tests/test_models.py tests/test_edge_cases.py); fi
if [ "$MODE" = order ]; then
  for o in reverse rotate; do
    echo "== wavefront order: $o"
    PGT_EMU_ORDER=$o python -m pytest "${ARGS[@]}" -x -q -m "not gpu"
  done
  exit 0
fi
OUT=${TMPDIR:-/tmp}/pgt_${MODE}_emu
export PGT_EMU_SANITIZER=$MODE      # tests skip checks that need a C++ exception to cross torch's autograd engine
mkdir -p "$OUT"
if [ "$MODE" = asan ]; then
  SAN="-fsanitize=address"; RT=$(gcc -print-file-name=libasan.so)
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1
else
  SAN="-fsanitize=undefined -fno-sanitize-recover=undefined -DPGT_EMU_STRICT_ALIGN"; RT=$(gcc -print-file-name=libubsan.so)
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
g++ -std=c++17 -O1 -g $SAN -fno-omit-frame-pointer -fPIC -shared -x c++ -DPGT_EMU \
    -I tests/hipemu -I include -I pytorch_geometric_temporal_amd/csrc pytorch_geometric_temporal_amd/csrc/*.hip \
    -o "$OUT/libpgt_emu.so"
LD_PRELOAD=$RT PGT_EMU_LIB="$OUT/libpgt_emu.so" \
python -m pytest "${ARGS[@]}" -x -q -m "not gpu"
