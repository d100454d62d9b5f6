// Shared helpers for the gfx950 kernels of libpgt_hip.so.  Written for CDNA4 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "pgt_hip.h"

#ifdef PGT_EMU
// tests/hipemu: kernels run as fibers on the CPU (test double, never shipped).
#define PGT_LAUNCH(kern, grid, block, stream, ...) \
  pgt_emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })
typedef pgt_emu_f32x16 pgt_f32x16;
#define PGT_MFMA_32x32x2(a, b, c) pgt_emu_mfma_32x32x2((a), (b), (c))
#define PGT_TARGET "emu"
#else
#define PGT_LAUNCH(kern, grid, block, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
typedef float pgt_f32x16 __attribute__((ext_vector_type(16)));
#define PGT_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define PGT_TARGET "gfx950"
#endif

#define PGT_WAVE 64

void pgt_set_error(const char* fmt, ...);

#define PGT_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      pgt_set_error(__VA_ARGS__);         \
      return PGT_ERR_INVALID;             \
    }                                     \
  } while (0)

static inline int pgt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pgt_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return PGT_ERR_LAUNCH;
  }
  return PGT_OK;
}

static inline int64_t pgt_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool pgt_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
