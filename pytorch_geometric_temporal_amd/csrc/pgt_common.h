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
#define PGT_UNIFORM(x) (x)
#else
// wave-uniform value -> SGPR, so loads indexed by it go through the scalar cache (s_load) instead of 64 identical
// vector lanes
#define PGT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define PGT_LAUNCH(kern, grid, block, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
typedef float pgt_f32x16 __attribute__((ext_vector_type(16)));
#define PGT_MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define PGT_TARGET "gfx950"
#endif

#define PGT_WAVE 64

// Wavefront-private LDS hand-off (one lane writes, another lane of the SAME wavefront reads): the hardware keeps a
// wavefront's LDS operations in order; the test double runs lanes as fibers and needs a rendezvous.
// PGT_LDS_BARRIER(): workgroup barrier for data that travels through LDS only.  __syncthreads() also drains vmcnt (its
// fence covers global memory), so a kernel that has global stores in flight pays their write acknowledgements at every
// barrier; here only the LDS counter is waited for, loads and stores stay in flight across the barrier.
#ifdef PGT_EMU
#define PGT_WAVE_SYNC() pgt_emu::wave_barrier()
#define PGT_SCHED_FENCE() ((void)0)
#define PGT_LDS_BARRIER() __syncthreads()
#else
#define PGT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PGT_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// compiler-only fence: no instruction is scheduled across it (pins software-pipelined LDS reads ahead of MFMAs)
#define PGT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// lab/ns_lab.hip defines this to record an in-kernel timeline; a no-op in the library
#ifndef PGT_TRACE_MARK
#define PGT_TRACE_MARK(slot) do { } while (0)
#endif

// 16-byte register vector.  HIP's float4 is a struct: arrays of it that live across a loop are copied with memcpy and
// end up in scratch / promoted to LDS; the native vector type stays in VGPRs.
#ifdef PGT_EMU
typedef float4 pgt_f4;
#else
typedef float pgt_f4 __attribute__((ext_vector_type(4)));
#endif
// 4 x uint32 register vector (slot vectors of the ELLW layout)
#ifdef PGT_EMU
struct pgt_u4 { unsigned x, y, z, w; };
#else
typedef unsigned int pgt_u4 __attribute__((ext_vector_type(4)));
#endif
// a product / a sum that must stay two roundings (hipcc contracts a * b + c into an fma by default)
#ifdef PGT_EMU
static inline float pgt_mul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float pgt_add_rn(float a, float b) { volatile float r = a + b; return r; }
#else
static __device__ __forceinline__ float pgt_mul_rn(float a, float b) { return __fmul_rn(a, b); }
static __device__ __forceinline__ float pgt_add_rn(float a, float b) { return __fadd_rn(a, b); }
#endif
// the one sigmoid of the library (gate kernels and fused GEMM epilogues must agree bit for bit)
static __device__ __forceinline__ float pgt_sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// GRU blend Z*H + (1-Z)*T with the contraction spelled out (the gate kernel and the fused GEMM epilogue must round alike)
static __device__ __forceinline__ float pgt_gru_blend(float z, float h, float t) { return fmaf(z, h, (1.f - z) * t); }

static __device__ __forceinline__ pgt_f4 pgt_mk4(float a, float b, float c, float d) {
  pgt_f4 r;
  r.x = a; r.y = b; r.z = c; r.w = d;
  return r;
}

void pgt_set_error(const char* fmt, ...);
// tuning knobs (pgt_tune): each translation unit owns its own
void pgt_gemm_set_force_small(int v);
void pgt_gemm_set_small_fill(int v);
void pgt_gemm_set_tn_fullk(int v);
void pgt_gemm_set_db(int v);
void pgt_gemm_set_db64(int v);
void pgt_slab_set_pairs(int v);
void pgt_slab_set_split(int v);
void pgt_slab_set_threads(int v);
void pgt_slab_set_wpc(int v);
void pgt_slab_set_quad(int v);
void pgt_slab_set_gu(int v);
void pgt_slab_set_sort(int v);
void pgt_tgcn_set_rows(int v);
int pgt_tgcn_set_probe(int v);    // 0: not compiled in (product library)
void pgt_tgcn_set_wgs(int v);
void pgt_gemm_set_tn_pipe(int v);
void pgt_gemm_set_skinny(int v);
void pgt_gemm_set_dbp(int v);
int pgt_spmm_tune(const char* key, int value);  // returns 1 when the key is known

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

// Arguments of the GEMM tile kernels (gemm.hip) and of the split-bf16 kernel (gemm_bx.hip).
struct PgtGemmArgs {
  const float* A; int64_t lda; int64_t a_seg_stride; int n_seg; int seg_k;
  const float* Bw; int64_t sbk; int64_t sbn;
  float* C; int64_t ldc; int64_t c_seg_stride; int c_seg_n;
  const float* bias; int M; int N; int accumulate;
  // fused GRU epilogues (pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32); epi = 0: plain GEMM
  //   1: C = sigmoid(acc + bias) [M, 2O];  eX[m, efin + o] = eH[m, o] * C[m, O + o]
  //   2: C = tanh(acc + bias) [M, O];  Hnew = Z * H + (1 - Z) * C with Z = eZ[m * 2O + o]  -> eO0 (and eO1 when non-null)
  int epi; int eO; int efin; int evec;   // evec bit 0: eH float4-loadable, 1: eX float2-storable, 2: eO0 float4, 3: eO1 float2
  const float* eH; int64_t eldh; float* eX; int64_t eldx;
  const float* eZ; float* eO0; int64_t eld0; float* eO1; int64_t eld1;
  // two-level row layout of eO0 (pgt_rowmap): row m at eO0 + (m / e0_period) * e0_hi + (m % e0_period) * eld0; 0 = plain
  int64_t e0_period; int64_t e0_hi;
};

// pgt_rowmap on the device: float offset of row m of an operand with row stride ld (period <= 0: a plain matrix)
__device__ __forceinline__ int64_t pgt_row_off(int64_t m, int64_t ld, int64_t period, int64_t stride_hi) {
  if (period <= 0) return m * ld;
  const uint32_t q = (uint32_t)m / (uint32_t)period;                   // m < 2^31, period < 2^31 (checked by the host)
  return (int64_t)q * stride_hi + (int64_t)((uint32_t)m - q * (uint32_t)period) * ld;
}
// Arguments of the weight-gradient kernels: dW[k, n] += sum_m A[m, k] G[m, n], db[n] += sum_m G[m, n].
struct PgtTnArgs {
  const float* A; int64_t lda; int64_t a_seg_stride; int n_seg; int seg_k;
  const float* G; int64_t ldg; float* dW; int64_t lddw; float* db; int M; int N; int rows_per_slab;
  // deterministic mode (pgt_gemm_tn_det_f32): every (slab, k, n) partial sum is STORED at part + slab * part_stride +
  // k * lddw + n (bias partials at dbpart + slab * N + n) and tn_reduce_kernel adds the slabs in index order; null =
  // fp32 atomics straight into dW / db
  float* part; int64_t part_stride; float* dbpart;
};
// gemm_bx.hip, weight gradient: plan = 1 when the split-bf16 kernel covers the shape (*nslab = partial-sum slabs it
// produces: the caller sizes the deterministic scratch with it), launch as below
int pgt_gemm_bx_tn_plan(const PgtTnArgs& t, int64_t* nslab);
int pgt_gemm_bx_tn_launch(const PgtTnArgs& t, pgt_stream_t stream);
// gemm_bx.hip: 1 = launched, 0 = shape not covered (the caller runs the fp32 MFMA kernels), < 0 = error
int pgt_gemm_bx_launch(const PgtGemmArgs& g, pgt_stream_t stream);
void pgt_gemm_bx_set(int v);
void pgt_gemm_bx_sym_set(int v);
void pgt_gemm_bx_tn_pc_set(int v);
void pgt_gemm_bx_sym_pc_set(int v);

// V-float (4 / 8 / 16-byte) global accesses; V is chosen by the host from pointer and stride alignment.
template <int VEC>
__device__ __forceinline__ void pgt_ldv(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void pgt_stv(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

// largest V in {4,2,1} such that `width` is a multiple of V and every (pointer, row stride) pair is V-float aligned
struct PgtVecPick {
  int v = 4;
  void width(int64_t w) { while (v > 1 && w % v) v >>= 1; }
  void operand(const void* p, int64_t ld) {
    if (p == nullptr) return;
    while (v > 1 && (ld % v || reinterpret_cast<uintptr_t>(p) % (4u * v))) v >>= 1;
  }
  void operand(const void* p, int64_t ld, const pgt_rowmap& m) {
    operand(p, ld);
    if (m.period > 0) while (v > 1 && m.stride_hi % v) v >>= 1;
  }
};

// a caller's pgt_rowmap (NULL = plain rows) by value; false when it cannot be indexed with 32-bit row arithmetic
static inline bool pgt_rowmap_take(const pgt_rowmap* in, int64_t M, pgt_rowmap* out) {
  out->period = 0; out->stride_hi = 0;
  if (in == nullptr || in->period == 0) return true;
  if (in->period < 0 || in->period >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 31)) return false;
  *out = *in;
  return true;
}
