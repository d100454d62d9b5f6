// LAB HARNESS (not shipped): split-bf16 GEMM prototype — the experiment log behind csrc/gemm_bx.hip.
// Compile-time switches select the variants that were measured (BX_KA, BX_PRIO, BX_TL / BX_TL2 timelines, BX_NOEPI,
// BX_NOCONV, BX_SLOWSPLIT); the shipped kernel is the default configuration plus the fused epilogues.
//
// fp32 operands are split into three bf16 pieces each (x = x1 + x2 + x3, 8 significant bits per piece = the 24 bits of an
// fp32 significand) and the product is accumulated from the six largest piece products on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate) — the bf16 pipe runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six products are 2.67x the fp32
// MFMA peak.  The dropped products (x2 y3, x3 y2, x3 y3) are below 2^-25 |x y|.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

#define LAB_HAS_GEMM
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/gemm.hip"
// gemm.hip routes tall products to csrc/gemm_bx.hip; this harness links without it (its own copy of the kernels below)
int pgt_gemm_bx_launch(const PgtGemmArgs&, pgt_stream_t) { return 0; }
int pgt_gemm_bx_tn_plan(const PgtTnArgs&, int64_t*) { return 0; }
int pgt_gemm_bx_tn_launch(const PgtTnArgs&, pgt_stream_t) { return PGT_ERR_INVALID; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef BX_KA
#define BX_KA 12
#endif
namespace {

typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bx_pack(float x, float y) {
  bx_f32x2 v = {x, y};
  bx_bf16x2 r = __builtin_convertvector(v, bx_bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
// (x, y) -> three packed bf16 pairs (low half = x's piece)
__device__ __forceinline__ void bx_split2(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = bx_pack(x, y);
  float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  rx = (fabsf(rx) <= 3.0e38f) ? rx : 0.f;      // inf / nan operand: the first piece carries it, the others are zero
  ry = (fabsf(ry) <= 3.0e38f) ? ry : 0.f;
  p2 = bx_pack(rx, ry);
  rx -= __uint_as_float(p2 << 16);
  ry -= __uint_as_float(p2 & 0xffff0000u);
  p3 = bx_pack(rx, ry);
}

__device__ long long* g_bx_clk = nullptr;
__device__ long long* g_bx_tl = nullptr;
#ifdef BX_TL2
#define BX_MARK(k) do { if (n_iter == 10) tl[(k)] = clock64(); } while (0)
#define BX_KMARK(k) do { if (n_iter == 10) tl[8 + (k)] = clock64(); } while (0)
#elif defined(BX_TL)
#define BX_KMARK(k) do { } while (0)
#define BX_MARK(k) do { if (g_bx_tl != nullptr && blockIdx.x == 0 && n_iter == 10 && lane == 0) g_bx_tl[wave * 8 + (k)] = clock64(); } while (0)
#else
#define BX_MARK(k) do { } while (0)
#define BX_KMARK(k) do { } while (0)
#endif
typedef uint32_t bx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bx_u32x2 __attribute__((ext_vector_type(2)));

// the streaming operand: first piece rounded to nearest, the other two cut off (x - x1 has at most 16 significant bits, the
// second cut leaves at most 9): 9 instructions per pair.  An infinite x gives x - x1 = nan, i.e. a nan row where the
// fp32 product would have +-inf or nan.
__device__ __forceinline__ void bx_split2_fast(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = bx_pack(x, y);
  float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
  rx -= __uint_as_float(p2 << 16);
  ry -= __uint_as_float(p2 & 0xffff0000u);
  p3 = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
}

__device__ __forceinline__ pgt_f32x16 bx_mfma(bx_u32x4 a, bx_u32x4 b, pgt_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bx_bf16x8, a), __builtin_bit_cast(bx_bf16x8, b), c, 0, 0, 0);
}

// Store one wavefront's 32 x (32 WN) block: the accumulators (lane = column) are turned through LDS 16 rows at a time
// into row-contiguous float4 pieces.  Plain 16-byte aligned output only (the host checks).
template <int WN>
__device__ __forceinline__ void bx_store_block(const GemmArgs& g, pgt_f32x16 (&acc)[1][WN], float* stage, int row0, int col0, int lane) {
  constexpr int EPW = 32 * WN + 4, LPR = 32 * WN / 4, RPP = 64 / LPR;
  const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    PGT_WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) stage[((r8 & 3) + 8 * (r8 >> 2) + 4 * hi) * EPW + j * 32 + lo] = acc[0][j][8 * h + r8];
    PGT_WAVE_SYNC();
#pragma unroll
    for (int rr = 0; rr < 16; rr += RPP) {
      const int row = rr + lane / LPR, c = (lane % LPR) * 4;
      const int gm = row0 + 16 * h + row, gn = col0 + c;
      const float4 v = *reinterpret_cast<const float4*>(stage + row * EPW + c);
      if (gm < g.M && gn < g.N) *reinterpret_cast<float4*>(g.C + (int64_t)gm * g.ldc + gn) = v;
    }
  }
}

// LDS-only workgroup barrier: the planes and partial sums travel through LDS (lgkmcnt); global loads of the blocks
// ahead and the epilogue's stores stay in flight across it (__syncthreads would drain vmcnt as well)
__device__ __forceinline__ void bx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// C[M, N] = A[M, NSEG * SEGK] . Bw + bias with the epilogues of GemmArgs.  One persistent 512-thread workgroup per CU =
// two wavefronts per SIMD.  Wavefronts 0 .. 3 ("consumers") own 32 * WN columns each and the first KA k-steps, run the
// epilogue and nothing else; wavefronts 4 .. 7 ("producers") own the same columns for the remaining k-steps and bring the
// next 32-row block in: global fp32 -> three bf16 planes in the other LDS buffer.  The B slice of a wavefront (its
// columns x its part of K) lives in registers for the whole launch.  The two K parts meet in LDS: the producer leaves
// its partial sums there and moves on, the consumer adds them and stores while the producer already converts.
template <int KSTEPS, int KA, int NSEG, int SEGK, int WN>
__global__ __launch_bounds__(512, 1) void gemm_bx_kernel(GemmArgs g, int n_blocks) {
  constexpr int BM = 32, KP = KSTEPS * 16, SROW = KP * 2 + 16, PLANE = BM * SROW, BUF = 3 * PLANE;
  constexpr int HALF = SEGK / 2, RPAIRS = NSEG * HALF, EPT = (KP / 2) / 8;   // pairs per row; per producer thread (8 per row)
  constexpr int PART = 64 * 16 * WN * 4, STAGE_PART = 4 * PART, STAGE_EPI = 0;   // a column's partial sums, accumulator layout
  constexpr int KB = KSTEPS - KA, KMAX = KA > KB ? KA : KB;
  static_assert(NSEG * SEGK <= KP && SEGK % 2 == 0 && (KP / 2) % 8 == 0 && KA >= 1 && KB >= 1, "shape");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + STAGE_PART + STAGE_EPI + 16];
  unsigned char* const stage_part = lds + 2 * BUF;
  typedef __attribute__((address_space(3))) volatile int bx_lds_vint;    // an LDS access (a generic pointer would be a FLAT load that drains vmcnt)
  bx_lds_vint* const part_seen = (bx_lds_vint*)(lds + 2 * BUF + STAGE_PART + STAGE_EPI);   // per column: blocks whose partial sums the consumer has picked up
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3;
  const bool producer = wave >= 4;
  const int nwg = gridDim.x;
  // ---- B slice -> registers (consumer: k-steps [0, KA), producer: [KA, KSTEPS))
  bx_u32x4 bf[KMAX][WN][3];
  {
    const int Ktot = NSEG * SEGK, kbase = producer ? KA : 0;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = (wc * WN + j) * 32 + (lane & 31), k0 = (kbase + i) * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int k = k0 + t;
          v[t] = (k < Ktot && col < g.N && i < (producer ? KB : KA)) ? g.Bw[(int64_t)k * g.sbk + (int64_t)col * g.sbn] : 0.f;
        }
        uint32_t p[3][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bx_split2(v[2 * t], v[2 * t + 1], p[0][t], p[1][t], p[2][t]);
#pragma unroll
        for (int q = 0; q < 3; ++q) { bx_u32x4 f = {p[q][0], p[q][1], p[q][2], p[q][3]}; bf[i][j][q] = f; }
      }
  }
  // ---- zero both A buffers once (the K padding columns are never written again)
  for (int i = tid; i < 2 * BUF / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 4) part_seen[tid] = 0;
  int rb = blockIdx.x;
  if (rb >= n_blocks) return;
  __syncthreads();
  const int arow = (lane & 31) * SROW + 16 * (lane >> 5);
  const long long c_begin = clock64(), w_begin = wall_clock64();
  int n_iter = 0;
  long long tl[24];
  for (int i = 0; i < 24; ++i) tl[i] = 0;
  if (producer) {
    // ---- element map of a 32-row block over the 256 producer threads: row = ptid / 8, pairs (ptid % 8) + 8 t
    const int ptid = tid - 256, erow = ptid >> 3, el = ptid & 7;
    uint32_t goff[EPT];   // byte offset from the block base; past the row's last pair: outside the descriptor (reads 0)
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
      const int pi = el + 8 * t, seg = pi / HALF, pp = pi % HALF;
      goff[t] = pi < RPAIRS ? (uint32_t)((seg * g.a_seg_stride + erow * g.lda + 2 * pp) * 4) : 0xfffffff0u;
    }
    const uint32_t lbase = (uint32_t)(erow * SROW + el * 4);
    // a block's rows are read through a buffer descriptor that ends with the last valid row of the last segment: rows
    // past M (ragged last block) and whole blocks past the end read as zero, without a branch
    const int64_t span_last = (int64_t)(NSEG - 1) * g.a_seg_stride * 4;
    auto block_rsrc = [&](int b) {
      const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
      const int64_t bytes = rows_left > 0 ? span_last + rows_left * g.lda * 4 : 0;
      const uint64_t base = reinterpret_cast<uint64_t>(g.A + (int64_t)(rows_left > 0 ? b : 0) * BM * g.lda);
      bx_u32x4 r = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) & 0xffffu,
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bytes), 0x00020000u};
      return r;
    };
    // The block loads are issued with inline asm and waited for by hand: the compiler's counter insertion drains vmcnt
    // at every loop back-edge, which would serialise the loads of the block after next with this block's k-steps.
    // Loads return in order and every conversion is followed by the reload of its register pair, so exactly EPT - 1
    // younger loads are in flight when element t of the previous round is due.
    bx_u32x2 raw[EPT];
    auto issue_load = [&](int t, const bx_u32x4& r) {
      asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(raw[t]) : "v"(goff[t]), "s"(r) : "memory");
    };
    auto wait_load = [&](int t) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(raw[t]) : "n"(EPT - 1)); };
    auto convert_one = [&](int t, unsigned char* buf) {
      uint32_t p1, p2, p3;
      wait_load(t);
#ifdef BX_SLOWSPLIT
      bx_split2(__uint_as_float(raw[t].x), __uint_as_float(raw[t].y), p1, p2, p3);
#else
      bx_split2_fast(__uint_as_float(raw[t].x), __uint_as_float(raw[t].y), p1, p2, p3);
#endif
      unsigned char* d = buf + lbase + 32 * t;
      *reinterpret_cast<uint32_t*>(d) = p1;
      *reinterpret_cast<uint32_t*>(d + PLANE) = p2;
      *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = p3;
    };
    {
      const bx_u32x4 r0 = block_rsrc(rb), r1 = block_rsrc(rb + nwg);
#pragma unroll
      for (int t = 0; t < EPT; ++t) issue_load(t, r0);
#pragma unroll
      for (int t = 0; t < EPT; ++t) {          // element t's load is followed by EPT - 1 others each time it is waited for
        convert_one(t, lds);
        issue_load(t, r1);
      }
    }
#ifdef BX_PRIO
    __builtin_amdgcn_s_setprio(BX_PRIO);
#endif
    bx_barrier();
    int cur = 0;
    for (; rb < n_blocks; rb += nwg, ++n_iter) {
      unsigned char* bcur = lds + cur * BUF;
      unsigned char* bnxt = lds + (cur ^ 1) * BUF;
      const bx_u32x4 r2 = block_rsrc(rb + 2 * nwg);
      BX_MARK(0);
      pgt_f32x16 am[WN], ac[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[j][r] = 0.f; ac[j][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        BX_KMARK(i);
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + (KA + i) * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          am[j] = bx_mfma(fa[0], bf[i][j][0], am[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][0], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][2], ac[j]);
          ac[j] = bx_mfma(fa[2], bf[i][j][0], ac[j]);
        }
        // this k-step's share of the next block: fp32 (in registers since the previous iteration) -> bf16 planes in the
        // other buffer, and the load of the block after it into the freed registers
#pragma unroll
        for (int t = i * EPT / KB; t < (i + 1) * EPT / KB; ++t) {
          convert_one(t, bnxt);
          issue_load(t, r2);
        }
      }
      BX_MARK(3);
      while (part_seen[wc] != n_iter) { }     // the consumer is done with the previous block's sums (long ago)
      {
        float4* d = reinterpret_cast<float4*>(stage_part + wc * PART);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            d[(j * 4 + r4) * 64 + lane] = make_float4(am[j][4 * r4] + ac[j][4 * r4], am[j][4 * r4 + 1] + ac[j][4 * r4 + 1],
                                                      am[j][4 * r4 + 2] + ac[j][4 * r4 + 2], am[j][4 * r4 + 3] + ac[j][4 * r4 + 3]);
      }
      BX_MARK(4);
      bx_barrier();      // partial sums visible; everyone is done with this block's planes and the next block's are complete
      BX_MARK(5);
      BX_MARK(6);
      cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    float bias_r[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) { const int gn = (wc * WN + j) * 32 + (lane & 31); bias_r[j] = (g.bias && gn < g.N) ? g.bias[gn] : 0.f; }
    GemmArgs gnb = g; gnb.bias = nullptr;
    bx_barrier();
    int cur = 0;
    for (; rb < n_blocks; rb += nwg, ++n_iter) {
      unsigned char* bcur = lds + cur * BUF;
      BX_MARK(0);
      pgt_f32x16 am[WN], ac[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[j][r] = 0.f; ac[j][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < KA; ++i) {
        BX_KMARK(i);
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + i * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          am[j] = bx_mfma(fa[0], bf[i][j][0], am[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][0], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][2], ac[j]);
          ac[j] = bx_mfma(fa[2], bf[i][j][0], ac[j]);
        }
      }
      BX_MARK(3);
      pgt_f32x16 acc[1][WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = am[j][r] + ac[j][r] + bias_r[j];
      BX_MARK(4);
      bx_barrier();
      BX_MARK(5);
      {
        // the producer's partial sums join in registers; the block is stored straight from the accumulator layout: a
        // register is one 128-byte row piece per half-wavefront
        const float4* d = reinterpret_cast<const float4*>(stage_part + wc * PART);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = d[(j * 4 + r4) * 64 + lane];
            acc[0][j][4 * r4] += v.x; acc[0][j][4 * r4 + 1] += v.y; acc[0][j][4 * r4 + 2] += v.z; acc[0][j][4 * r4 + 3] += v.w;
          }
        if (lane == 0) part_seen[wc] = n_iter + 1;     // after the reads above: a wavefront's LDS operations complete in order
        const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int gn = (wc * WN + j) * 32 + lo;
          float* cp = g.C + gn;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int gm = rb * BM + (r & 3) + 8 * (r >> 2) + 4 * hi;
#ifndef BX_NOEPI
            if (gm < g.M && gn < g.N) cp[(int64_t)gm * g.ldc] = acc[0][j][r];
#else
            if (acc[0][j][r] == 123.456f) cp[0] = 1.f;
#endif
          }
        }
      }
      BX_MARK(6);
      cur ^= 1;
    }
  }
#ifdef BX_TL2
  if (g_bx_tl != nullptr && blockIdx.x == 0 && lane == 0) for (int i = 0; i < 24; ++i) g_bx_tl[wave * 24 + i] = tl[i];
#endif
  if (g_bx_clk != nullptr && tid == 0 && blockIdx.x < 4) {
    g_bx_clk[blockIdx.x * 3] = clock64() - c_begin; g_bx_clk[blockIdx.x * 3 + 1] = wall_clock64() - w_begin; g_bx_clk[blockIdx.x * 3 + 2] = n_iter;
  }
}

template <int CHAINS>
__global__ __launch_bounds__(512, 1) void bx_peak_kernel(float* out, int iters) {
  pgt_f32x16 acc[CHAINS];
  for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bx_u32x4 a = {threadIdx.x, 1, 2, 3}, b = {blockIdx.x, 5, 6, 7};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12 / CHAINS; ++u)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) acc[i] = bx_mfma(a, b, acc[i]);
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

// MFMA rate with the GEMM's operand feed: three ds_read_b128 per six MFMAs, PRE k-steps ahead
template <int PRE>
__global__ __launch_bounds__(512, 1) void bx_peak_lds_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 22016];
  for (int i = threadIdx.x; i < 3 * 22016 / 16; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = make_uint4(i, 1, 2, 3);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int arow = (lane & 31) * 688 + 16 * (lane >> 5);
  pgt_f32x16 am, ac;
  for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; }
  bx_u32x4 b0 = {blockIdx.x, 5, 6, 7}, b1 = {1, 2, 3, 4}, b2 = {9, 8, 7, 6};
  bx_u32x4 f[PRE + 1][3];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < PRE; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) f[p][q] = *reinterpret_cast<const bx_u32x4*>(lds + q * 22016 + arow + p * 32);
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) {
#pragma unroll
      for (int q = 0; q < 3; ++q) f[(ks + PRE) % (PRE + 1)][q] = *reinterpret_cast<const bx_u32x4*>(lds + q * 22016 + arow + ((ks + PRE) % 21) * 32);
      const bx_u32x4* fa = f[ks % (PRE + 1)];
      am = bx_mfma(fa[0], b0, am);
      ac = bx_mfma(fa[0], b1, ac);
      ac = bx_mfma(fa[1], b0, ac);
      ac = bx_mfma(fa[1], b1, ac);
      ac = bx_mfma(fa[0], b2, ac);
      ac = bx_mfma(fa[2], b0, ac);
    }
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += am[r] + ac[r];
  if (s == 123.456f) out[0] = s;
}

}  // namespace

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 211968, S = 5, C = 66, K = S * C, N = 128;
  const int NBUF = 3;
  float *A[NBUF], *W, *C0, *C1, *bias;
  for (int b = 0; b < NBUF; ++b) CK(hipMalloc(&A[b], (size_t)S * M * C * 4));
  CK(hipMalloc(&W, (size_t)K * N * 4)); CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
  CK(hipMalloc(&bias, N * 4));
  std::vector<float> hA((size_t)S * M * C), hW((size_t)K * N), hb(N);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hW) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  for (int b = 0; b < NBUF; ++b) CK(hipMemcpy(A[b], hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto fn) {
    for (int i = 0; i < 3; ++i) fn(i);
    CK(hipEventRecord(e0, st));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) fn(i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flop = 2.0 * M * K * N;
    printf("%-44s %8.1f us  %6.1f TF(fp32-equivalent)  %6.2f TB/s\n", name, us, flop / us * 1e-6,
           ((double)M * K * 4 + (double)M * N * 4) / us * 1e-6);
  };
  auto run_bx = [&](int i, float* out, int wgs) {
    GemmArgs g{A[i % NBUF], C, (int64_t)M * C, S, C, W, N, 1, out, N, 0, N, bias, M, N, 0, 0, 0, 0, 0, nullptr, 0, nullptr, 0,
               nullptr, nullptr, 0, nullptr, 0};
    const int n_blocks = (M + 31) / 32;
    hipLaunchKernelGGL((gemm_bx_kernel<21, BX_KA, 5, 66, 1>), dim3(std::min(wgs, n_blocks)), dim3(512), 0, st, g, n_blocks);
  };
  auto run_f32 = [&](int i, float* out) {
    if (pgt_gemm_f32(A[i % NBUF], C, (int64_t)M * C, S, C, W, N, 1, out, N, 0, N, bias, M, N, 0, st)) {
      printf("pgt_gemm_f32: %s\n", pgt_last_error()); exit(1);
    }
  };
  {
    auto peak = [&](const char* nm, auto kern, int threads) {
      const int iters = 2000;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, st, C0, iters);
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, st, C0, iters);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double n_mfma = 256.0 * (threads / 64) * iters * 12;
      printf("%-40s %7.1f us  %7.1f TF bf16  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", nm, ms * 1e3, n_mfma * 32768 / ms * 1e-9,
             ms * 1e-3 * 2.4e9 / (iters * 12.0 * (threads / 256)));
    };
    peak("bf16 MFMA, 1 wave/SIMD, 1 chain", bx_peak_kernel<1>, 256);
    peak("bf16 MFMA, 1 wave/SIMD, 2 chains", bx_peak_kernel<2>, 256);
    peak("bf16 MFMA, 1 wave/SIMD, 4 chains", bx_peak_kernel<4>, 256);
    peak("bf16 MFMA, 2 waves/SIMD, 1 chain", bx_peak_kernel<1>, 512);
    peak("bf16 MFMA, 2 waves/SIMD, 2 chains", bx_peak_kernel<2>, 512);
    auto peak2 = [&](const char* nm, auto kern, int threads) {
      const int iters = 200;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, st, C0, iters);
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, st, C0, iters);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double n_mfma = 256.0 * (threads / 64) * iters * 120;
      printf("%-44s %7.1f us  %7.1f TF bf16\n", nm, ms * 1e3, n_mfma * 32768 / ms * 1e-9);
    };
    peak2("MFMA + LDS feed, 1 wave/SIMD, no prefetch", bx_peak_lds_kernel<0>, 256);
    peak2("MFMA + LDS feed, 1 wave/SIMD, 1 step ahead", bx_peak_lds_kernel<1>, 256);
    peak2("MFMA + LDS feed, 1 wave/SIMD, 2 steps ahead", bx_peak_lds_kernel<2>, 256);
    peak2("MFMA + LDS feed, 2 waves/SIMD, no prefetch", bx_peak_lds_kernel<0>, 512);
    peak2("MFMA + LDS feed, 2 waves/SIMD, 1 step ahead", bx_peak_lds_kernel<1>, 512);
  }
  timeit("fp32 MFMA (pgt_gemm_f32)", [&](int i) { run_f32(i, C0); });
  timeit("split-bf16 x6, 256 workgroups", [&](int i) { run_bx(i, C1, 256); });
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  // accuracy of both against a double-precision product on sampled rows
  run_f32(0, C0); run_bx(0, C1, 256);
  CK(hipStreamSynchronize(st));
  std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
  CK(hipMemcpy(c0.data(), C0, c0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost));
  double e0max = 0, e1max = 0, e0sum = 0, e1sum = 0, dmax = 0; size_t cnt = 0;
  for (int m = 0; m < M; m += (M > 4000 ? 997 : 1)) {
    for (int n = 0; n < N; ++n) {
      double r = hb[n];
      for (int sgi = 0; sgi < S; ++sgi)
        for (int c = 0; c < C; ++c) r += (double)hA[((size_t)sgi * M + m) * C + c] * (double)hW[(size_t)(sgi * C + c) * N + n];
      const double d0 = fabs(c0[(size_t)m * N + n] - r), d1 = fabs(c1[(size_t)m * N + n] - r);
      e0max = std::max(e0max, d0); e1max = std::max(e1max, d1); e0sum += d0; e1sum += d1; ++cnt;
      dmax = std::max(dmax, (double)fabs(c0[(size_t)m * N + n] - c1[(size_t)m * N + n]));
    }
  }
  // also the last rows (ragged block) in full
  for (int m = std::max(0, M - 40); m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double r = hb[n];
      for (int sgi = 0; sgi < S; ++sgi)
        for (int c = 0; c < C; ++c) r += (double)hA[((size_t)sgi * M + m) * C + c] * (double)hW[(size_t)(sgi * C + c) * N + n];
      e1max = std::max(e1max, fabs(c1[(size_t)m * N + n] - r));
    }
  printf("error vs fp64 on %zu sampled outputs: fp32 MFMA max %.3e mean %.3e | split-bf16 max %.3e mean %.3e | max |diff| %.3e\n",
         cnt, e0max, e0sum / cnt, e1max, e1sum / cnt, dmax);
  {
    long long* dclk; CK(hipMalloc(&dclk, 12 * 8)); CK(hipMemset(dclk, 0, 96));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_clk), &dclk, sizeof(dclk)));
    run_bx(0, C1, 256); CK(hipStreamSynchronize(st));
    long long h[12]; CK(hipMemcpy(h, dclk, 96, hipMemcpyDeviceToHost));
    for (int w = 0; w < 2; ++w)
      printf("workgroup %d: %lld iterations, %.0f shader cycles and %.3f us per iteration (%.2f GHz)\n", w, h[w * 3 + 2],
             (double)h[w * 3] / h[w * 3 + 2], h[w * 3 + 1] / 100.0 / h[w * 3 + 2], h[w * 3] / (h[w * 3 + 1] * 10.0));
    dclk = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_clk), &dclk, sizeof(dclk)));
#ifdef BX_TL2
    {
      long long* dtl; CK(hipMalloc(&dtl, 8 * 24 * 8)); CK(hipMemset(dtl, 0, 8 * 24 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_tl), &dtl, sizeof(dtl)));
      run_bx(0, C1, 256); CK(hipStreamSynchronize(st));
      long long tl[8 * 24]; CK(hipMemcpy(tl, dtl, sizeof tl, hipMemcpyDeviceToHost));
      for (int w : {0, 4, 1, 5}) {
        const long long t0 = tl[0 * 24 + 0];
        printf("wave %d: marks", w);
        for (int k = 0; k < 7; ++k) printf(" %lld", tl[w * 24 + k] ? tl[w * 24 + k] - t0 : -1);
        printf(" | k-steps");
        for (int k = 0; k < 13; ++k) if (tl[w * 24 + 8 + k]) printf(" %lld", tl[w * 24 + 8 + k] - t0);
        printf("\n");
      }
      dtl = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_tl), &dtl, sizeof(dtl)));
    }
#endif
    long long* dtl; CK(hipMalloc(&dtl, 64 * 8)); CK(hipMemset(dtl, 0, 512));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_tl), &dtl, sizeof(dtl)));
    run_bx(0, C1, 256); CK(hipStreamSynchronize(st));
    long long tl[64]; CK(hipMemcpy(tl, dtl, 512, hipMemcpyDeviceToHost));
    for (int w = 0; w < 8; ++w) {
      printf("wave %d:", w);
      for (int k = 1; k < 7; ++k) printf(" +%lld", tl[w * 8 + k] - tl[w * 8 + k - 1]);
      printf("   (start %+lld vs wave 0)\n", tl[w * 8] - tl[0]);
    }
    dtl = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_tl), &dtl, sizeof(dtl)));
  }
  for (int wgs : {128, 192, 256}) {
    char nm[64]; snprintf(nm, sizeof nm, "split-bf16 x6, %d workgroups", wgs);
    timeit(nm, [&](int i) { run_bx(i, C1, wgs); });
  }
  return 0;
}
