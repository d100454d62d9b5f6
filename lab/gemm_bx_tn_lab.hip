// LAB HARNESS (not shipped): split-bf16 weight-gradient GEMM  dW[k, n] += sum_m A[m, k] G[m, n],  db[n] += sum_m G[m, n].
// Both operands stream; both are cut into three bf16 planes and stored TRANSPOSED in LDS ([column][row], 48-byte rows),
// so that the MFMA operand of a lane (eight consecutive rows of one column) is one ds_read_b128.
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

#ifndef FLUSH
#define FLUSH 160
#endif
namespace {

typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t bx_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bx_pack(float x, float y) {
  bx_f32x2 v = {x, y};
  bx_bf16x2 r = __builtin_convertvector(v, bx_bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
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
__device__ __forceinline__ void bx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BxTnArgs {
  const float* A; int64_t lda; int64_t a_seg_stride; int n_seg; int seg_k;
  const float* G; int64_t ldg; float* dW; int64_t lddw; float* db; int M; int N;
};

// NCB: 32-column blocks of G (4: N <= 128, 2: N <= 64).  K + 1 <= 352 rows of dW (row K = the bias gradient: A^T gets a
// row of ones there).  One persistent 512-thread workgroup per CU; a stage = 16 rows of A and G.
template <int NCB>
__global__ __launch_bounds__(512, 1) void gemm_bx_tn_kernel(BxTnArgs g, int n_stages) {
  constexpr int RB = 11, ROWB = 48, APL = RB * 32 * ROWB, GPL = NCB * 32 * ROWB, BUF = 3 * (APL + GPL);
  constexpr int RSTEP = 8 / NCB, MAXB = (RB + RSTEP - 1) / RSTEP;
  constexpr int EPT_A = 6, EPT_G = NCB * 32 * 8 / 512, EPT = EPT_A + EPT_G;     // (column, row pair) units per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave % NCB, r0 = wave / NCB;
  const int nwg = gridDim.x;
  const int K = g.n_seg * g.seg_k;
  // ---- LDS: zeros, then the row of ones at column K of A^T (first plane; 1.0 = 0x3f80)
  for (int i = tid; i < 2 * BUF / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 16) reinterpret_cast<uint32_t*>(lds + (tid >> 3) * BUF + K * ROWB)[tid & 7] = 0x3f803f80u;
  // ---- unit map: unit = (column, row pair); lanes along the columns
  uint32_t goff[EPT];   // byte offset of the unit's first row from the stage base (A or G)
  uint32_t loff[EPT];   // byte offset of the unit's dword inside a buffer, first plane
#pragma unroll
  for (int t = 0; t < EPT_A; ++t) {
    const int u = tid + 512 * t;
    if (u < K * 8) {
      const int rp = u / K, c = u - rp * K, seg = c / g.seg_k, cc = c - seg * g.seg_k;
      goff[t] = (uint32_t)((seg * g.a_seg_stride + 2 * rp * g.lda + cc) * 4);
      loff[t] = (uint32_t)(c * ROWB + rp * 4);
    } else {
      goff[t] = 0xfffffff0u;                      // outside the descriptor: reads zero ...
      loff[t] = (uint32_t)((RB * 32 - 1) * ROWB + 32);   // ... and lands in the padding of the last row
    }
  }
#pragma unroll
  for (int t = 0; t < EPT_G; ++t) {
    const int u = tid + 512 * t, rp = u / (NCB * 32), c = u - rp * (NCB * 32);
    goff[EPT_A + t] = c < g.N ? (uint32_t)((2 * rp * g.ldg + c) * 4) : 0xfffffff0u;
    loff[EPT_A + t] = (uint32_t)(3 * APL + c * ROWB + rp * 4);
  }
  const uint32_t lda4 = (uint32_t)(g.lda * 4), ldg4 = (uint32_t)(g.ldg * 4);
  auto rsrc = [&](const float* p, int64_t bytes) {
    const uint64_t base = reinterpret_cast<uint64_t>(p);
    bx_u32x4 r = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) & 0xffffu,
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > 0 ? bytes : 0)), 0x00020000u};
    return r;
  };
  const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
  auto a_rsrc = [&](int st) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)st * 16;
    const int64_t rows = rows_left < 16 ? rows_left : 16;
    return rsrc(g.A + (int64_t)(rows_left > 0 ? st : 0) * 16 * g.lda, rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0);
  };
  auto g_rsrc = [&](int st) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)st * 16;
    const int64_t rows = rows_left < 16 ? rows_left : 16;
    return rsrc(g.G + (int64_t)(rows_left > 0 ? st : 0) * 16 * g.ldg, rows_left > 0 ? (rows - 1) * g.ldg * 4 + (int64_t)g.N * 4 : 0);
  };
  // hand-issued loads, hand-counted waits: every unit is two loads (its two rows), consumed in issue order and reissued
  // right after its conversion: 2 (EPT - 1) younger loads are in flight when a unit is due
  float raw0[EPT], raw1[EPT];
  auto issue = [&](int t, const bx_u32x4& ra, const bx_u32x4& rg) {
    if (t < EPT_A) {
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(raw0[t]) : "v"(goff[t]), "s"(ra) : "memory");
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(raw1[t]) : "v"(goff[t]), "s"(ra), "s"(lda4) : "memory");
    } else {
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(raw0[t]) : "v"(goff[t]), "s"(rg) : "memory");
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(raw1[t]) : "v"(goff[t]), "s"(rg), "s"(ldg4) : "memory");
    }
  };
  // rows_left: valid rows of the stage being converted (A rows past M inside the earlier segments are other data, not
  // zeros: masked here; G rows past M read zero through the descriptor anyway)
  auto convert = [&](int t, unsigned char* buf, int rows_left) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(raw0[t]), "+v"(raw1[t]) : "n"(2 * (EPT - 1)));
    float x = raw0[t], y = raw1[t];
    if (rows_left < 16 && t < EPT_A) {
      const int m0 = 2 * (int)((loff[t] % ROWB) >> 2);
      x = m0 < rows_left ? x : 0.f;
      y = m0 + 1 < rows_left ? y : 0.f;
    }
    uint32_t p1, p2, p3;
    bx_split2_fast(x, y, p1, p2, p3);
    constexpr int PL_A = APL, PL_G = GPL;
    unsigned char* d = buf + loff[t];
    const int pl = t < EPT_A ? PL_A : PL_G;
    *reinterpret_cast<uint32_t*>(d) = p1;
    *reinterpret_cast<uint32_t*>(d + pl) = p2;
    *reinterpret_cast<uint32_t*>(d + 2 * pl) = p3;
  };
  int st = blockIdx.x;
  pgt_f32x16 acc[MAXB];
  const int lo = lane & 31, hi = lane >> 5;
  const int n = cb * 32 + lo;
  auto flush = [&]() {
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      const int rb = r0 + RSTEP * b;
      if (rb >= RB || n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (k < K) atomicAdd(g.dW + (int64_t)k * g.lddw + n, acc[b][r]);
        else if (k == K && g.db != nullptr) atomicAdd(g.db + n, acc[b][r]);
        acc[b][r] = 0.f;
      }
    }
  };
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  __syncthreads();
  if (st < n_stages) {
    {
      const bx_u32x4 ra0 = a_rsrc(st), rg0 = g_rsrc(st), ra1 = a_rsrc(st + nwg), rg1 = g_rsrc(st + nwg);
#pragma unroll
      for (int t = 0; t < EPT; ++t) issue(t, ra0, rg0);
#pragma unroll
      for (int t = 0; t < EPT; ++t) {
        convert(t, lds, g.M - st * 16);
        issue(t, ra1, rg1);
      }
    }
    bx_barrier();
    int cur = 0, since_flush = 0;
    const int afrag = (lane & 31) * ROWB + 16 * (lane >> 5);
    for (; st < n_stages; st += nwg) {
      unsigned char* bcur = lds + cur * BUF;
      unsigned char* bnxt = lds + (cur ^ 1) * BUF;
      const bx_u32x4 ra2 = a_rsrc(st + 2 * nwg), rg2 = g_rsrc(st + 2 * nwg);
      const int rows_next = g.M - (st + nwg) * 16;
      bx_u32x4 fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bx_u32x4*>(bcur + 3 * APL + q * GPL + cb * 32 * ROWB + afrag);
      // the six piece products of a stage meet in a zeroed accumulator and join the running sum with ONE rounded add per
      // stage (the running sums grow to ~1e3 over a workgroup's 10 000 rows: six roundings per stage against them cost
      // 3x the error of the fp32 kernel); the add of block b rides behind the MFMAs of block b + 1
#pragma unroll
      for (int b = 0; b < MAXB; ++b) {
        const int rb = r0 + RSTEP * b;
        if (rb < RB) {
          bx_u32x4 fa[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * APL + rb * 32 * ROWB + afrag);
          acc[b] = bx_mfma(fa[2], fb[0], acc[b]);
          acc[b] = bx_mfma(fa[0], fb[2], acc[b]);
          acc[b] = bx_mfma(fa[1], fb[1], acc[b]);
          acc[b] = bx_mfma(fa[1], fb[0], acc[b]);
          acc[b] = bx_mfma(fa[0], fb[1], acc[b]);
          acc[b] = bx_mfma(fa[0], fb[0], acc[b]);
        }
#pragma unroll
        for (int t = b * EPT / MAXB; t < (b + 1) * EPT / MAXB; ++t) {
          convert(t, bnxt, rows_next);
          issue(t, ra2, rg2);
        }
      }
      // the running sums leave for dW every FLUSH stages: the rounding error of a sum grows with its length and size
      if (++since_flush == FLUSH) { flush(); since_flush = 0; }
      bx_barrier();
      cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  flush();     // ---- the rest of the workgroup's sums join dW / db
}

}  // namespace

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2543616, N = argc > 2 ? atoi(argv[2]) : 128, S = 5, C = 66, K = S * C;
  float *A, *G, *dW0, *dW1, *db0, *db1;
  CK(hipMalloc(&A, (size_t)S * M * C * 4)); CK(hipMalloc(&G, (size_t)M * N * 4));
  CK(hipMalloc(&dW0, (size_t)K * N * 4)); CK(hipMalloc(&dW1, (size_t)K * N * 4)); CK(hipMalloc(&db0, N * 4)); CK(hipMalloc(&db1, N * 4));
  std::vector<float> hA((size_t)S * M * C), hG((size_t)M * N);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hG) v = rnd();
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_bx = [&](float* dW, float* db) {
    BxTnArgs g{A, C, (int64_t)M * C, S, C, G, N, dW, N, db, M, N};
    const int n_stages = (M + 15) / 16;
    if (N > 64) hipLaunchKernelGGL((gemm_bx_tn_kernel<4>), dim3(std::min(256, n_stages)), dim3(512), 0, st, g, n_stages);
    else hipLaunchKernelGGL((gemm_bx_tn_kernel<2>), dim3(std::min(256, n_stages)), dim3(512), 0, st, g, n_stages);
  };
  auto run_f32 = [&](float* dW, float* db) {
    if (pgt_gemm_tn_acc_f32(A, C, (int64_t)M * C, S, C, G, N, dW, N, db, M, N, st)) { printf("tn: %s\n", pgt_last_error()); exit(1); }
  };
  auto timeit = [&](const char* name, auto fn) {
    for (int i = 0; i < 2; ++i) fn();
    CK(hipEventRecord(e0, st));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flop = 2.0 * M * K * N;
    printf("%-40s %9.1f us  %6.1f TF(fp32-equivalent)  %5.2f TB/s\n", name, us, flop / us * 1e-6,
           ((double)M * K * 4 + (double)M * N * 4) / us * 1e-6);
  };
  timeit("fp32 MFMA (pgt_gemm_tn_acc_f32)", [&]() { run_f32(dW0, db0); });
  timeit("split-bf16 x6", [&]() { run_bx(dW1, db1); });
  CK(hipGetLastError());
  // correctness: one accumulation each into zeroed outputs
  CK(hipMemsetAsync(dW0, 0, (size_t)K * N * 4, st)); CK(hipMemsetAsync(dW1, 0, (size_t)K * N * 4, st));
  CK(hipMemsetAsync(db0, 0, N * 4, st)); CK(hipMemsetAsync(db1, 0, N * 4, st));
  run_f32(dW0, db0); run_bx(dW1, db1);
  CK(hipStreamSynchronize(st));
  std::vector<float> w0((size_t)K * N), w1((size_t)K * N), b0(N), b1(N);
  CK(hipMemcpy(w0.data(), dW0, w0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(w1.data(), dW1, w1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b0.data(), db0, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b1.data(), db1, N * 4, hipMemcpyDeviceToHost));
  // fp64 reference on a sample of (k, n) entries
  double e0m = 0, e1m = 0, scale = 0;
  const int ks[] = {0, 1, 65, 66, 131, 200, 264, 329}, ns[] = {0, 1, 31, 32, 63, N - 1};
  for (int k : ks) for (int n : ns) {
    const int sg = k / C, c = k % C;
    double r = 0;
    for (int m = 0; m < M; ++m) r += (double)hA[((size_t)sg * M + m) * C + c] * (double)hG[(size_t)m * N + n];
    e0m = std::max(e0m, fabs(w0[(size_t)k * N + n] - r)); e1m = std::max(e1m, fabs(w1[(size_t)k * N + n] - r)); scale = std::max(scale, fabs(r));
  }
  double bm0 = 0, bm1 = 0;
  for (int n : ns) { double r = 0; for (int m = 0; m < M; ++m) r += hG[(size_t)m * N + n]; bm0 = std::max(bm0, fabs(b0[n] - r)); bm1 = std::max(bm1, fabs(b1[n] - r)); }
  double dmax = 0; for (size_t i = 0; i < w0.size(); ++i) dmax = std::max(dmax, (double)fabs(w0[i] - w1[i]));
  printf("max |error| vs fp64 on 48 entries (scale %.1f): fp32 MFMA %.3e, split-bf16 %.3e; db: %.3e / %.3e; max |dW diff| over all entries %.3e\n",
         scale, e0m, e1m, bm0, bm1, dmax);
  return 0;
}
