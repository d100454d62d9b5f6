// Fused T-GCN cell (temporalgcn.py:82-130) for hidden width 32 — BASELINE configs[2] / [3]: A3TGCN2(2, 32), TGCN2(2, 32).
//
// With the aggregation AX = A_hat X taken once at the input width and conv_g -> linear_g folded into one operand per gate pair
// (csrc/tgcn.hip), everything else in the cell is ROW-LOCAL: Z | R = sigmoid([AX | H] Wzr + bzr), the candidate
// tanh([AX | H * R] Wh + bh), the blend.  Round 4's first form ran that as two fused-epilogue products plus movers (read / write
// ~1.4 KB per row forward, eight launches backward); here a row makes ONE trip through a CU each way:
//   tgcn_cell_fwd_kernel   a wavefront owns 32 rows: H is fetched in the accumulator layout (lane = column, 16 rows per lane), parked
//       k-major in the wavefront's own LDS strip as the A operand, Z | R on two v_mfma_f32_32x32x2_f32 accumulators, sigmoid, H * R
//       written over H in the strip (the lane that parked an element owns it), the candidate on a third accumulator, tanh and
//       the blend against the H values still in registers.  Reads 4 (Fin + 32) bytes per row, writes Z | R, the candidate (for
//       the adjoint) and H': 648 B per row at Fin = 2.  No workgroup barrier: the four wavefronts never touch each other's rows.
//   tgcn_cell_bwd_kernel   the whole adjoint: gate chains on registers, d(H R) = d_pre_h Wh^T and dH += d_pre_zr Wzr^T as two
//       products against the weights' hidden rows, and BOTH weight gradients as row-contracting products ([AX | 1 | H]^T d_pre:
//       the row of ones makes the bias gradients fall out) on six accumulators that live across the tiles of a persistent
//       workgroup; per-workgroup partial sums go to a scratch buffer and tgcn_cell_reduce_kernel adds them in index order
//       (deterministic, no float atomics).  776 B per row; replaces eight launches that moved ~3.3 KB per row.
//   d/dX (the transposed aggregation's operand) is not produced here: a caller that needs it runs the unfused adjoint.
#include "pgt_common.h"

#include <initializer_list>
#include <type_traits>

namespace {

int g_tc_rows = 1;                  // pgt_tune("tgcn_rows", 0): the round-4 column-per-lane kernels for every shape
int g_tc_wgs = 0;                   // pgt_tune("tgcn_wgs", n): workgroups of the row-per-lane forward kernel (0 = default), lab use
int g_tc_probe = 0;                 // lab variants of the forward kernel (wrong results; -DPGT_LAB_PROBES builds only), see PROBE

constexpr int TC_O = 32;            // hidden width these kernels are built for
constexpr int TC_LD = 33;           // LDS row pitch of a wavefront's strips (32 rows + 1: conflict-free both ways)

struct TcArgs {
  const float* AX; int64_t ldax; const float* H; int64_t ldh;
  const float* Wzr; const float* bzr; const float* Wh; const float* bh;     // [Fin + 32, 64], [64] | null, [Fin + 32, 32], [32] | null
  float* ZR; float* HT; float* Hn; int64_t ldhn;
  int M, Fin, tiles;
  // adjoint
  const float* dHn; int64_t lddhn; float* dH; int64_t lddh; float* part; int n_wg;
};

// sigmoid / tanh on the hardware exp / reciprocal (<= 3e-7 from the library functions; tanh by its odd series below |x| = 0.04,
// where 1 - 2 / (1 + e^2x) loses relative accuracy) — the same forms as the fused gate epilogues of csrc/gemm_bx.hip.  The library
// expf + IEEE division + tanhf are ~80 VALU instructions per element: 48 elements per lane and strip cost more issue cycles than
// the strip's 51 MFMAs.
#ifdef PGT_EMU
__device__ __forceinline__ float tc_rcp(float x) { return 1.f / x; }
__device__ __forceinline__ float tc_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float tc_rcp(float x) { return __frcp_rn(x); }
__device__ __forceinline__ float tc_exp(float x) { return __expf(x); }
#endif
__device__ __forceinline__ float tc_sigmoidf(float x) { return tc_rcp(1.f + tc_exp(-x)); }
__device__ __forceinline__ float tc_tanhf(float x) {
  const float x2 = x * x;
  const float small = x * fmaf(x2, fmaf(x2, 0.13333334f, -0.33333334f), 1.f);
  const float big = 1.f - 2.f * tc_rcp(1.f + tc_exp(2.f * x));
  return fabsf(x) < 0.04f ? small : big;
}

// accumulator (D) layout of v_mfma_f32_32x32x2_f32: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
__device__ __forceinline__ int tc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// KF: the padded contraction length Fin + 32 (+ 1) when known at compile time (34: one or two input columns — TGCN2(2, 32) — so the
// product loops unroll completely and the LDS operand reads run ahead of the MFMAs instead of one read-wait-MFMA round per
// k-step), 0 = run time
template <int KF>
__global__ __launch_bounds__(256) void tgcn_cell_fwd_kernel(TcArgs g) {
  // per wavefront: A strip [K2][33]; shared: Wzr [K2][65], Wh [K2][33]
  __shared__ float s_w[64 * 65 + 64 * 33];
  __shared__ float s_a[4][64 * TC_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int C = g.Fin + TC_O, K2 = KF > 0 ? KF : ((C + 1) & ~1);
  float* wzr = s_w;                   // [K2][65]
  float* wh = s_w + 64 * 65;          // [K2][33]
  for (int e = tid; e < K2 * 64; e += 256) { const int k = e >> 6, j = e & 63; wzr[k * 65 + j] = k < C ? g.Wzr[(int64_t)k * 64 + j] : 0.f; }
  for (int e = tid; e < K2 * 32; e += 256) { const int k = e >> 5, j = e & 31; wh[k * 33 + j] = k < C ? g.Wh[(int64_t)k * 32 + j] : 0.f; }
  float* As = s_a[wave];
  if (K2 > C) for (int i = lane; i < TC_LD; i += 64) As[C * TC_LD + i] = 0.f;      // the padding k-row stays zero
  __syncthreads();
  const float b_z = g.bzr ? g.bzr[lo] : 0.f, b_r = g.bzr ? g.bzr[32 + lo] : 0.f, b_h = g.bh ? g.bh[lo] : 0.f;
  // KF > 0: the B operands of this lane (its column of the three weight blocks, k = kk + hi) live in REGISTERS for the whole launch
  constexpr int KS = KF > 0 ? KF / 2 : 1;
  float wbz[KS], wbr[KS], wbh[KS];
  if constexpr (KF > 0) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      wbz[i] = wzr[(2 * i + hi) * 65 + lo];
      wbr[i] = wzr[(2 * i + hi) * 65 + 32 + lo];
      wbh[i] = wh[(2 * i + hi) * 33 + lo];
    }
  }
  const float* __restrict__ Hg = g.H;
  const float* __restrict__ AXg = g.AX;
  float* __restrict__ ZRg = g.ZR;
  float* __restrict__ HTg = g.HT;
  float* __restrict__ Hng = g.Hn;
  // the strip of the NEXT tile is fetched while this one is on the matrix cores (two register sets): a wavefront's phases would
  // otherwise run load -> products -> stores strictly in turn, with two wavefronts per SIMD to hide all of it
  const int axc = lo < g.Fin ? lo : 0;
  float hn[16], axn[16];
  auto fetch = [&](int t) {
    const int64_t mt = (int64_t)t * 128 + wave * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t mr = mt + tc_row(r, hi);
      const int64_t m = mr < g.M ? mr : g.M - 1;
      hn[r] = Hg[m * g.ldh + lo];
      axn[r] = AXg[m * g.ldax + axc];
    }
  };
  if ((int)blockIdx.x < g.tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int64_t m0 = (int64_t)tile * 128 + wave * 32;
    float h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      h[r] = hn[r];
      As[(g.Fin + lo) * TC_LD + row] = h[r];
      if (lo < g.Fin) As[lo * TC_LD + row] = axn[r];
    }
    if (tile + (int)gridDim.x < g.tiles) fetch(tile + gridDim.x);
    PGT_WAVE_SYNC();
    pgt_f32x16 az, ar, ah;
#pragma unroll
    for (int r = 0; r < 16; ++r) { az[r] = 0.f; ar[r] = 0.f; ah[r] = 0.f; }
    if constexpr (KF > 0) {
      float aop[KS];                                       // every A operand of the product is requested before the first MFMA
#pragma unroll
      for (int i = 0; i < KS; ++i) aop[i] = As[(2 * i + hi) * TC_LD + lo];
      PGT_SCHED_FENCE();                                   // (the scheduler otherwise sinks every read back next to its MFMA)
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        az = PGT_MFMA_32x32x2(aop[i], wbz[i], az);
        ar = PGT_MFMA_32x32x2(aop[i], wbr[i], ar);
      }
    } else {
      for (int kk = 0; kk < K2; kk += 2) {
        const float a = As[(kk + hi) * TC_LD + lo];
        az = PGT_MFMA_32x32x2(a, wzr[(kk + hi) * 65 + lo], az);
        ar = PGT_MFMA_32x32x2(a, wzr[(kk + hi) * 65 + 32 + lo], ar);
      }
    }
    PGT_WAVE_SYNC();
    float z[16];
    const bool full = m0 + 32 <= g.M;                                      // wavefront-uniform: whole strips store without per-row branches
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      z[r] = tc_sigmoidf(az[r] + b_z);
      const float rr = tc_sigmoidf(ar[r] + b_r);
      As[(g.Fin + lo) * TC_LD + row] = h[r] * rr;                       // [AX | H * R]: this lane parked (row, lo) itself
      if (full || m0 + row < g.M) {
        float* zr = ZRg + (m0 + row) * 64;
        zr[lo] = z[r];
        zr[32 + lo] = rr;
      }
    }
    PGT_WAVE_SYNC();
    if constexpr (KF > 0) {
      float aop[KS];
#pragma unroll
      for (int i = 0; i < KS; ++i) aop[i] = As[(2 * i + hi) * TC_LD + lo];
      PGT_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < KS; ++i) ah = PGT_MFMA_32x32x2(aop[i], wbh[i], ah);
    } else {
      for (int kk = 0; kk < K2; kk += 2) ah = PGT_MFMA_32x32x2(As[(kk + hi) * TC_LD + lo], wh[(kk + hi) * 33 + lo], ah);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      if (full || m0 + row < g.M) {
        const float ht = tc_tanhf(ah[r] + b_h);
        HTg[(m0 + row) * 32 + lo] = ht;
        Hng[(m0 + row) * g.ldhn + lo] = pgt_gru_blend(z[r], h[r], ht);
      }
    }
    PGT_WAVE_SYNC();                                                       // the strip is rewritten by the next tile
  }
}

// partial layout per workgroup (floats): dWzr [C][64] | dbzr [64] | dWh [C][32] | dbh [32]
__device__ __forceinline__ int tc_part_floats(int Fin) { return (Fin + TC_O) * 96 + 96; }

__global__ __launch_bounds__(256) void tgcn_cell_bwd_kernel(TcArgs g) {
  // shared: the hidden rows of the weights, transposed for "d_pre x W^T": whT [k = o][n = i] = Wh[Fin + i][o]  (32 x 33),
  //         wzrT [k = j][n = i] = Wzr[Fin + i][j] (64 x 33).  per wavefront: Dz [64][33] (d_pre_z | d_pre_r), Dh [32][33],
  //         Xs [64][33]: rows 0 .. Fin - 1 = AX columns, row Fin = ones, Fin + 1 .. 31 = zero, rows 32 .. 63 = H (then H * R)
  __shared__ float s_wt[32 * 33 + 64 * 33];
  __shared__ float s_t[4][(64 + 32 + 64) * TC_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int Fin = g.Fin, C = Fin + TC_O;
  float* whT = s_wt;
  float* wzrT = s_wt + 32 * 33;
  for (int e = tid; e < 32 * 32; e += 256) { const int o = e >> 5, i = e & 31; whT[o * 33 + i] = g.Wh[(int64_t)(Fin + i) * 32 + o]; }
  for (int e = tid; e < 64 * 32; e += 256) { const int j = e >> 5, i = e & 31; wzrT[j * 33 + i] = g.Wzr[(int64_t)(Fin + i) * 64 + j]; }
  float* Dz = s_t[wave];
  float* Dh = Dz + 64 * TC_LD;
  float* Xs = Dh + 32 * TC_LD;
  for (int e = lane; e < 32 * TC_LD; e += 64) Xs[e] = (e / TC_LD == Fin) ? 1.f : 0.f;   // ones row, zero padding (AX rows are rewritten per tile)
  __syncthreads();
  // this lane's B operands of the two "d_pre x W^T" products, in registers for the whole launch
  float wbh[16], wbz[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) wbh[i] = whT[(2 * i + hi) * 33 + lo];
#pragma unroll
  for (int i = 0; i < 32; ++i) wbz[i] = wzrT[(2 * i + hi) * 33 + lo];
  const float* __restrict__ dHg = g.dHn;
  const float* __restrict__ ZRg = g.ZR;
  const float* __restrict__ HTg = g.HT;
  const float* __restrict__ Hg = g.H;
  const float* __restrict__ AXg = g.AX;
  pgt_f32x16 wz0a, wz0b, wz1a, wz1b, wh0, wh1;         // [AX | 1]^T dzr (two column blocks), H^T dzr, [AX | 1]^T dph, (H R)^T dph
#pragma unroll
  for (int r = 0; r < 16; ++r) { wz0a[r] = wz0b[r] = wz1a[r] = wz1b[r] = wh0[r] = wh1[r] = 0.f; }
  const int axc = lo < Fin ? lo : 0;
  float ggn[16], zzn[16], rrn[16], htn[16], hn[16], axn[16];       // the next tile's operands (see the forward kernel)
  auto fetch = [&](int t) {
    const int64_t mt = (int64_t)t * 128 + wave * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t mr = mt + tc_row(r, hi);
      const int64_t m = mr < g.M ? mr : g.M - 1;
      ggn[r] = dHg[m * g.lddhn + lo];
      zzn[r] = ZRg[m * 64 + lo];
      rrn[r] = ZRg[m * 64 + 32 + lo];
      htn[r] = HTg[m * 32 + lo];
      hn[r] = Hg[m * g.ldh + lo];
      axn[r] = AXg[m * g.ldax + axc];
    }
  };
  if ((int)blockIdx.x < g.tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int64_t m0 = (int64_t)tile * 128 + wave * 32;
    float gz[16], rr[16], h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      const bool ok = m0 + row < g.M;
      const float g1 = ok ? ggn[r] : 0.f;                                  // rows past the end contribute nothing
      const float z1 = zzn[r], t1 = htn[r];
      rr[r] = rrn[r];
      h[r] = hn[r];
      Dh[lo * TC_LD + row] = g1 * (1.f - z1) * (1.f - t1 * t1);              // d_pre_h
      Dz[lo * TC_LD + row] = g1 * (h[r] - t1) * z1 * (1.f - z1);             // d_pre_z
      gz[r] = g1 * z1;
      Xs[(32 + lo) * TC_LD + row] = ok ? h[r] : 0.f;
      if (lo < Fin) Xs[lo * TC_LD + row] = ok ? axn[r] : 0.f;
      if (lo == Fin) Xs[Fin * TC_LD + row] = ok ? 1.f : 0.f;
    }
    if (tile + (int)gridDim.x < g.tiles) fetch(tile + gridDim.x);
    PGT_WAVE_SYNC();
    pgt_f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    {
      float aop[16];                                       // operands requested before the first MFMA (no read-wait-MFMA rounds)
#pragma unroll
      for (int i = 0; i < 16; ++i) aop[i] = Dh[(2 * i + hi) * TC_LD + lo];
      PGT_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 16; ++i) p = PGT_MFMA_32x32x2(aop[i], wbh[i], p);                          // d(H R)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      Dz[(32 + lo) * TC_LD + row] = (m0 + row < g.M) ? p[r] * h[r] * rr[r] * (1.f - rr[r]) : 0.f;   // d_pre_r
      gz[r] = fmaf(p[r], rr[r], gz[r]);
    }
    PGT_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    {
      float aop[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) aop[i] = Dz[(2 * i + hi) * TC_LD + lo];
      PGT_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 32; ++i) p = PGT_MFMA_32x32x2(aop[i], wbz[i], p);                          // d_pre_zr Wzr_H^T
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tc_row(r, hi);
      if (m0 + row < g.M) g.dH[(m0 + row) * g.lddh + lo] = gz[r] + p[r];
    }
    // weight gradients: D[i][j] += sum_m Xs[i][m] d[m][j]  (A operand = Xs rows, k = the tile's rows; B operand = d, stored [j][m])
#pragma unroll
    for (int half = 0; half < 2; ++half) {                 // eight k-steps' operands at a time ahead of their 32 MFMAs
      float a0[8], a1[8], bz[8], br[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = 16 * half + 2 * i + hi;
        a0[i] = Xs[lo * TC_LD + kk]; a1[i] = Xs[(32 + lo) * TC_LD + kk];
        bz[i] = Dz[lo * TC_LD + kk]; br[i] = Dz[(32 + lo) * TC_LD + kk];
      }
      PGT_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        wz0a = PGT_MFMA_32x32x2(a0[i], bz[i], wz0a);
        wz0b = PGT_MFMA_32x32x2(a0[i], br[i], wz0b);
        wz1a = PGT_MFMA_32x32x2(a1[i], bz[i], wz1a);
        wz1b = PGT_MFMA_32x32x2(a1[i], br[i], wz1b);
      }
    }
    PGT_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; ++r) Xs[(32 + lo) * TC_LD + tc_row(r, hi)] *= rr[r];       // H -> H * R (rows past the end are zero)
    PGT_WAVE_SYNC();
    {
      float a0[16], a1[16], b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = 2 * i + hi;
        a0[i] = Xs[lo * TC_LD + kk]; a1[i] = Xs[(32 + lo) * TC_LD + kk]; b[i] = Dh[lo * TC_LD + kk];
      }
      PGT_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        wh0 = PGT_MFMA_32x32x2(a0[i], b[i], wh0);
        wh1 = PGT_MFMA_32x32x2(a1[i], b[i], wh1);
      }
    }
    PGT_WAVE_SYNC();
  }
  // ---- the four wavefronts' sums meet in LDS (the strips are dead), one partial per workgroup
  __syncthreads();
  // six accumulators x 1024 floats per wavefront would be 24 KB each; the (dead) strips hold 4 x 21 KB: three accumulators at a time
  float* red = &s_t[0][0];
  float* part = g.part + (int64_t)blockIdx.x * tc_part_floats(Fin);
  float* pWzr = part;
  float* pbzr = part + (int64_t)C * 64;
  float* pWh = pbzr + 64;
  float* pbh = pWh + (int64_t)C * 32;
  for (int round = 0; round < 2; ++round) {
    float* mine = red + wave * 3 * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = tc_row(r, hi);
      if (round == 0) { mine[i * 32 + lo] = wz0a[r]; mine[1024 + i * 32 + lo] = wz0b[r]; mine[2048 + i * 32 + lo] = wh0[r]; }
      else { mine[i * 32 + lo] = wz1a[r]; mine[1024 + i * 32 + lo] = wz1b[r]; mine[2048 + i * 32 + lo] = wh1[r]; }
    }
    __syncthreads();
    for (int e = tid; e < 3 * 1024; e += 256) {
      const float v = ((red[e] + red[3 * 1024 + e]) + red[6 * 1024 + e]) + red[9 * 1024 + e];
      const int blk = e >> 10, i = (e >> 5) & 31, j = e & 31;
      if (round == 0) {                    // rows of [AX | 1]: i < Fin -> weight rows, i == Fin -> bias
        if (blk < 2) { if (i < Fin) pWzr[(int64_t)i * 64 + blk * 32 + j] = v; else if (i == Fin) pbzr[blk * 32 + j] = v; }
        else { if (i < Fin) pWh[(int64_t)i * 32 + j] = v; else if (i == Fin) pbh[j] = v; }
      } else {                             // the hidden rows
        if (blk < 2) pWzr[(int64_t)(Fin + i) * 64 + blk * 32 + j] = v;
        else pWh[(int64_t)(Fin + i) * 32 + j] = v;
      }
    }
    __syncthreads();
  }
}

// out[e] = sum over the workgroups' partials, in a FIXED order: a workgroup of sixteen wavefronts owns 64 elements, every wavefront
// adds a contiguous sixteenth of the partials (eight loads in flight per lane), the sixteenths meet in LDS in index order.  (One
// thread per element walking all 256 partials in turn was a 61 us launch; four wavefronts per 64 elements 8.6 us at 512 partials.)
__global__ __launch_bounds__(1024) void tgcn_cell_reduce_kernel(const float* __restrict__ part, int n_wg, int n, float* __restrict__ dWzr,
                                                               float* __restrict__ dbzr, float* __restrict__ dWh, float* __restrict__ dbh,
                                                               int C, int accumulate) {
  __shared__ float red[16][64];        // sixteen wavefronts, a contiguous sixteenth of the partials each
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int per = (n_wg + 15) / 16, w0 = q * per, w1 = (w0 + per < n_wg) ? w0 + per : n_wg;
  float acc = 0.f;
  if (e < n) {
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(w + u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; w < w1; ++w) acc += part[(int64_t)w * n + e];
  }
  red[q][lane] = acc;
  __syncthreads();
  if (q != 0 || e >= n) return;
  acc = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) acc += red[w][lane];          // in index order: the same sum launch after launch
  const int nWzr = C * 64, nWh = C * 32;
  float* dst = e < nWzr ? dWzr + e : e < nWzr + 64 ? (dbzr ? dbzr + (e - nWzr) : nullptr)
               : e < nWzr + 64 + nWh ? dWh + (e - nWzr - 64) : (dbh ? dbh + (e - nWzr - 64 - nWh) : nullptr);
  if (dst) *dst = accumulate ? *dst + acc : acc;           // (accumulate: a T-step loop's cells sum into one buffer, ops.py)
}

// ------------------------------------------------------------------------------------------------------------------------------
// Row-per-lane forms (Fin <= 2, every operand 16-byte addressable): the products are computed TRANSPOSED — D^T = W^T X^T — so the
// weights are the A operand (lane = output column, in registers for the whole launch) and a lane's B operand is ITS OWN ROW:
// lane (lo, hi) holds columns 8q + 4hi + j (q, j = 0 .. 3) of row lo, exactly the four 16-byte pieces a global_load_dwordx4
// brings in, and the accumulator comes back in the same layout (register r <-> column tc_row(r, hi), lane <-> row).  The
// contraction index is visited in that permuted order (any order is a valid sum; the weights are fetched to match), so
//   * every global access is a 16-byte row piece (the first forms moved 4 bytes per lane: 96 memory instructions per 32-row strip
//     against 21 here),
//   * H, Z | R, H * R, the candidate and the blend never leave the lane's registers: the forward kernel touches no LDS strip,
//   * the adjoint needs LDS only where the contraction runs over ROWS (the weight gradients): three row-major [32][36] matrices
//     per wavefront, written with ds_write_b128 and read column-wise, conflict-free both ways.
// Rounds 4's kernels above stay as the path for 3 <= Fin <= 30 and for operands that are not 16-byte addressable.

// Prefetches as 16-byte loads the compiler cannot see, released by a hand-placed wait.  Why not leave it to the compiler: its wait
// for a prefetched register sits wherever the register is first read, which for a loop-carried prefetch is the loop head, behind
// whatever the scheduler moved there.  Why the wait is vmcnt(0) and not "all but this strip's sixteen stores": vmcnt is ONE counter
// for loads and stores, loads retire in order among themselves but a younger store may be acknowledged BEFORE an older load lands,
// so a count that leaves the stores outstanding can pass with a load still in flight (csrc/gemm_bx.hip found this the hard way:
// four stale rows in 200 000, scripts/bx_sym_race_probe.py).  Measured, the exact-count form bought nothing anyway (76.6 vs
// 81.5 us): the kernel's time was in the CU's store path, not in this wait.
#ifdef PGT_EMU
#define TC_LOAD4(dst, p, OFF) ((dst) = *reinterpret_cast<const pgt_f4*>((p) + (OFF) / 4))
#define TC_LOAD1(dst, p) ((dst) = *(p))
#define TC_WAIT5(n, a, b, c, d, e) ((void)0)
#define TC_WAIT21(g, z, r, t, h, ax) ((void)0)
#else
#define TC_LOAD4(dst, p, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(p) : "memory")
#define TC_LOAD1(dst, p) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory")
#define TC_WAIT5(n, a, b, c, d, e) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(n))
// vmcnt(0) tied to the adjoint kernel's five prefetched row sets + the input column
#define TC_WAIT21(g, z, r, t, h, ax)                                                                                              \
  asm volatile("s_waitcnt vmcnt(0)"                                                                                               \
               : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(r[0]),      \
                 "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(h[0]), "+v"(h[1]),      \
                 "+v"(h[2]), "+v"(h[3]), "+v"(ax))
#endif

// bias: the accumulators start from it (12 broadcast ds_read_b128 per strip instead of 48 registers or three more MFMAs)
__device__ __forceinline__ void tc_bias_init(pgt_f32x16& acc, const float* sb, int hi) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const pgt_f4 b = *reinterpret_cast<const pgt_f4*>(sb + 8 * q + 4 * hi);
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
}

constexpr int TC_P = 36;            // row pitch of the adjoint's row-major LDS matrices (16-byte rows: ds_write_b128; 36 mod 32 = 4)

__device__ __forceinline__ void tc_park_rows(float* mat, int lo, int hi, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<pgt_f4*>(mat + lo * TC_P + 8 * q + 4 * hi) = pgt_mk4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// PROBE (pgt_tune("tgcn_probe"), lab use only): 1 = no MFMAs, 2 = no Z | R / candidate stores, 3 = neither — what the memory system
// alone makes of this access pattern.  0 is the kernel.
template <int PROBE>
__global__ __launch_bounds__(256, 2) void tgcn_cell_fwd_rows_kernel(TcArgs g) {
  __shared__ __attribute__((aligned(16))) float s_b[96];                   // bzr | bh
  // per wavefront: one [32 rows][36] matrix through which every result goes from "lane = row, 32-byte pieces" (the accumulator
  // layout) to whole 128-byte rows per eight lanes.  A store instruction in the accumulator layout writes 32 pieces of 32 bytes
  // (32 L2 requests: PMC TCP_TCC_WRITE_REQ = 32 per instruction) and the CU's store path, not HBM, set the kernel's time (no MFMAs:
  // 75 of 87 us; outputs cut to H' alone: 51 us); staged, an instruction writes eight complete rows.
  __shared__ __attribute__((aligned(16))) float s_o[4][32 * TC_P];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int Fin = g.Fin;
  float* mat = s_o[wave];
  const int srow = lane >> 3, spc = (lane & 7) * 4;                 // store role of the lane: rows srow + 8 j, floats spc .. spc + 3
  if (tid < 96) s_b[tid] = tid < 64 ? (g.bzr ? g.bzr[tid] : 0.f) : (g.bh ? g.bh[tid - 64] : 0.f);
  // A operands: this lane's column `lo` of the three weight blocks, k in the order the row pieces arrive; step 16 = the inputs
  // (coalesced loads staged through LDS once per workgroup, and H fetched as whole rows through the staging matrix, were tried
  // and measured: 64 -> 69 us; the L1 serves the repeated row pieces of the layout below well enough)
  float wz[17], wr[17], wh[17];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int64_t k = Fin + tc_row(s, hi);
    wz[s] = g.Wzr[k * 64 + lo]; wr[s] = g.Wzr[k * 64 + 32 + lo]; wh[s] = g.Wh[k * 32 + lo];
  }
  {
    const bool in = hi < Fin;
    wz[16] = in ? g.Wzr[(int64_t)hi * 64 + lo] : 0.f; wr[16] = in ? g.Wzr[(int64_t)hi * 64 + 32 + lo] : 0.f;
    wh[16] = in ? g.Wh[(int64_t)hi * 32 + lo] : 0.f;
  }
  __syncthreads();
  const float* __restrict__ Hg = g.H;
  const float* __restrict__ AXg = g.AX;
  float* __restrict__ ZRg = g.ZR;
  float* __restrict__ HTg = g.HT;
  float* __restrict__ Hng = g.Hn;
  // a strip's rows start at a wavefront-UNIFORM address (scalar registers); a lane adds a 32-bit offset of its own
  const int64_t n_strips = ((int64_t)g.M + 31) >> 5, n_full = (int64_t)g.M >> 5, stride = (int64_t)gridDim.x * 4;
  const int wave_u = PGT_UNIFORM(wave);
  pgt_f4 hq[4];
  float axv = 0.f;
  const int axc = hi < Fin ? hi : Fin - 1;                          // every lane loads (an unconditional instruction: the wait counts it)
  auto fetch = [&](int64_t st) {
    const int64_t r0 = st * 32, left = (int64_t)g.M - 1 - r0;
    const int lr = lo < left ? lo : (int)left;                       // rows past the end re-read the last one (never stored)
    const float* hp = Hg + r0 * g.ldh + (lr * (int)g.ldh + 4 * hi);
    const float* ap = AXg + r0 * g.ldax + (lr * (int)g.ldax + axc);
    TC_LOAD4(hq[0], hp, 0); TC_LOAD4(hq[1], hp, 32); TC_LOAD4(hq[2], hp, 64); TC_LOAD4(hq[3], hp, 96);
    TC_LOAD1(axv, ap);
  };
  // One strip.  FULL strips store without a branch; the one partial strip (stores behind `row < M`) runs after the loop.
  auto strip = [&](int64_t st, auto full_c, bool prefetch) {
    constexpr bool FULL = decltype(full_c)::value;
    float h[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[4 * q] = hq[q].x; h[4 * q + 1] = hq[q].y; h[4 * q + 2] = hq[q].z; h[4 * q + 3] = hq[q].w; }
    const float ax = hi < Fin ? axv : 0.f;
    if (prefetch) fetch(st + stride);                          // the next strip's pieces travel while this one is on the matrix cores
    pgt_f32x16 az, ar, ah;
    tc_bias_init(az, s_b, hi);
    tc_bias_init(ar, s_b + 32, hi);
    if constexpr ((PROBE & 1) == 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        az = PGT_MFMA_32x32x2(wz[s], h[s], az);
        ar = PGT_MFMA_32x32x2(wr[s], h[s], ar);
      }
      az = PGT_MFMA_32x32x2(wz[16], ax, az);
      ar = PGT_MFMA_32x32x2(wr[16], ax, ar);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { az[r] += h[r] * wz[r]; ar[r] += h[r] * wr[r] + ax; }
    }
    const int64_t r0 = st * 32;
    const bool ok = FULL || r0 + lo < g.M;
    float z[16], hr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      z[r] = tc_sigmoidf(az[r]);
      ar[r] = tc_sigmoidf(ar[r]);
      hr[r] = h[r] * ar[r];
    }
    auto put = [&](const float (&v)[16], float* dst, int ld) {      // v: this lane's 16 columns of its row -> dst[row][0 .. 31]
      tc_park_rows(mat, lo, hi, v);
      PGT_WAVE_SYNC();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = srow + 8 * j;
        const pgt_f4 t4 = *reinterpret_cast<const pgt_f4*>(mat + row * TC_P + spc);
        if (FULL || r0 + row < g.M) *reinterpret_cast<pgt_f4*>(dst + (row * ld + spc)) = t4;
      }
      PGT_WAVE_SYNC();
    };
    (void)ok;
    if constexpr ((PROBE & 2) == 0) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = ar[r];
      put(z, ZRg + r0 * 64, 64);
      put(rv, ZRg + r0 * 64 + 32, 64);
    }
    tc_bias_init(ah, s_b + 64, hi);
    if constexpr ((PROBE & 1) == 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) ah = PGT_MFMA_32x32x2(wh[s], hr[s], ah);
      ah = PGT_MFMA_32x32x2(wh[16], ax, ah);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) ah[r] += hr[r] * wh[r] + ax;
    }
    {
      float tv[16], nv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { tv[r] = tc_tanhf(ah[r]); nv[r] = pgt_gru_blend(z[r], h[r], tv[r]); }
      if constexpr ((PROBE & 2) == 0) put(tv, HTg + r0 * 32, 32);
      put(nv, Hng + r0 * g.ldhn, (int)g.ldhn);
    }
    if (prefetch) {
      TC_WAIT5(0, hq[0], hq[1], hq[2], hq[3], axv);              // (everything: see TC_LOAD4)
    }
  };
  int64_t st = (int64_t)blockIdx.x * 4 + wave_u;
  if (st < n_strips) {
    fetch(st);
    TC_WAIT5(0, hq[0], hq[1], hq[2], hq[3], axv);
  }
  for (; st + stride < n_full; st += stride) strip(st, std::true_type{}, true);    // (the prefetched strip is a full one too)
  if (st < n_full) {                                                                // this wavefront's last full strip; a partial one may follow
    const bool more = st + stride < n_strips;
    strip(st, std::true_type{}, more);
    st += stride;
  }
  if (st < n_strips) strip(st, std::false_type{}, false);                           // the partial strip, if this wavefront owns it
}

__global__ __launch_bounds__(256, 2) void tgcn_cell_bwd_rows_kernel(TcArgs g) {
  __shared__ __attribute__((aligned(16))) float s_t[4][3 * 32 * TC_P];     // per wavefront: M0 | M1 | M2, each [32 rows][36]
  __shared__ __attribute__((aligned(16))) float s_ax[4][32 * 2];            // per wavefront: the strip's input columns [row][2]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int Fin = g.Fin, C = Fin + TC_O;
  float* M0 = s_t[wave];
  float* M1 = M0 + 32 * TC_P;
  float* M2 = M1 + 32 * TC_P;
  float* axs = s_ax[wave];
  const float* m0l = M0 + hi * TC_P + lo;      // this lane's MFMA operand columns: row 2 s + hi of a matrix = base + 2 s TC_P
  const float* m1l = M1 + hi * TC_P + lo;
  const float* m2l = M2 + hi * TC_P + lo;
  const float* axl = axs + hi * 2;              // the input columns of row 2 s + hi = base + 4 s
  // A operands of the two "weights x d_pre^T" products: row Fin + lo of the hidden blocks, k in the order of the row pieces —
  // [step][lane] in LDS (48 conflict-free reads per strip; as registers they pushed the kernel past 256)
  __shared__ float s_w[48 * 64];
  for (int e = tid; e < 48 * 64; e += 256) {
    const int s = (e >> 6) & 15, blk = e >> 10, l = e & 63, c = tc_row(s, l >> 5);
    const int64_t i = Fin + (l & 31);
    s_w[e] = blk == 0 ? g.Wh[i * 32 + c] : g.Wzr[i * 64 + (blk - 1) * 32 + c];
  }
  const float* a1 = s_w + lane;
  const float* a2z = s_w + 1024 + lane;
  const float* a2r = s_w + 2048 + lane;
  __syncthreads();
  const float* __restrict__ dHg = g.dHn;
  const float* __restrict__ ZRg = g.ZR;
  const float* __restrict__ HTg = g.HT;
  const float* __restrict__ Hg = g.H;
  const float* __restrict__ AXg = g.AX;
  pgt_f32x16 wz1a, wz1b, wh1;                          // H^T d_pre_z, H^T d_pre_r, (H R)^T d_pre_h
#pragma unroll
  for (int r = 0; r < 16; ++r) { wz1a[r] = 0.f; wz1b[r] = 0.f; wh1[r] = 0.f; }
  float tz[3] = {0.f, 0.f, 0.f}, th[3] = {0.f, 0.f, 0.f};   // [AX | 1]^T d: column `lane` of d_pre_z | d_pre_r; column lo of d_pre_h (half the rows per hi)
  const int64_t n_strips = ((int64_t)g.M + 31) >> 5, stride = (int64_t)gridDim.x * 4;
  const int wave_u = PGT_UNIFORM(wave);
  const int axc = hi < Fin ? hi : Fin - 1;
  // The next strip's twenty row pieces are requested when this strip's operands have moved to LDS (the 48 weight-gradient MFMAs
  // that follow run from there, and eighty registers are free): invisible to the compiler (TC_LOAD4), released by one hand-placed
  // wait at the top of the next round.  Loaded the plain way, each strip opened with ~2 us of HBM latency nothing covered.
  pgt_f4 pg[4], pz[4], pr[4], pt[4], ph[4];
  float pax = 0.f;
  auto fetch = [&](int64_t st) {
    const int64_t r0 = st * 32, left = (int64_t)g.M - 1 - r0;
    const int lr = lo < left ? lo : (int)left;                       // rows past the end re-read the last one (their g is zeroed)
    const float* gp = dHg + r0 * g.lddhn + (lr * (int)g.lddhn + 4 * hi);
    const float* zp = ZRg + r0 * 64 + (lr * 64 + 4 * hi);
    const float* tp = HTg + r0 * 32 + (lr * 32 + 4 * hi);
    const float* hp = Hg + r0 * g.ldh + (lr * (int)g.ldh + 4 * hi);
    const float* ap = AXg + r0 * g.ldax + (lr * (int)g.ldax + axc);
    TC_LOAD4(pg[0], gp, 0); TC_LOAD4(pg[1], gp, 32); TC_LOAD4(pg[2], gp, 64); TC_LOAD4(pg[3], gp, 96);
    TC_LOAD4(pz[0], zp, 0); TC_LOAD4(pz[1], zp, 32); TC_LOAD4(pz[2], zp, 64); TC_LOAD4(pz[3], zp, 96);
    TC_LOAD4(pr[0], zp, 128); TC_LOAD4(pr[1], zp, 160); TC_LOAD4(pr[2], zp, 192); TC_LOAD4(pr[3], zp, 224);
    TC_LOAD4(pt[0], tp, 0); TC_LOAD4(pt[1], tp, 32); TC_LOAD4(pt[2], tp, 64); TC_LOAD4(pt[3], tp, 96);
    TC_LOAD4(ph[0], hp, 0); TC_LOAD4(ph[1], hp, 32); TC_LOAD4(ph[2], hp, 64); TC_LOAD4(ph[3], hp, 96);
    TC_LOAD1(pax, ap);
  };
  int64_t st = (int64_t)blockIdx.x * 4 + wave_u;
  if (st < n_strips) fetch(st);
  for (; st < n_strips; st += stride) {
    const int64_t r0 = st * 32, left = (int64_t)g.M - 1 - r0;
    const bool ok = lo <= left;
    TC_WAIT21(pg, pz, pr, pt, ph, pax);                                // nothing younger than the prefetch is in flight
    float gg[16], z[16], rr[16], t[16], h[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gg[4 * q] = pg[q].x; gg[4 * q + 1] = pg[q].y; gg[4 * q + 2] = pg[q].z; gg[4 * q + 3] = pg[q].w;
      z[4 * q] = pz[q].x; z[4 * q + 1] = pz[q].y; z[4 * q + 2] = pz[q].z; z[4 * q + 3] = pz[q].w;
      rr[4 * q] = pr[q].x; rr[4 * q + 1] = pr[q].y; rr[4 * q + 2] = pr[q].z; rr[4 * q + 3] = pr[q].w;
      t[4 * q] = pt[q].x; t[4 * q + 1] = pt[q].y; t[4 * q + 2] = pt[q].z; t[4 * q + 3] = pt[q].w;
      h[4 * q] = ph[q].x; h[4 * q + 1] = ph[q].y; h[4 * q + 2] = ph[q].z; h[4 * q + 3] = ph[q].w;
    }
    const float ax = (ok && hi < Fin) ? pax : 0.f;
    float dph[16], dz[16], gz[16], hr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float g1 = ok ? gg[r] : 0.f;                               // rows past the end contribute nothing anywhere
      dph[r] = g1 * (1.f - z[r]) * (1.f - t[r] * t[r]);                // d_pre_h
      dz[r] = g1 * (h[r] - t[r]) * z[r] * (1.f - z[r]);                // d_pre_z
      gz[r] = g1 * z[r];
      hr[r] = h[r] * rr[r];
    }
    // the row-contracting products need (row, column) transposed: their operands go row-major into the wavefront's strip.
    // First H R and d_pre_h (the candidate's weight gradient), so that H, d_pre_z, d_pre_r can stay in registers meanwhile.
    tc_park_rows(M0, lo, hi, hr);
    tc_park_rows(M1, lo, hi, dph);
    axs[lo * 2 + hi] = ax;
    pgt_f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) p = PGT_MFMA_32x32x2(a1[s * 64], dph[s], p);             // d(H R)^T = Wh_H d_pre_h^T
    float dr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dr[r] = p[r] * hr[r] * (1.f - rr[r]);                            // d_pre_r
      gz[r] = fmaf(p[r], rr[r], gz[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      p = PGT_MFMA_32x32x2(a2z[s * 64], dz[s], p);                     // d_pre_zr Wzr_H^T, transposed
      p = PGT_MFMA_32x32x2(a2r[s * 64], dr[s], p);
    }
    if (ok) {
      float* dp = g.dH + r0 * g.lddh + (lo * (int)g.lddh + 4 * hi);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<pgt_f4*>(dp + 8 * q) = pgt_mk4(gz[4 * q] + p[4 * q], gz[4 * q + 1] + p[4 * q + 1], gz[4 * q + 2] + p[4 * q + 2],
                                                          gz[4 * q + 3] + p[4 * q + 3]);
    }
    PGT_WAVE_SYNC();
#pragma unroll
    for (int s = 0; s < 16; ++s) {                                     // (lane bases + compile-time offsets: rows 2 s + hi)
      const float d = m1l[2 * s * TC_P];
      wh1 = PGT_MFMA_32x32x2(m0l[2 * s * TC_P], d, wh1);            // dWh (hidden rows) += (H R)^T d_pre_h: A[m = feature][k = row]
      // its input rows and bias ([AX | 1]^T d) on the vector unit: three sums per column instead of a 32-row MFMA block of which
      // three rows are alive (a third of the first form's matrix-core time); this lane's half of the rows
      th[0] = fmaf(axl[4 * s], d, th[0]);
      th[1] = fmaf(axl[4 * s + 1], d, th[1]);
      th[2] += d;
    }
    PGT_WAVE_SYNC();
    tc_park_rows(M0, lo, hi, h);
    tc_park_rows(M1, lo, hi, dz);
    tc_park_rows(M2, lo, hi, dr);
    if (st + stride < n_strips) fetch(st + stride);                    // (every operand of what follows is in LDS now)
    PGT_WAVE_SYNC();
    // dWzr (hidden rows) += H^T [d_pre_z | d_pre_r]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = m0l[2 * s * TC_P], bz = m1l[2 * s * TC_P], br = m2l[2 * s * TC_P];
      wz1a = PGT_MFMA_32x32x2(a, bz, wz1a);
      wz1b = PGT_MFMA_32x32x2(a, br, wz1b);
    }
    {
      const float* dcol = (hi ? M2 : M1) + lo;                         // lane = column `lane` of [d_pre_z | d_pre_r]
#pragma unroll 4
      for (int rho = 0; rho < 32; ++rho) {
        const float d = dcol[rho * TC_P];
        tz[0] = fmaf(axs[rho * 2], d, tz[0]);
        tz[1] = fmaf(axs[rho * 2 + 1], d, tz[1]);
        tz[2] += d;
      }
    }
    PGT_WAVE_SYNC();                                                   // the strip is rewritten by the next one
  }
  // ---- the four wavefronts' sums meet in LDS (the strips are dead), one partial per workgroup
  __syncthreads();
  float* red = &s_t[0][0];                                             // 4 x 3 x 1024 accumulator floats, then 4 x 384 thin sums
  float* thin = red + 4 * 3 * 1024;
  {
    float* mine = red + wave * 3 * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = tc_row(r, hi);
      mine[i * 32 + lo] = wz1a[r]; mine[1024 + i * 32 + lo] = wz1b[r]; mine[2048 + i * 32 + lo] = wh1[r];
    }
    float* tm = thin + wave * 384;                                     // [3][64] of d_pre_zr, then [3][2 halves][32] of d_pre_h
#pragma unroll
    for (int k = 0; k < 3; ++k) { tm[k * 64 + lane] = tz[k]; tm[192 + k * 64 + hi * 32 + lo] = th[k]; }
  }
  __syncthreads();
  float* part = g.part + (int64_t)blockIdx.x * tc_part_floats(Fin);
  float* pWzr = part;
  float* pbzr = part + (int64_t)C * 64;
  float* pWh = pbzr + 64;
  float* pbh = pWh + (int64_t)C * 32;
  for (int e = tid; e < 3 * 1024; e += 256) {
    const float v = ((red[e] + red[3 * 1024 + e]) + red[6 * 1024 + e]) + red[9 * 1024 + e];
    const int blk = e >> 10, i = (e >> 5) & 31, j = e & 31;
    if (blk < 2) pWzr[(int64_t)(Fin + i) * 64 + blk * 32 + j] = v;
    else pWh[(int64_t)(Fin + i) * 32 + j] = v;
  }
  for (int e = tid; e < 192 + 96; e += 256) {
    if (e < 192) {                                                     // k = 0, 1: input rows of dWzr; k = 2: dbzr
      const int k = e >> 6, j = e & 63;
      const float v = ((thin[e] + thin[384 + e]) + thin[768 + e]) + thin[1152 + e];
      if (k < Fin) pWzr[(int64_t)k * 64 + j] = v; else if (k == 2) pbzr[j] = v;
    } else {
      const int k = (e - 192) >> 5, j = (e - 192) & 31;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += thin[w * 384 + 192 + k * 64 + j] + thin[w * 384 + 192 + k * 64 + 32 + j];
      if (k < Fin) pWh[(int64_t)k * 32 + j] = v; else if (k == 2) pbh[j] = v;
    }
  }
}

#ifdef PGT_EMU
constexpr int TC_WGS = 3;
constexpr int TC_WGS_ROWS_FWD = 3, TC_WGS_ROWS_BWD = 3;
#else
constexpr int TC_WGS = 256;
constexpr int TC_WGS_ROWS_FWD = 768;      // three 256-thread workgroups per CU (168 registers), no LDS strips
constexpr int TC_WGS_ROWS_BWD = 512;      // two per CU (<= 256 registers, 68 KB of LDS each)
#endif

// the row-per-lane kernels move 16-byte row pieces: every base pointer and row stride must allow it
bool tc_rows_ok(int64_t Fin, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
  if (g_tc_rows == 0 || Fin > 2) return false;
  for (const void* p : ptrs) if (!pgt_aligned(p, 16)) return false;
  for (int64_t ld : lds) if (ld % 4) return false;
  return true;
}

}  // namespace

extern "C" int pgt_tgcn_cell_fits(int64_t Fin, int64_t O) { return (O == TC_O && Fin >= 1 && Fin <= 30) ? 1 : 0; }

extern "C" int64_t pgt_tgcn_cell_bwd_ws_floats(int64_t Fin, int64_t O) {
  if (!pgt_tgcn_cell_fits(Fin, O)) return 0;
  return (int64_t)(TC_WGS_ROWS_BWD > TC_WGS ? TC_WGS_ROWS_BWD : TC_WGS) * ((Fin + TC_O) * 96 + 96);
}

void pgt_tgcn_set_rows(int v) { g_tc_rows = v ? 1 : 0; }
void pgt_tgcn_set_wgs(int v) { g_tc_wgs = v > 0 ? v : 0; }          // pgt_tune("tgcn_wgs", n): at most n workgroups per launch (0: the default)
// pgt_tune("tgcn_probe", n): forward-kernel variants with parts switched off (WRONG results: what a phase costs) — compiled only
// into a library built with -DPGT_LAB_PROBES; the product library rejects the key
int pgt_tgcn_set_probe(int v) {
#ifdef PGT_LAB_PROBES
  g_tc_probe = v;
  return 1;
#else
  (void)v;
  return 0;
#endif
}

extern "C" int pgt_tgcn_cell_f32(const float* AX, int64_t ldax, const float* H, int64_t ldh, const float* Wzr, const float* bzr,
                                 const float* Wh, const float* bh, int64_t M, int64_t Fin, int64_t O, float* ZR, float* HT, float* Hn,
                                 int64_t ldhn, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0, "pgt_tgcn_cell_f32: negative size");
  PGT_REQUIRE(pgt_tgcn_cell_fits(Fin, O), "pgt_tgcn_cell_f32: built for hidden width 32 and 1 .. 30 input columns (got %lld, %lld)",
              (long long)O, (long long)Fin);
  if (M == 0) return PGT_OK;
  PGT_REQUIRE(AX && H && Wzr && Wh && ZR && HT && Hn, "pgt_tgcn_cell_f32: null pointer");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - 256 && ldax >= Fin && ldh >= O && ldhn >= O, "pgt_tgcn_cell_f32: extent out of range");
  TcArgs g{};
  g.AX = AX; g.ldax = ldax; g.H = H; g.ldh = ldh; g.Wzr = Wzr; g.bzr = bzr; g.Wh = Wh; g.bh = bh;
  g.ZR = ZR; g.HT = HT; g.Hn = Hn; g.ldhn = ldhn; g.M = (int)M; g.Fin = (int)Fin; g.tiles = (int)pgt_cdiv(M, 128);
  if (tc_rows_ok(Fin, {H, ZR, HT, Hn}, {ldh, ldhn})) {
    const int64_t cap = g_tc_wgs > 0 ? g_tc_wgs : TC_WGS_ROWS_FWD;
    const int64_t wg = g.tiles < cap ? g.tiles : cap;
    switch (g_tc_probe) {
#ifdef PGT_LAB_PROBES
      case 1: PGT_LAUNCH(tgcn_cell_fwd_rows_kernel<1>, dim3((unsigned)wg), dim3(256), stream, g); break;
      case 2: PGT_LAUNCH(tgcn_cell_fwd_rows_kernel<2>, dim3((unsigned)wg), dim3(256), stream, g); break;
      case 3: PGT_LAUNCH(tgcn_cell_fwd_rows_kernel<3>, dim3((unsigned)wg), dim3(256), stream, g); break;
#endif
      default: PGT_LAUNCH(tgcn_cell_fwd_rows_kernel<0>, dim3((unsigned)wg), dim3(256), stream, g);
    }
    return pgt_check_launch("pgt_tgcn_cell_f32");
  }
  int64_t wgs = g.tiles < 4 * TC_WGS ? g.tiles : 4 * TC_WGS;
  if (Fin <= 2) PGT_LAUNCH((tgcn_cell_fwd_kernel<34>), dim3((unsigned)wgs), dim3(256), stream, g);
  else PGT_LAUNCH((tgcn_cell_fwd_kernel<0>), dim3((unsigned)wgs), dim3(256), stream, g);
  return pgt_check_launch("pgt_tgcn_cell_f32");
}

static int tc_bwd_impl(const float* dHn, int64_t lddhn, const float* AX, int64_t ldax, const float* H, int64_t ldh,
                       const float* ZR, const float* HT, const float* Wzr, const float* Wh, int64_t M, int64_t Fin,
                       int64_t O, float* dH, int64_t lddh, float* dWzr, float* dbzr, float* dWh, float* dbh, int accumulate, float* ws,
                       int64_t ws_floats, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0, "pgt_tgcn_cell_bwd_f32: negative size");
  PGT_REQUIRE(pgt_tgcn_cell_fits(Fin, O), "pgt_tgcn_cell_bwd_f32: built for hidden width 32 and 1 .. 30 input columns");
  PGT_REQUIRE(dWzr && dWh, "pgt_tgcn_cell_bwd_f32: null weight gradient");
  const int C = (int)Fin + TC_O, n = C * 96 + 96;
  if (M == 0) {
    if (accumulate) return PGT_OK;
    if (hipMemsetAsync(dWzr, 0, (size_t)C * 64 * 4, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(dWh, 0, (size_t)C * 32 * 4, (hipStream_t)stream) != hipSuccess ||
        (dbzr && hipMemsetAsync(dbzr, 0, 64 * 4, (hipStream_t)stream) != hipSuccess) ||
        (dbh && hipMemsetAsync(dbh, 0, 32 * 4, (hipStream_t)stream) != hipSuccess)) {
      pgt_set_error("pgt_tgcn_cell_bwd_f32: memset failed");
      return PGT_ERR_LAUNCH;
    }
    return PGT_OK;
  }
  PGT_REQUIRE(dHn && AX && H && ZR && HT && Wzr && Wh && dH && ws, "pgt_tgcn_cell_bwd_f32: null pointer");
  PGT_REQUIRE(ws_floats >= pgt_tgcn_cell_bwd_ws_floats(Fin, O), "pgt_tgcn_cell_bwd_f32: scratch too small (pgt_tgcn_cell_bwd_ws_floats)");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - 256 && ldax >= Fin && ldh >= O && lddh >= O && lddhn >= O, "pgt_tgcn_cell_bwd_f32: extent out of range");
  TcArgs g{};
  g.AX = AX; g.ldax = ldax; g.H = H; g.ldh = ldh; g.Wzr = Wzr; g.Wh = Wh; g.ZR = const_cast<float*>(ZR); g.HT = const_cast<float*>(HT);
  g.dHn = dHn; g.lddhn = lddhn; g.dH = dH; g.lddh = lddh; g.part = ws;
  g.M = (int)M; g.Fin = (int)Fin; g.tiles = (int)pgt_cdiv(M, 128);
  const bool rows = tc_rows_ok(Fin, {dHn, H, ZR, HT, dH}, {lddhn, ldh, lddh});
  const int cap = rows ? TC_WGS_ROWS_BWD : TC_WGS;
  const int wgs = g.tiles < cap ? g.tiles : cap;
  g.n_wg = wgs;
  if (rows) PGT_LAUNCH(tgcn_cell_bwd_rows_kernel, dim3((unsigned)wgs), dim3(256), stream, g);
  else PGT_LAUNCH(tgcn_cell_bwd_kernel, dim3((unsigned)wgs), dim3(256), stream, g);
  PGT_LAUNCH(tgcn_cell_reduce_kernel, dim3((unsigned)pgt_cdiv(n, 64)), dim3(1024), stream, ws, wgs, n, dWzr, dbzr, dWh, dbh, C,
             accumulate);
  return pgt_check_launch("pgt_tgcn_cell_bwd_f32");
}

extern "C" int pgt_tgcn_cell_bwd_f32(const float* dHn, int64_t lddhn, const float* AX, int64_t ldax, const float* H, int64_t ldh,
                                     const float* ZR, const float* HT, const float* Wzr, const float* Wh, int64_t M, int64_t Fin,
                                     int64_t O, float* dH, int64_t lddh, float* dWzr, float* dbzr, float* dWh, float* dbh, float* ws,
                                     int64_t ws_floats, pgt_stream_t stream) {
  return tc_bwd_impl(dHn, lddhn, AX, ldax, H, ldh, ZR, HT, Wzr, Wh, M, Fin, O, dH, lddh, dWzr, dbzr, dWh, dbh, 0, ws, ws_floats, stream);
}

extern "C" int pgt_tgcn_cell_bwd_acc_f32(const float* dHn, int64_t lddhn, const float* AX, int64_t ldax, const float* H, int64_t ldh,
                                         const float* ZR, const float* HT, const float* Wzr, const float* Wh, int64_t M, int64_t Fin,
                                         int64_t O, float* dH, int64_t lddh, float* dWzr, float* dbzr, float* dWh, float* dbh, float* ws,
                                         int64_t ws_floats, pgt_stream_t stream) {
  return tc_bwd_impl(dHn, lddhn, AX, ldax, H, ldh, ZR, HT, Wzr, Wh, M, Fin, O, dH, lddh, dWzr, dbzr, dWh, dbh, 1, ws, ws_floats, stream);
}
