// Dense feature transform on the matrix cores, exact fp32: v_mfma_f32_32x32x2_f32
// (f32 in / f32 accumulate, bitwise an fmaf chain; 157 TFLOP/s peak on MI355X — there is no TF32 on gfx950).
//
//  gemm_kernel      C(m,n) = sum_j A_j[m,:] . Bw[j*seg_k:(j+1)*seg_k, n] + bias[n]  (+C)
//                   A is "segmented along K": the 2K-1 diffusion terms T_k of DConv (dcrnn.py:81-105) stay in
//                   their own [M, C] buffers and are consumed as one [M, (2K-1)*C] operand — no concatenation.
//                   Generic B strides make the same kernel serve X.W (NN) and dY.W^T (NT, feature gradient).
//  gemm_tn_kernel   dW += A^T G over a slab of rows per workgroup, fp32 atomics (weight gradient), and the
//                   bias gradient (column sums of G) on the side.
//
// Tile: 256 threads = 4 wavefronts, 64x64 output tile, each wavefront one 32x32 accumulator (16 VGPRs),
// BK = 32.  A is staged k-major in LDS (row stride 65 floats: conflict-free ds_write_b32 and ds_read_b32 for
// the MFMA A-operand map lane -> A[i = lane&31][k = lane>>5]); B is staged k-major, natural.
#include "pgt_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32;

__global__ __launch_bounds__(256) void gemm_kernel(
    const float* __restrict__ A, int64_t lda, int64_t a_seg_stride, int n_seg, int seg_k,
    const float* __restrict__ Bw, int64_t sbk, int64_t sbn, float* C, int64_t ldc, int64_t c_seg_stride,
    int c_seg_n, const float* __restrict__ bias, int M, int N, int accumulate) {
  __shared__ float As[BK][BM + 1];
  __shared__ float Bs[BK][BN];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
  const int Ktot = n_seg * seg_k;

  pgt_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int a_kk = tid & 31, a_mm = tid >> 5;  // A tile: lanes run along k (contiguous in memory)
  const int b_n = tid & 63, b_kq = tid >> 6;   // B tile: lanes run along n
  const int gn_b = n0 + b_n;

  for (int k0 = 0; k0 < Ktot; k0 += BK) {
    {
      const int kg = k0 + a_kk;
      const bool kv = kg < Ktot;
      const int j = kv ? kg / seg_k : 0;
      const int c = kg - j * seg_k;
      const float* base = A + (int64_t)j * a_seg_stride + c;
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) {
        const int m = a_mm + 8 * i;
        const int gm = m0 + m;
        As[a_kk][m] = (kv && gm < M) ? base[(int64_t)gm * lda] : 0.f;
      }
    }
    {
#pragma unroll
      for (int i = 0; i < BK / 4; ++i) {
        const int k = b_kq + 4 * i;
        const int kg = k0 + k;
        Bs[k][b_n] = (kg < Ktot && gn_b < N) ? Bw[(int64_t)kg * sbk + (int64_t)gn_b * sbn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = PGT_MFMA_32x32x2(a, b, acc);
    }
    __syncthreads();
  }

  // D map (cdna_hip_programming.md §3): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int gn = n0 + wn * 32 + (lane & 31);
  if (gn < N) {
    const float bv = bias ? bias[gn] : 0.f;
    const int js = gn / c_seg_n;
    float* cbase = C + (int64_t)js * c_seg_stride + (gn - js * c_seg_n);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (gm < M) {
        float v = acc[r] + bv;
        float* p = cbase + (int64_t)gm * ldc;
        if (accumulate) v += *p;
        *p = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(
    const float* __restrict__ A, int64_t lda, int64_t a_seg_stride, int n_seg, int seg_k,
    const float* __restrict__ G, int64_t ldg, float* dW, int64_t lddw, float* db, int M, int N,
    int rows_per_slab) {
  __shared__ float As[BK][BM];  // [m][kc]
  __shared__ float Gs[BK][BN];  // [m][n]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wk = wave & 1, wn = wave >> 1;
  const int k0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
  const int Ktot = n_seg * seg_k;
  const int ms = (int)blockIdx.z * rows_per_slab;
  const int me = (ms + rows_per_slab < M) ? ms + rows_per_slab : M;

  pgt_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int l_c = tid & 63, l_mq = tid >> 6;  // lanes along the column (kc or n), 4 row phases
  const int kg = k0 + l_c;
  const bool kv = kg < Ktot;
  const int j = kv ? kg / seg_k : 0;
  const float* abase = A + (int64_t)j * a_seg_stride + (kg - j * seg_k);
  const int gn_l = n0 + l_c;
  const bool nv = gn_l < N;
  const bool do_bias = (db != nullptr) && (blockIdx.x == 0) && (tid < BN);
  float bsum = 0.f;

  for (int mb = ms; mb < me; mb += BK) {
#pragma unroll
    for (int i = 0; i < BK / 4; ++i) {
      const int m = l_mq + 4 * i;
      const int gm = mb + m;
      const bool mv = gm < me;
      As[m][l_c] = (kv && mv) ? abase[(int64_t)gm * lda] : 0.f;
      Gs[m][l_c] = (nv && mv) ? G[(int64_t)gm * ldg + gn_l] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < BK; mm += 2) {
      const float a = As[mm + (lane >> 5)][wk * 32 + (lane & 31)];
      const float b = Gs[mm + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = PGT_MFMA_32x32x2(a, b, acc);
    }
    if (do_bias) {
#pragma unroll
      for (int m = 0; m < BK; ++m) bsum += Gs[m][tid];
    }
    __syncthreads();
  }

  const int gn = n0 + wn * 32 + (lane & 31);
  if (gn < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gk = k0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (gk < Ktot) atomicAdd(dW + (int64_t)gk * lddw + gn, acc[r]);
    }
  }
  if (do_bias && (n0 + tid) < N) atomicAdd(db + n0 + tid, bsum);
}

}  // namespace

extern "C" int pgt_gemm_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                            const float* Bw, int64_t sbk, int64_t sbn, float* C, int64_t ldc,
                            int64_t c_seg_stride, int64_t c_seg_n, const float* bias, int64_t M, int64_t N,
                            int accumulate, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && N >= 0 && n_seg >= 0 && seg_k >= 0, "pgt_gemm_f32: negative size");
  if (M == 0 || N == 0) return PGT_OK;
  PGT_REQUIRE(C != nullptr, "pgt_gemm_f32: null output");
  PGT_REQUIRE(n_seg * seg_k == 0 || (A && Bw), "pgt_gemm_f32: null operand");
  PGT_REQUIRE(c_seg_n > 0, "pgt_gemm_f32: c_seg_n must be positive");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - BM && N < ((int64_t)1 << 31) - BN && n_seg * seg_k < ((int64_t)1 << 31) - BK,
              "pgt_gemm_f32: size exceeds int32 indexing");
  const int64_t gx = pgt_cdiv(M, BM), gy = pgt_cdiv(N, BN);
  PGT_REQUIRE(gy <= 65535, "pgt_gemm_f32: N too large");
  dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  PGT_LAUNCH(gemm_kernel, grid, block, stream, A, lda, a_seg_stride, (int)n_seg, (int)(seg_k > 0 ? seg_k : 1), Bw,
             sbk, sbn, C, ldc, c_seg_stride, (int)c_seg_n, bias, (int)M, (int)N, accumulate);
  return pgt_check_launch("pgt_gemm_f32");
}

extern "C" int pgt_gemm_tn_acc_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg,
                                   int64_t seg_k, const float* G, int64_t ldg, float* dW, int64_t lddw,
                                   float* db, int64_t M, int64_t N, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && N >= 0 && n_seg >= 0 && seg_k >= 0, "pgt_gemm_tn_acc_f32: negative size");
  const int64_t Ktot = n_seg * seg_k;
  if (M == 0 || N == 0) return PGT_OK;
  PGT_REQUIRE(G != nullptr, "pgt_gemm_tn_acc_f32: null gradient");
  PGT_REQUIRE(Ktot == 0 || (A && dW), "pgt_gemm_tn_acc_f32: null operand");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - 4096 && N < ((int64_t)1 << 31) - BN && Ktot < ((int64_t)1 << 31) - BM,
              "pgt_gemm_tn_acc_f32: size exceeds int32 indexing");
  // at least one k-tile so that the bias gradient (blockIdx.x == 0) is produced even when Ktot == 0
  const int64_t gx = Ktot > 0 ? pgt_cdiv(Ktot, BM) : 1, gy = pgt_cdiv(N, BN);
  PGT_REQUIRE(gy <= 65535, "pgt_gemm_tn_acc_f32: N too large");
  // enough row slabs for ~1024 workgroups, each slab a multiple of BK rows
  int64_t nslab = pgt_cdiv(1024, gx * gy);
  const int64_t max_slab = pgt_cdiv(M, BK);
  if (nslab > max_slab) nslab = max_slab;
  if (nslab > 65535) nslab = 65535;
  if (nslab < 1) nslab = 1;
  int64_t rows = pgt_cdiv(pgt_cdiv(M, nslab), BK) * BK;
  nslab = pgt_cdiv(M, rows);
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)nslab), block(256);
  PGT_LAUNCH(gemm_tn_kernel, grid, block, stream, A, lda, a_seg_stride, (int)n_seg, (int)(seg_k > 0 ? seg_k : 1), G,
             ldg, dW, lddw, db, (int)M, (int)N, (int)rows);
  return pgt_check_launch("pgt_gemm_tn_acc_f32");
}
