// Dense attention score matrices of ASTGCN (astgcn.py:201-328): S = softmax_dim1( V . sigmoid( L R + b ) ) for a batch
// of B matrices with n x n scores (n = nodes for the spatial, time steps for the temporal attention), L [B, n, m],
// R [B, m, n] the two embeddings (m = the other extent: small for the spatial attention).
//
// The reference materialises L R [B, n, n], + b, sigmoid, the batched product with V and the softmax as five torch
// ops.  Here the intermediate lives in ONE layout, [i][b][j] (row of the score matrix outermost), which makes every
// access of every stage coalesced along j and turns the batched V . sigma_b into a single MFMA GEMM
// [n, n] x [n, B n] through pgt_gemm_f32:
//   att_sigmoid_scores_kernel : sig[i][b][j] = sigmoid( sum_t L[b,i,t] R[b,t,j] + bias[i,j] )     (L R never stored)
//   pgt_gemm_f32              : C[i][b][j]   = sum_k V[i,k] sig[k][b][j]
//   att_softmax_rows_kernel   : S[b][i][j]   = exp(C[i][b][j] - max_i) / sum_i                    (softmax over dim 1)
// and for the backward
//   att_softmax_rows_bwd_kernel : dC[i][b][j] = S (dS - sum_i dS S)
//   pgt_gemm_f32 (x2)           : dV = dC sig^T,  dsig = V^T dC
//   att_sigmoid_bwd_kernel      : dP[b][i][j] = dsig sig (1 - sig)   (back in batch-major layout for the small products
//                                 with L and R that follow), dbias[i][j] += sum_b dP
#include "pgt_common.h"

namespace {

// one thread per (i, b, j); lanes run along j: R[b,t,j], bias[i,j] and the store are coalesced, L[b,i,t] is a broadcast
__global__ __launch_bounds__(256) void att_sigmoid_scores_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                                 const float* __restrict__ bias, int B, int n, int m,
                                                                 float* __restrict__ sig) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n * B * n;
  if (e >= total) return;
  const int j = (int)(e % n);
  const int b = (int)((e / n) % B);
  const int i = (int)(e / ((int64_t)n * B));
  const float* l = L + ((int64_t)b * n + i) * m;
  const float* r = R + (int64_t)b * m * n + j;
  float acc = 0.f;
  for (int t = 0; t < m; ++t) acc = fmaf(l[t], r[(int64_t)t * n], acc);
  sig[e] = pgt_sigmoidf(acc + bias[(int64_t)i * n + j]);
}

// one thread per (b, j): three passes down the rows i of C[i][b][j] (stride B n: coalesced across the lanes)
__global__ __launch_bounds__(256) void att_softmax_rows_kernel(const float* __restrict__ C, int B, int n, float* __restrict__ S) {
  const int64_t bj = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t BN = (int64_t)B * n;
  if (bj >= BN) return;
  const int b = (int)(bj / n), j = (int)(bj % n);
  float mx = -INFINITY;
  for (int i = 0; i < n; ++i) mx = fmaxf(mx, C[(int64_t)i * BN + bj]);
  float sum = 0.f;
  for (int i = 0; i < n; ++i) sum += expf(C[(int64_t)i * BN + bj] - mx);
  const float inv = 1.f / sum;
  float* out = S + (int64_t)b * n * n + j;
  for (int i = 0; i < n; ++i) out[(int64_t)i * n] = expf(C[(int64_t)i * BN + bj] - mx) * inv;
}

__global__ __launch_bounds__(256) void att_softmax_rows_bwd_kernel(const float* __restrict__ S, const float* __restrict__ dS,
                                                                   int B, int n, float* __restrict__ dC) {
  const int64_t bj = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t BN = (int64_t)B * n;
  if (bj >= BN) return;
  const int b = (int)(bj / n), j = (int)(bj % n);
  const float* s = S + (int64_t)b * n * n + j;
  const float* g = dS + (int64_t)b * n * n + j;
  float dot = 0.f;
  for (int i = 0; i < n; ++i) dot = fmaf(g[(int64_t)i * n], s[(int64_t)i * n], dot);
  for (int i = 0; i < n; ++i) dC[(int64_t)i * BN + bj] = s[(int64_t)i * n] * (g[(int64_t)i * n] - dot);
}

// one thread per (i, j), loop over the batch: dP[b][i][j] = dsig[i][b][j] sig (1 - sig); dbias[i][j] = sum_b dP
__global__ __launch_bounds__(256) void att_sigmoid_bwd_kernel(const float* __restrict__ sig, const float* __restrict__ dsig,
                                                              int B, int n, float* __restrict__ dP, float* __restrict__ dbias) {
  const int64_t ij = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ij >= (int64_t)n * n) return;
  const int i = (int)(ij / n), j = (int)(ij % n);
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const int64_t e = ((int64_t)i * B + b) * n + j;
    const float s = sig[e];
    const float d = dsig[e] * s * (1.f - s);
    dP[((int64_t)b * n + i) * n + j] = d;
    acc += d;
  }
  if (dbias != nullptr) dbias[ij] = acc;
}

// ---- strided batched product of SMALL matrices: C[b, i, j] (+)= sum_k A[b, i, k] B[b, k, j], every operand through
// three strides (batch stride 0 = one matrix shared by the batch, swapped strides = a transposed view).  The embeddings
// around ASTGCN's attention are products with 1 .. 64 columns or rows (X W1, (X W1) W2, W3 X, X E; astgcn.py:252-256,
// :318-322, :437 and their adjoints): too small and too oddly shaped for the tile kernels of gemm.hip, and a library
// batched GEMM per product is what the reference pays for.  16 x 16 output tile per 256-thread workgroup, the K loop in
// 16-deep LDS stages, plain fmaf chains in k order (deterministic).
struct BmmArgs {
  const float* A; int64_t sab, sai, sak;
  const float* B; int64_t sbb, sbk, sbj;
  float* C; int64_t scb, sci, scj;
  int M, N, K, accumulate;
  int nb, ksplit, kper;      // batches; K cut into `ksplit` ranges of `kper` (a multiple of 16): partial sums meet in C by atomics
};
// grid: x = row tiles (up to 2^31), y = column tiles, z = (batch chunk, K range); a workgroup walks the batches b = z / ksplit,
// + gridDim.z / ksplit, ... (more than 65 535 batches) and owns one K range.
__global__ __launch_bounds__(256) void bmm_small_kernel(BmmArgs g) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  const int ks = blockIdx.z % g.ksplit, bstep = gridDim.z / g.ksplit;
  const int kbeg = ks * g.kper, kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
  for (int b = blockIdx.z / g.ksplit; b < g.nb; b += bstep) {
    const float* A = g.A + (int64_t)b * g.sab;
    const float* Bm = g.B + (int64_t)b * g.sbb;
    float acc = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
      const int ia = i0 + ty, ka = k0 + tx;
      sa[ty][tx] = (ia < g.M && ka < kend) ? A[(int64_t)ia * g.sai + (int64_t)ka * g.sak] : 0.f;
      const int kb = k0 + ty, jb = j0 + tx;
      sb[ty][tx] = (kb < kend && jb < g.N) ? Bm[(int64_t)kb * g.sbk + (int64_t)jb * g.sbj] : 0.f;
      __syncthreads();
      const int kk = kend - k0 < 16 ? kend - k0 : 16;     // (the zero padding must not enter the chain: 0 * inf = nan)
      for (int t = 0; t < kk; ++t) acc = fmaf(sa[ty][t], sb[t][tx], acc);
      __syncthreads();
    }
    const int i = i0 + ty, j = j0 + tx;
    if (i < g.M && j < g.N) {
      float* c = g.C + (int64_t)b * g.scb + (int64_t)i * g.sci + (int64_t)j * g.scj;
      if (g.ksplit > 1) atomicAdd(c, acc);                // C was zeroed (or holds the addend) before the launch
      else *c = g.accumulate ? *c + acc : acc;
    }
  }
}
// C = 0 through its strides (before a K-split product that does not accumulate)
__global__ __launch_bounds__(256) void bmm_zero_kernel(BmmArgs g) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)g.M * g.N;
  if (e >= per * g.nb) return;
  const int64_t b = e / per, r = e - b * per;
  g.C[b * g.scb + (r / g.N) * g.sci + (r % g.N) * g.scj] = 0.f;
}

// ---- Y = LayerNorm(relu(Z)) over the C columns of every row, and its adjoint (ASTGCN block tail, astgcn.py:476-478:
// `self._layer_norm(F.relu(X + X_hat))` — Z is the sum the two convolutions left in one buffer).  One 64-lane wavefront per
// row group: a row of C <= 1024 floats sits in registers (C / 64 per lane), mean / variance by shuffles.  eps as
// torch.nn.LayerNorm (biased variance).  Rows are picked by a two-level map (row r of Y <- row (r / rp) * hi + (r % rp) * lo
// of Z) so that a strided time convolution reads its own outputs and skips the padding rows.
template <int CPL>
__global__ __launch_bounds__(256) void relu_layernorm_kernel(const float* __restrict__ Z, int64_t rp, int64_t hi, int64_t lo,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int64_t rows, int C, float* __restrict__ Y,
                                                             float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* z = Z + ((r / rp) * hi + (r % rp) * lo) * C;
  float v[CPL];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    v[q] = c < C ? fmaxf(z[c], 0.f) : 0.f;
    sum += v[q];
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    const float d = c < C ? v[q] - mean : 0.f;
    var = fmaf(d, d, var);
  }
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = 1.f / sqrtf(var / (float)C + eps);
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = lane + 64 * q;
    if (c < C) Y[r * C + c] = (v[q] - mean) * rstd * gamma[c] + beta[c];
  }
  if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
}

// adjoint: dZ (same row map as Z; rows the map does not reach stay untouched: the caller zeroes the buffer).  A workgroup walks
// rows r = 4 blockIdx.x + wave, + 4 gridDim.x, ...: every lane keeps the dgamma / dbeta sums of its columns in registers over all
// of its rows, the four wavefronts meet in LDS and ONE atomic per column and workgroup goes out (round 3 issued one per
// (row, column): rows * C contended atomics onto C addresses).
template <int CPL>
__global__ __launch_bounds__(256) void relu_layernorm_bwd_kernel(const float* __restrict__ Z, int64_t rp, int64_t hi, int64_t lo,
                                                                 const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                 const float* __restrict__ dY, int64_t rows, int C,
                                                                 float* __restrict__ dZ, float* dgamma, float* dbeta) {
  __shared__ float red[2][4][64 * CPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[CPL], ab[CPL], gm[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    ag[q] = ab[q] = 0.f;
    gm[q] = lane + 64 * q < C ? gamma[lane + 64 * q] : 0.f;
  }
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t zr = ((r / rp) * hi + (r % rp) * lo) * C;
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float zv[CPL], xh[CPL], g[CPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = lane + 64 * q;
      zv[q] = c < C ? Z[zr + c] : 0.f;
      const float a = fmaxf(zv[q], 0.f);
      xh[q] = c < C ? (a - mean) * rstd : 0.f;
      const float dy = c < C ? dY[r * C + c] : 0.f;
      g[q] = dy * gm[q];
      s1 += g[q];
      s2 = fmaf(g[q], xh[q], s2);
      ag[q] = fmaf(dy, xh[q], ag[q]);
      ab[q] += dy;
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float inv = 1.f / (float)C;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = lane + 64 * q;
      if (c < C) {
        const float da = rstd * (g[q] - inv * s1 - xh[q] * inv * s2);
        dZ[zr + c] = zv[q] > 0.f ? da : 0.f;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q) { red[0][wave][lane + 64 * q] = ag[q]; red[1][wave][lane + 64 * q] = ab[q]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(&dgamma[c], ((red[0][0][c] + red[0][1][c]) + red[0][2][c]) + red[0][3][c]);
    atomicAdd(&dbeta[c], ((red[1][0][c] + red[1][1][c]) + red[1][2][c]) + red[1][3][c]);
  }
}

inline int grid1d(int64_t total, const char* what, dim3* grid) {
  const int64_t nb = pgt_cdiv(total, 256);
  if (nb >= ((int64_t)1 << 31)) {
    pgt_set_error("%s: grid too large", what);
    return PGT_ERR_INVALID;
  }
  *grid = dim3((unsigned)nb);
  return PGT_OK;
}

}  // namespace

extern "C" int pgt_att_sigmoid_scores_f32(const float* L, const float* R, const float* bias, int64_t B, int64_t n, int64_t m,
                                          float* sig, pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && n >= 0 && m >= 0, "pgt_att_sigmoid_scores_f32: negative size");
  if (B == 0 || n == 0) return PGT_OK;
  PGT_REQUIRE(L && R && bias && sig, "pgt_att_sigmoid_scores_f32: null pointer");
  PGT_REQUIRE(B < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) && m < ((int64_t)1 << 31), "pgt_att_sigmoid_scores_f32: size exceeds int32");
  dim3 grid;
  if (int e = grid1d(n * B * n, "pgt_att_sigmoid_scores_f32", &grid)) return e;
  PGT_LAUNCH(att_sigmoid_scores_kernel, grid, dim3(256), stream, L, R, bias, (int)B, (int)n, (int)m, sig);
  return pgt_check_launch("pgt_att_sigmoid_scores_f32");
}

extern "C" int pgt_att_softmax_rows_f32(const float* C, int64_t B, int64_t n, float* S, pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && n >= 0, "pgt_att_softmax_rows_f32: negative size");
  if (B == 0 || n == 0) return PGT_OK;
  PGT_REQUIRE(C && S, "pgt_att_softmax_rows_f32: null pointer");
  PGT_REQUIRE(B < ((int64_t)1 << 31) && n < ((int64_t)1 << 31), "pgt_att_softmax_rows_f32: size exceeds int32");
  dim3 grid;
  if (int e = grid1d(B * n, "pgt_att_softmax_rows_f32", &grid)) return e;
  PGT_LAUNCH(att_softmax_rows_kernel, grid, dim3(256), stream, C, (int)B, (int)n, S);
  return pgt_check_launch("pgt_att_softmax_rows_f32");
}

extern "C" int pgt_att_softmax_rows_bwd_f32(const float* S, const float* dS, int64_t B, int64_t n, float* dC,
                                            pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && n >= 0, "pgt_att_softmax_rows_bwd_f32: negative size");
  if (B == 0 || n == 0) return PGT_OK;
  PGT_REQUIRE(S && dS && dC, "pgt_att_softmax_rows_bwd_f32: null pointer");
  PGT_REQUIRE(B < ((int64_t)1 << 31) && n < ((int64_t)1 << 31), "pgt_att_softmax_rows_bwd_f32: size exceeds int32");
  dim3 grid;
  if (int e = grid1d(B * n, "pgt_att_softmax_rows_bwd_f32", &grid)) return e;
  PGT_LAUNCH(att_softmax_rows_bwd_kernel, grid, dim3(256), stream, S, dS, (int)B, (int)n, dC);
  return pgt_check_launch("pgt_att_softmax_rows_bwd_f32");
}

extern "C" int pgt_att_sigmoid_bwd_f32(const float* sig, const float* dsig, int64_t B, int64_t n, float* dP, float* dbias,
                                       pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && n >= 0, "pgt_att_sigmoid_bwd_f32: negative size");
  if (B == 0 || n == 0) return PGT_OK;
  PGT_REQUIRE(sig && dsig && dP, "pgt_att_sigmoid_bwd_f32: null pointer");
  PGT_REQUIRE(B < ((int64_t)1 << 31) && n < ((int64_t)1 << 31), "pgt_att_sigmoid_bwd_f32: size exceeds int32");
  dim3 grid;
  if (int e = grid1d(n * n, "pgt_att_sigmoid_bwd_f32", &grid)) return e;
  PGT_LAUNCH(att_sigmoid_bwd_kernel, grid, dim3(256), stream, sig, dsig, (int)B, (int)n, dP, dbias);
  return pgt_check_launch("pgt_att_sigmoid_bwd_f32");
}

extern "C" int pgt_bmm_f32(const float* A, int64_t sab, int64_t sai, int64_t sak, const float* B, int64_t sbb, int64_t sbk,
                           int64_t sbj, float* C, int64_t scb, int64_t sci, int64_t scj, int64_t nb, int64_t M, int64_t N,
                           int64_t K, int accumulate, pgt_stream_t stream) {
  PGT_REQUIRE(nb >= 0 && M >= 0 && N >= 0 && K >= 0, "pgt_bmm_f32: negative size");
  if (nb == 0 || M == 0 || N == 0) return PGT_OK;
  PGT_REQUIRE(C != nullptr && (K == 0 || (A && B)), "pgt_bmm_f32: null pointer");
  PGT_REQUIRE(M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31) && K < ((int64_t)1 << 31), "pgt_bmm_f32: size exceeds int32");
  const int64_t tx = pgt_cdiv(M, 16), ty = pgt_cdiv(N, 16);
  PGT_REQUIRE(tx < ((int64_t)1 << 31) && ty <= 65535 && nb < ((int64_t)1 << 31), "pgt_bmm_f32: more than 2^31 row tiles / batches or 65 535 column tiles");
  // a tall contraction into a handful of tiles (the adjoint of a batch-shared embedding: K = B N F, one 16 x 16 tile) would be one
  // workgroup walking the whole K: cut K so that the launch fills the chip, the ranges meet in C by fp32 atomics
  int64_t ksplit = 1;
  const int64_t tiles = tx * ty * nb;
  if (K >= 4096 && tiles < 512) {
    ksplit = pgt_cdiv(1024, tiles);
    if (ksplit > pgt_cdiv(K, 512)) ksplit = pgt_cdiv(K, 512);
    if (ksplit > 4096) ksplit = 4096;
  }
  int64_t kper = pgt_cdiv(pgt_cdiv(K, ksplit), 16) * 16;
  if (kper < 16) kper = 16;
  ksplit = K > 0 ? pgt_cdiv(K, kper) : 1;
  int64_t bz = nb;
  if (bz * ksplit > 65535) bz = 65535 / ksplit;
  BmmArgs g{A, sab, sai, sak, B, sbb, sbk, sbj, C, scb, sci, scj, (int)M, (int)N, (int)K, accumulate, (int)nb, (int)ksplit, (int)kper};
  if (ksplit > 1 && !accumulate) {
    const int64_t nz = pgt_cdiv(nb * M * N, 256);
    PGT_LAUNCH(bmm_zero_kernel, dim3((unsigned)nz), dim3(256), stream, g);
  }
  dim3 grid((unsigned)tx, (unsigned)ty, (unsigned)(bz * ksplit));
  PGT_LAUNCH(bmm_small_kernel, grid, dim3(256), stream, g);
  return pgt_check_launch("pgt_bmm_f32");
}

extern "C" int pgt_relu_layernorm_f32(const float* Z, int64_t row_period, int64_t stride_hi, int64_t stride_lo, const float* gamma,
                                      const float* beta, float eps, int64_t rows, int64_t C, float* Y, float* stats,
                                      pgt_stream_t stream) {
  PGT_REQUIRE(rows >= 0 && C >= 0, "pgt_relu_layernorm_f32: negative size");
  if (rows == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(Z && gamma && beta && Y && stats, "pgt_relu_layernorm_f32: null pointer");
  PGT_REQUIRE(C <= 1024 && row_period >= 1, "pgt_relu_layernorm_f32: at most 1024 columns, row_period >= 1");
  dim3 grid;
  if (int e = grid1d(rows * 64, "pgt_relu_layernorm_f32", &grid)) return e;
  const int cpl = (int)pgt_cdiv(C, 64);
#define PGT_LN_GO(Q_) PGT_LAUNCH((relu_layernorm_kernel<Q_>), grid, dim3(256), stream, Z, row_period, stride_hi, stride_lo, gamma, beta, eps, rows, (int)C, Y, stats)
  if (cpl == 1) PGT_LN_GO(1); else if (cpl == 2) PGT_LN_GO(2); else if (cpl <= 4) PGT_LN_GO(4); else if (cpl <= 8) PGT_LN_GO(8); else PGT_LN_GO(16);
#undef PGT_LN_GO
  return pgt_check_launch("pgt_relu_layernorm_f32");
}

extern "C" int pgt_relu_layernorm_bwd_f32(const float* Z, int64_t row_period, int64_t stride_hi, int64_t stride_lo,
                                          const float* gamma, const float* stats, const float* dY, int64_t rows, int64_t C,
                                          float* dZ, float* dgamma, float* dbeta, pgt_stream_t stream) {
  PGT_REQUIRE(rows >= 0 && C >= 0, "pgt_relu_layernorm_bwd_f32: negative size");
  if (rows == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(Z && gamma && stats && dY && dZ && dgamma && dbeta, "pgt_relu_layernorm_bwd_f32: null pointer");
  PGT_REQUIRE(C <= 1024 && row_period >= 1, "pgt_relu_layernorm_bwd_f32: at most 1024 columns, row_period >= 1");
  int64_t wgs = pgt_cdiv(rows, 4);
  if (wgs > 2048) wgs = 2048;                                  // <= 2048 atomics per column for the dgamma / dbeta sums
  dim3 grid((unsigned)wgs);
  const int cpl = (int)pgt_cdiv(C, 64);
#define PGT_LN_GO(Q_) PGT_LAUNCH((relu_layernorm_bwd_kernel<Q_>), grid, dim3(256), stream, Z, row_period, stride_hi, stride_lo, gamma, stats, dY, rows, (int)C, dZ, dgamma, dbeta)
  if (cpl == 1) PGT_LN_GO(1); else if (cpl == 2) PGT_LN_GO(2); else if (cpl <= 4) PGT_LN_GO(4); else if (cpl <= 8) PGT_LN_GO(8); else PGT_LN_GO(16);
#undef PGT_LN_GO
  return pgt_check_launch("pgt_relu_layernorm_bwd_f32");
}
