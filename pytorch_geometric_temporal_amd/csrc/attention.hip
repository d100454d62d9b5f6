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
