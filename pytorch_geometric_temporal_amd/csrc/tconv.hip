// The dense halves of an ST-Conv block (torch_geometric_temporal/nn/attention/stgcn.py):
//
//  tconv_glu_kernel      TemporalConv.forward (stgcn.py:27-44): three Conv2d(Cin -> Cout, (1, k)) over the time axis and the
//      gate H = relu(P * sigmoid(Q) + R), on the reference's OWN layout [B, T, N, Cin] -> [B, T - k + 1, N, Cout]: no
//      permute to channels-first and back, no im2col.  Row m = (b T + t) N + n of X holds the Cin channels of one
//      (sample, step, node); tap dt of a (1, k) convolution over time is the same matrix shifted by dt N rows, so the three
//      convolutions are ONE product [rows, k Cin] x [k Cin, 3 Cout] on v_mfma_f32_32x32x2_f32 whose K-segment j starts j N
//      rows further down.  P, Q and R of one (row, channel) land in the SAME lane of three accumulators (the D map does not
//      depend on the column block), so the gate is computed in registers and only H — plus P and sigmoid(Q) when a
//      backward pass will want them — is written.  Tiles are cut in OUTPUT rows (b, t < T', n): the k - 1 steps at the end
//      of every sample that have no output cost nothing.
//  tconv_glu_bwd_kernel  the gate's adjoint, streaming: dR = dH [H > 0], dP = dR S, dQ = dR P S (1 - S) written as the
//      [rows, 3 Cout] operand of the two gradient products (weight gradient on pgt_gemm_tn_acc_f32, input gradient on
//      pgt_gemm_f32: the same shifted segments), in INPUT row numbering with zero rows where a step has no output and
//      (k - 1) N zero rows in front, so that the input gradient dX[m] = sum_dt dZ[m - dt N] W_dt^T is one K-segmented product.
//  bn_nodes_kernel / bn_nodes_bwd_kernel   BatchNorm2d(num_nodes) of STConv (stgcn.py:129, :156-159: the reference permutes
//      to [B, N, T, C] so that the NODE is the normalised "channel"): one workgroup per node walks the node's B T' rows of C
//      floats in place in [B, T', N, C] (no permute), three passes (mean, centred second moment, normalise: the node's
//      data is read from HBM once and then from L2), deterministic sums, running statistics updated in the same launch.
#include "pgt_common.h"

namespace {

struct TconvArgs {
  const float* X; int64_t ldx;
  const float* Wp; const float* bias;
  float* H; float* P; float* S;
  int Cin, Cout, k, Ktot;
  int64_t N, TN, TpN, Mout;
};

constexpr int TBM = 128, TBK = 32;

// input row (b T + t) N + n of output row r = (b T' + t) N + n
__device__ __forceinline__ int64_t tconv_in_row(int64_t r, int64_t TN, int64_t TpN) {
  const int64_t b = r / TpN;
  return b * TN + (r - b * TpN);
}

// NCB: 32-channel column blocks per workgroup (1 | 2).  VECA: A is fetched as float4 along k (Cin % 4 == 0, 16-byte aligned
// rows), else one float per (row, k) with the lanes along the ROWS (Cin = 1, 2: adjacent lanes read adjacent samples).
template <int NCB, bool VECA>
__global__ __launch_bounds__(256) void tconv_glu_kernel(TconvArgs g) {
  constexpr int BN = 32 * NCB;
  struct Tiles { float As[TBK][TBM + 1]; float Bs[TBK][3][BN + 1]; };
  __shared__ Tiles tl;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lo = lane & 31, hi = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * TBM;
  const int c0 = (int)blockIdx.y * BN;

  pgt_f32x16 acc[3][NCB];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;

  // ---- A staging: 128 rows x 32 k = 16 floats per thread
  float ra[16];
  // VECA: thread = (k quad a_kq, row a_m + 32 p); else thread = (row tid & 127, k = (tid >> 7) + 2 i)
  const int a_kq = tid & 7, a_m = tid >> 3;
  int64_t in_off[VECA ? 4 : 1];
  bool in_ok[VECA ? 4 : 1];
  if constexpr (VECA) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = m0 + a_m + 32 * p;
      in_ok[p] = r < g.Mout;
      in_off[p] = tconv_in_row(in_ok[p] ? r : 0, g.TN, g.TpN) * g.ldx;
    }
  } else {
    const int64_t r = m0 + (tid & 127);
    in_ok[0] = r < g.Mout;
    in_off[0] = tconv_in_row(in_ok[0] ? r : 0, g.TN, g.TpN) * g.ldx;
  }
  const int64_t tap = g.N * g.ldx;

  // ---- B staging: 32 k x 3 gates x BN columns
  constexpr int B_KPP = 256 / BN;              // k rows per pass
  constexpr int B_PASSES = TBK / B_KPP;
  float rb[3][B_PASSES];
  const int b_n = tid % BN, b_k = tid / BN;
  const bool b_cv = c0 + b_n < g.Cout;
  const int ldw = 3 * g.Cout;

  auto load_tile = [&](int k0) {
    if constexpr (VECA) {
      const int kg = k0 + 4 * a_kq;
      const bool kv = kg < g.Ktot;
      const int j = kv ? kg / g.Cin : 0;
      const int64_t koff = (int64_t)j * tap + (kg - j * g.Cin);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kv && in_ok[p]) t = *reinterpret_cast<const float4*>(g.X + in_off[p] + koff);
        ra[4 * p] = t.x; ra[4 * p + 1] = t.y; ra[4 * p + 2] = t.z; ra[4 * p + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kg = k0 + (tid >> 7) + 2 * i;
        float v = 0.f;
        if (kg < g.Ktot && in_ok[0]) {
          const int j = kg / g.Cin;
          v = g.X[in_off[0] + (int64_t)j * tap + (kg - j * g.Cin)];
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) {
        const int kg = k0 + b_k + p * B_KPP;
        rb[q][p] = (b_cv && kg < g.Ktot) ? g.Wp[(int64_t)kg * ldw + q * g.Cout + c0 + b_n] : 0.f;
      }
  };
  auto store_tile = [&]() {
    if constexpr (VECA) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) tl.As[4 * a_kq + e][a_m + 32 * p] = ra[4 * p + e];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) tl.As[(tid >> 7) + 2 * i][tid & 127] = ra[i];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) tl.Bs[b_k + p * B_KPP][q][b_n] = rb[q][p];
  };

  load_tile(0);
  for (int k0 = 0; k0 < g.Ktot; k0 += TBK) {
    store_tile();
    __syncthreads();
    if (k0 + TBK < g.Ktot) load_tile(k0 + TBK);      // in flight while the matrix cores work on this tile
    const int kend = g.Ktot - k0 < TBK ? g.Ktot - k0 : TBK;
#pragma unroll 4
    for (int kk = 0; kk < kend; kk += 2) {            // an odd Ktot: the last pair's second k is a zero row of both tiles
      const float a = tl.As[kk + hi][wave * 32 + lo];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[q][j] = PGT_MFMA_32x32x2(a, tl.Bs[kk + hi][q][j * 32 + lo], acc[q][j]);
    }
    __syncthreads();
  }

  // ---- gate in registers: the three accumulators of a lane hold P, Q, R of the same (row, channel)
#pragma unroll
  for (int j = 0; j < NCB; ++j) {
    const int c = c0 + j * 32 + lo;
    if (c >= g.Cout) continue;
    float bp = 0.f, bq = 0.f, br = 0.f;
    if (g.bias) { bp = g.bias[c]; bq = g.bias[g.Cout + c]; br = g.bias[2 * g.Cout + c]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < g.Mout) {
        const float p = acc[0][j][r] + bp;
        const float s = pgt_sigmoidf(acc[1][j][r] + bq);
        const float h = fmaxf(pgt_add_rn(pgt_mul_rn(p, s), acc[2][j][r] + br), 0.f);   // PQ rounded, then + R (stgcn.py:41-42)
        const int64_t o = row * g.Cout + c;
        g.H[o] = h;
        if (g.P) { g.P[o] = p; g.S[o] = s; }
      }
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void tconv_glu_bwd_kernel(const float* __restrict__ dH, const float* __restrict__ H,
                                                             const float* __restrict__ P, const float* __restrict__ S,
                                                             int64_t pad_rows, int64_t rows, int64_t TN, int64_t TpN, int Cout,
                                                             float* __restrict__ dZ) {
  const int CV = Cout / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * CV) return;
  const int64_t row = idx / CV;
  const int c = (int)(idx - row * CV) * V;
  float dp[V], dq[V], dr[V];
#pragma unroll
  for (int i = 0; i < V; ++i) dp[i] = dq[i] = dr[i] = 0.f;
  const int64_t m = row - pad_rows;
  if (m >= 0) {
    const int64_t b = m / TN, rem = m - b * TN;
    if (rem < TpN) {
      const int64_t o = (b * TpN + rem) * Cout + c;
      float g[V], h[V], p[V], s[V];
      pgt_ldv<V>(dH + o, g); pgt_ldv<V>(H + o, h); pgt_ldv<V>(P + o, p); pgt_ldv<V>(S + o, s);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        dr[i] = h[i] > 0.f ? g[i] : 0.f;
        dp[i] = dr[i] * s[i];
        dq[i] = dr[i] * p[i] * (s[i] * (1.f - s[i]));
      }
    }
  }
  float* z = dZ + row * 3 * Cout + c;
  pgt_stv<V>(z, dp); pgt_stv<V>(z + Cout, dq); pgt_stv<V>(z + 2 * Cout, dr);
}

// ---- sums over a 256-thread workgroup in a fixed order (shuffle tree inside a wavefront, the four wavefronts in index order)
__device__ __forceinline__ float block_sum256(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();                       // the previous reduction's readers are done with red[]
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

template <int V>
__global__ __launch_bounds__(256) void bn_nodes_kernel(const float* __restrict__ X, int64_t R, int64_t N, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* running_mean, float* running_var, float momentum, float eps,
                                                        int training, float* __restrict__ Y, float* __restrict__ stats) {
  __shared__ float red[4];
  const int64_t n = blockIdx.x;
  const int CV = C / V;
  const int64_t total = R * CV, row_stride = N * C;
  const float* x0 = X + n * C;
  float mean, invstd;
  if (training) {
    float s = 0.f;
#pragma unroll 4
  for (int64_t e = threadIdx.x; e < total; e += 256) {
      const int64_t r = e / CV;
      float v[V];
      pgt_ldv<V>(x0 + r * row_stride + (e - r * CV) * V, v);
#pragma unroll
      for (int i = 0; i < V; ++i) s += v[i];
    }
    const float cnt = (float)(R * C);
    mean = block_sum256(s, red) / cnt;
    float q = 0.f;
#pragma unroll 4
  for (int64_t e = threadIdx.x; e < total; e += 256) {
      const int64_t r = e / CV;
      float v[V];
      pgt_ldv<V>(x0 + r * row_stride + (e - r * CV) * V, v);
#pragma unroll
      for (int i = 0; i < V; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    }
    const float ssq = block_sum256(q, red);
    const float var = ssq / cnt;
    invstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
      if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
      if (running_var) running_var[n] = (1.f - momentum) * running_var[n] + momentum * (ssq / (cnt - 1.f));
    }
  } else {
    mean = running_mean[n];
    invstd = 1.f / sqrtf(running_var[n] + eps);
  }
  if (threadIdx.x == 0 && stats) { stats[2 * n] = mean; stats[2 * n + 1] = invstd; }
  const float w = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
  float* y0 = Y + n * C;
#pragma unroll 4
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int64_t r = e / CV;
    const int64_t o = r * row_stride + (e - r * CV) * V;
    float v[V];
    pgt_ldv<V>(x0 + o, v);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = (v[i] - mean) * invstd * w + b;
    pgt_stv<V>(y0 + o, v);
  }
}

template <int V>
__global__ __launch_bounds__(256) void bn_nodes_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            int64_t R, int64_t N, int C, int training, float* __restrict__ dX,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[4];
  const int64_t n = blockIdx.x;
  const int CV = C / V;
  const int64_t total = R * CV, row_stride = N * C;
  const float mean = stats[2 * n], invstd = stats[2 * n + 1];
  const float* x0 = X + n * C;
  const float* g0 = dY + n * C;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int64_t r = e / CV;
    const int64_t o = r * row_stride + (e - r * CV) * V;
    float v[V], d[V];
    pgt_ldv<V>(x0 + o, v); pgt_ldv<V>(g0 + o, d);
#pragma unroll
    for (int i = 0; i < V; ++i) { s1 += d[i]; s2 = fmaf(d[i], (v[i] - mean) * invstd, s2); }
  }
  s1 = block_sum256(s1, red);
  s2 = block_sum256(s2, red);
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[n] = s2;
    if (dbeta) dbeta[n] = s1;
  }
  if (!dX) return;
  const float w = gamma ? gamma[n] : 1.f;
  const float inv = 1.f / (float)(R * C);
  const float k1 = training ? s1 * inv : 0.f, k2 = training ? s2 * inv : 0.f;
  float* d0 = dX + n * C;
#pragma unroll 4
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int64_t r = e / CV;
    const int64_t o = r * row_stride + (e - r * CV) * V;
    float v[V], d[V];
    pgt_ldv<V>(x0 + o, v); pgt_ldv<V>(g0 + o, d);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = w * invstd * (d[i] - k1 - (v[i] - mean) * invstd * k2);
    pgt_stv<V>(d0 + o, d);
  }
}

#define TCONV_VDISPATCH(v, KERN, grid, block, stream, ...)                      \
  do {                                                                          \
    if ((v) == 4) { PGT_LAUNCH(KERN<4>, grid, block, stream, __VA_ARGS__); }    \
    else if ((v) == 2) { PGT_LAUNCH(KERN<2>, grid, block, stream, __VA_ARGS__); } \
    else { PGT_LAUNCH(KERN<1>, grid, block, stream, __VA_ARGS__); }             \
  } while (0)

}  // namespace

extern "C" int pgt_tconv_glu_f32(const float* X, int64_t ldx, int64_t B, int64_t T, int64_t N, int64_t Cin, int64_t Cout,
                                 int64_t k, const float* Wp, const float* bias3, float* H, float* P, float* S,
                                 pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && T >= 0 && N >= 0 && Cin >= 1 && Cout >= 1 && k >= 1, "pgt_tconv_glu_f32: bad size");
  PGT_REQUIRE(k <= T || B * N == 0, "pgt_tconv_glu_f32: kernel size %lld exceeds the %lld time steps", (long long)k, (long long)T);
  const int64_t Tp = T - k + 1, Mout = B * Tp * N;
  if (Mout <= 0) return PGT_OK;
  PGT_REQUIRE(X && Wp && H, "pgt_tconv_glu_f32: null pointer");
  PGT_REQUIRE((P == nullptr) == (S == nullptr), "pgt_tconv_glu_f32: P and S go together");
  PGT_REQUIRE(ldx >= Cin, "pgt_tconv_glu_f32: row stride smaller than the row");
  PGT_REQUIRE(k * Cin < ((int64_t)1 << 30) && Cout < ((int64_t)1 << 28) && B * T * N * ldx < ((int64_t)1 << 62),
              "pgt_tconv_glu_f32: extent out of range");
  TconvArgs g;
  g.X = X; g.ldx = ldx; g.Wp = Wp; g.bias = bias3; g.H = H; g.P = P; g.S = S;
  g.Cin = (int)Cin; g.Cout = (int)Cout; g.k = (int)k; g.Ktot = (int)(k * Cin);
  g.N = N; g.TN = T * N; g.TpN = Tp * N; g.Mout = Mout;
  const int64_t gx = pgt_cdiv(Mout, TBM);
  PGT_REQUIRE(gx < ((int64_t)1 << 31), "pgt_tconv_glu_f32: grid too large");
  const bool veca = Cin % 4 == 0 && ldx % 4 == 0 && pgt_aligned(X, 16);
  const int ncb = Cout > 32 ? 2 : 1;
  const int64_t gy = pgt_cdiv(Cout, 32 * ncb);
  PGT_REQUIRE(gy <= 65535, "pgt_tconv_glu_f32: too many output channels");
  dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  if (ncb == 2) {
    if (veca) PGT_LAUNCH((tconv_glu_kernel<2, true>), grid, block, stream, g);
    else PGT_LAUNCH((tconv_glu_kernel<2, false>), grid, block, stream, g);
  } else {
    if (veca) PGT_LAUNCH((tconv_glu_kernel<1, true>), grid, block, stream, g);
    else PGT_LAUNCH((tconv_glu_kernel<1, false>), grid, block, stream, g);
  }
  return pgt_check_launch("pgt_tconv_glu_f32");
}

extern "C" int pgt_tconv_glu_bwd_f32(const float* dH, const float* H, const float* P, const float* S, int64_t B, int64_t T,
                                     int64_t N, int64_t Cout, int64_t k, float* dZ, pgt_stream_t stream) {
  PGT_REQUIRE(B >= 0 && T >= 0 && N >= 0 && Cout >= 1 && k >= 1 && k <= (T > 0 ? T : k), "pgt_tconv_glu_bwd_f32: bad size");
  const int64_t pad = (k - 1) * N, rows = pad + B * T * N;
  if (rows == 0) return PGT_OK;
  PGT_REQUIRE(dZ && (B * T * N == 0 || (dH && H && P && S)), "pgt_tconv_glu_bwd_f32: null pointer");
  PgtVecPick pick;
  pick.width(Cout);
  pick.operand(dH, Cout); pick.operand(H, Cout); pick.operand(P, Cout); pick.operand(S, Cout); pick.operand(dZ, 3 * Cout);
  const int64_t total = rows * (Cout / pick.v);
  const int64_t nb = pgt_cdiv(total, 256);
  PGT_REQUIRE(nb < ((int64_t)1 << 31), "pgt_tconv_glu_bwd_f32: grid too large");
  dim3 grid((unsigned)nb), block(256);
  TCONV_VDISPATCH(pick.v, tconv_glu_bwd_kernel, grid, block, stream, dH, H, P, S, pad, rows, T * N, (T - k + 1) * N, (int)Cout,
                  dZ);
  return pgt_check_launch("pgt_tconv_glu_bwd_f32");
}

extern "C" int pgt_batchnorm_nodes_f32(const float* X, int64_t R, int64_t N, int64_t C, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, int training,
                                       float* Y, float* stats, pgt_stream_t stream) {
  PGT_REQUIRE(R >= 0 && N >= 0 && C >= 0, "pgt_batchnorm_nodes_f32: negative size");
  if (R == 0 || N == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(X && Y, "pgt_batchnorm_nodes_f32: null pointer");
  PGT_REQUIRE(training || (running_mean && running_var), "pgt_batchnorm_nodes_f32: evaluation needs the running statistics");
  PGT_REQUIRE(!training || R * C > 1, "pgt_batchnorm_nodes_f32: more than one value per node is needed to train");
  PGT_REQUIRE(N < ((int64_t)1 << 31) && C < ((int64_t)1 << 30), "pgt_batchnorm_nodes_f32: extent out of range");
  PgtVecPick pick;
  pick.width(C);
  pick.operand(X, C); pick.operand(Y, C);
  dim3 grid((unsigned)N), block(256);
  TCONV_VDISPATCH(pick.v, bn_nodes_kernel, grid, block, stream, X, R, N, (int)C, gamma, beta, running_mean, running_var,
                  momentum, eps, training ? 1 : 0, Y, stats);
  return pgt_check_launch("pgt_batchnorm_nodes_f32");
}

extern "C" int pgt_batchnorm_nodes_bwd_f32(const float* dY, const float* X, const float* stats, const float* gamma, int64_t R,
                                           int64_t N, int64_t C, int training, float* dX, float* dgamma, float* dbeta,
                                           pgt_stream_t stream) {
  PGT_REQUIRE(R >= 0 && N >= 0 && C >= 0, "pgt_batchnorm_nodes_bwd_f32: negative size");
  if (R == 0 || N == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(dY && X && stats, "pgt_batchnorm_nodes_bwd_f32: null pointer");
  PGT_REQUIRE(N < ((int64_t)1 << 31) && C < ((int64_t)1 << 30), "pgt_batchnorm_nodes_bwd_f32: extent out of range");
  PgtVecPick pick;
  pick.width(C);
  pick.operand(X, C); pick.operand(dY, C); pick.operand(dX, C);
  dim3 grid((unsigned)N), block(256);
  TCONV_VDISPATCH(pick.v, bn_nodes_bwd_kernel, grid, block, stream, dY, X, stats, gamma, R, N, (int)C, training ? 1 : 0, dX,
                  dgamma, dbeta);
  return pgt_check_launch("pgt_batchnorm_nodes_bwd_f32");
}
