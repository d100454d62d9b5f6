// The per-node read-out the reference's models put behind their recurrent layer — `self.linear(F.relu(h))` with a
// torch.nn.Linear(hidden, 1 .. 4) (examples/indexBatching/tgcn/metr_la_main.py:43-45, examples/recurrent/dcrnn_example.py:27-31,
// examples/indexBatching/DCRNN/*_main.py) — as ONE streaming pass each way over the [rows, hidden] states:
//   relu_linear_fwd_kernel   y[m, n] = sum_k relu(x[m, k]) W[n, k] + b[n]
//   relu_linear_bwd_kernel   dX[m, k] = (x[m, k] > 0) sum_n dy[m, n] W[n, k]     (whole 128-byte rows per LPR lanes)
//                            dW[n, k] = sum_m dy[m, n] relu(x[m, k]),  db[n] = sum_m dy[m, n]   (per-workgroup partial sums,
//                            added in index order by relu_linear_reduce_kernel: deterministic, no float atomics)
// Unfused, the adjoint is three passes over the states (threshold_backward, dY W, X^T dY: 22 + 13 + 33 us per call at 400 000 rows
// of 32 on config 4, rocprofv3 profiles/r05_tgcn50k_bench_kernel_stats.csv); here x is read once and dX written once.
// LPR = lanes per row: hidden / 4 rounded up to 8 or 16 (a lane holds a float4 of its row), 64 / LPR rows per wavefront pass.
#include "pgt_common.h"

namespace {

struct RoArgs {
  const float* X; int64_t ldx; const float* W; const float* b;   // W [N][K] row-major (torch.nn.Linear.weight), b [N] | null
  float* Y; int64_t ldy;
  const float* dY; int64_t lddy; float* dX; int64_t lddx;          // dX null: no input gradient wanted
  float* part;                                                      // [n_wg][N K + N]
  int M, K, N, relu;                                                // relu = 0: a plain skinny Linear through the same kernels
};

constexpr int RO_U = 4;             // row groups in flight per wavefront pass

template <int LPR>
__global__ __launch_bounds__(256) void relu_linear_fwd_kernel(RoArgs g) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, q = lane % LPR, rg = lane / LPR;
  const int wave_id = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), n_waves = (int)gridDim.x * 4;
  const bool kv = 4 * q < g.K;
  float4 w[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
    w[n] = (kv && n < g.N) ? *reinterpret_cast<const float4*>(g.W + (int64_t)n * g.K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float bq = (g.b && q < g.N) ? g.b[q] : 0.f;
  for (int64_t base = (int64_t)wave_id * (RPW * RO_U); base < g.M; base += (int64_t)n_waves * (RPW * RO_U)) {
    float4 a[RO_U];
#pragma unroll
    for (int u = 0; u < RO_U; ++u) {
      int64_t row = base + RPW * u + rg;
      row = row < g.M ? row : g.M - 1;
      a[u] = kv ? *reinterpret_cast<const float4*>(g.X + row * g.ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < RO_U; ++u) {
      float4 r = a[u];
      if (g.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      float s[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        s[n] = fmaf(r.w, w[n].w, fmaf(r.z, w[n].z, fmaf(r.y, w[n].y, r.x * w[n].x)));
        if (n < g.N) {                                             // (uniform)
#pragma unroll
          for (int m = LPR / 2; m >= 1; m >>= 1) s[n] += __shfl_xor(s[n], m, LPR);
        }
      }
      const int64_t row = base + RPW * u + rg;
      if (row < g.M && q < g.N) g.Y[row * g.ldy + q] = (q == 0 ? s[0] : q == 1 ? s[1] : q == 2 ? s[2] : s[3]) + bq;
    }
  }
}

__device__ __forceinline__ int ro_part_floats(int K, int N) { return N * K + N; }

template <int LPR>
__global__ __launch_bounds__(256) void relu_linear_bwd_kernel(RoArgs g) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[4][4][LPR][4];          // [wave][n][q][i]
  __shared__ float bred[4][4];
  const int wave = (int)threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane % LPR, rg = lane / LPR;
  const int wave_id = (int)blockIdx.x * 4 + wave, n_waves = (int)gridDim.x * 4;
  const bool kv = 4 * q < g.K;
  float4 w[4], acc[4];
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    w[n] = (kv && n < g.N) ? *reinterpret_cast<const float4*>(g.W + (int64_t)n * g.K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t base = (int64_t)wave_id * (RPW * RO_U); base < g.M; base += (int64_t)n_waves * (RPW * RO_U)) {
    float4 a[RO_U];
    float gv[RO_U][4];
#pragma unroll
    for (int u = 0; u < RO_U; ++u) {
      const int64_t row = base + RPW * u + rg;
      const bool rv = row < g.M;
      const int64_t rc = rv ? row : g.M - 1;
      a[u] = kv ? *reinterpret_cast<const float4*>(g.X + rc * g.ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < 4; ++n) gv[u][n] = (rv && n < g.N) ? g.dY[rc * g.lddy + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RO_U; ++u) {
      const float4 x = a[u];
      float4 r = x;
      if (g.relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float gn = gv[u][n];
        d.x = fmaf(gn, w[n].x, d.x); d.y = fmaf(gn, w[n].y, d.y); d.z = fmaf(gn, w[n].z, d.z); d.w = fmaf(gn, w[n].w, d.w);
        acc[n].x = fmaf(r.x, gn, acc[n].x); acc[n].y = fmaf(r.y, gn, acc[n].y);
        acc[n].z = fmaf(r.z, gn, acc[n].z); acc[n].w = fmaf(r.w, gn, acc[n].w);
        if (q == 0) bs[n] += gn;
      }
      const int64_t row = base + RPW * u + rg;
      if (g.dX != nullptr && kv && row < g.M) {
        if (g.relu) {                                               // relu'(x) = [x > 0]  (threshold_backward)
          d.x = x.x > 0.f ? d.x : 0.f; d.y = x.y > 0.f ? d.y : 0.f; d.z = x.z > 0.f ? d.z : 0.f; d.w = x.w > 0.f ? d.w : 0.f;
        }
        *reinterpret_cast<float4*>(g.dX + row * g.lddx + 4 * q) = d;
      }
    }
  }
  // fold the RPW row groups of the wavefront (lanes q, q + LPR, ...), then the four wavefronts through LDS
#pragma unroll
  for (int n = 0; n < 4; ++n) {
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) {
      acc[n].x += __shfl_xor(acc[n].x, m); acc[n].y += __shfl_xor(acc[n].y, m);
      acc[n].z += __shfl_xor(acc[n].z, m); acc[n].w += __shfl_xor(acc[n].w, m);
      bs[n] += __shfl_xor(bs[n], m);
    }
  }
  if (rg == 0) {
#pragma unroll
    for (int n = 0; n < 4; ++n) { red[wave][n][q][0] = acc[n].x; red[wave][n][q][1] = acc[n].y; red[wave][n][q][2] = acc[n].z; red[wave][n][q][3] = acc[n].w; }
  }
  if (lane == 0)
#pragma unroll
    for (int n = 0; n < 4; ++n) bred[wave][n] = bs[n];
  __syncthreads();
  float* part = g.part + (int64_t)blockIdx.x * ro_part_floats(g.K, g.N);
  for (int e = threadIdx.x; e < 4 * LPR * 4; e += 256) {
    const int i = e & 3, qq = (e >> 2) % LPR, n = e / (4 * LPR), k = 4 * qq + i;
    if (n < g.N && k < g.K) part[n * g.K + k] = ((red[0][n][qq][i] + red[1][n][qq][i]) + red[2][n][qq][i]) + red[3][n][qq][i];
  }
  if ((int)threadIdx.x < g.N) part[g.N * g.K + threadIdx.x] = ((bred[0][threadIdx.x] + bred[1][threadIdx.x]) + bred[2][threadIdx.x]) + bred[3][threadIdx.x];
}

// out[e] = sum over the workgroups' partials in a FIXED order: a workgroup of sixteen wavefronts owns 64 elements, every wavefront adds
// a contiguous sixteenth of the partials (eight loads in flight per lane), the sixteenths meet in LDS in index order.  (Four
// wavefronts per workgroup walked 256 partials each: 13.5 us per call, as long as the forward kernel.)
__global__ __launch_bounds__(1024) void relu_linear_reduce_kernel(const float* __restrict__ part, int n_wg, int n, int NK, float* __restrict__ dW,
                                                                 float* __restrict__ db) {
  __shared__ float red[16][64];        // sixteen wavefronts, a contiguous sixteenth of the partials each
  const int lane = threadIdx.x & 63, qd = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int per = (n_wg + 15) / 16, w0 = qd * per, w1 = (w0 + per < n_wg) ? w0 + per : n_wg;
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
  red[qd][lane] = acc;
  __syncthreads();
  if (qd != 0 || e >= n) return;
  acc = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) acc += red[w][lane];          // in index order: the same sum launch after launch
  if (e < NK) { if (dW) dW[e] = acc; }
  else if (db) db[e - NK] = acc;
}

#ifdef PGT_EMU
constexpr int RO_WGS = 3;
#else
constexpr int RO_WGS = 1024;
#endif

bool ro_shape_ok(int64_t K, int64_t N) { return K >= 4 && K <= 64 && K % 4 == 0 && N >= 1 && N <= 4; }

}  // namespace

extern "C" int pgt_relu_linear_fits(int64_t K, int64_t N) { return ro_shape_ok(K, N) ? 1 : 0; }

extern "C" int64_t pgt_relu_linear_bwd_ws_floats(int64_t K, int64_t N) {
  return ro_shape_ok(K, N) ? (int64_t)RO_WGS * (N * K + N) : 0;
}

extern "C" int pgt_relu_linear_f32(const float* X, int64_t ldx, const float* W, const float* b, int64_t M, int64_t K, int64_t N,
                                   int relu, float* Y, int64_t ldy, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0, "pgt_relu_linear_f32: negative size");
  PGT_REQUIRE(ro_shape_ok(K, N), "pgt_relu_linear_f32: built for 4 <= in <= 64 (a multiple of 4) and 1 .. 4 outputs (got %lld, %lld)",
              (long long)K, (long long)N);
  if (M == 0) return PGT_OK;
  PGT_REQUIRE(X && W && Y, "pgt_relu_linear_f32: null pointer");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - 256 && ldx >= K && ldx % 4 == 0 && ldy >= N && pgt_aligned(X, 16) && pgt_aligned(W, 16),
              "pgt_relu_linear_f32: rows must be 16-byte addressable (extent out of range or misaligned)");
  RoArgs g{};
  g.X = X; g.ldx = ldx; g.W = W; g.b = b; g.Y = Y; g.ldy = ldy; g.M = (int)M; g.K = (int)K; g.N = (int)N; g.relu = relu ? 1 : 0;
  const int lpr = K <= 32 ? 8 : 16;
  const int64_t need = pgt_cdiv(M, (int64_t)(64 / lpr) * RO_U * 4);
  const int64_t wgs = need < 4 * RO_WGS ? need : 4 * RO_WGS;
  if (lpr == 8) PGT_LAUNCH((relu_linear_fwd_kernel<8>), dim3((unsigned)wgs), dim3(256), stream, g);
  else PGT_LAUNCH((relu_linear_fwd_kernel<16>), dim3((unsigned)wgs), dim3(256), stream, g);
  return pgt_check_launch("pgt_relu_linear_f32");
}

extern "C" int pgt_relu_linear_bwd_f32(const float* X, int64_t ldx, const float* dY, int64_t lddy, const float* W, int64_t M, int64_t K,
                                       int64_t N, int relu, float* dX, int64_t lddx, float* dW, float* db, float* ws, int64_t ws_floats,
                                       pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0, "pgt_relu_linear_bwd_f32: negative size");
  PGT_REQUIRE(ro_shape_ok(K, N), "pgt_relu_linear_bwd_f32: built for 4 <= in <= 64 (a multiple of 4) and 1 .. 4 outputs");
  if (M == 0) {
    if ((dW && hipMemsetAsync(dW, 0, (size_t)N * K * 4, (hipStream_t)stream) != hipSuccess) ||
        (db && hipMemsetAsync(db, 0, (size_t)N * 4, (hipStream_t)stream) != hipSuccess)) {
      pgt_set_error("pgt_relu_linear_bwd_f32: memset failed");
      return PGT_ERR_LAUNCH;
    }
    return PGT_OK;
  }
  PGT_REQUIRE(X && dY && W && ws, "pgt_relu_linear_bwd_f32: null pointer");
  PGT_REQUIRE(ws_floats >= pgt_relu_linear_bwd_ws_floats(K, N), "pgt_relu_linear_bwd_f32: scratch too small (pgt_relu_linear_bwd_ws_floats)");
  PGT_REQUIRE(M < ((int64_t)1 << 31) - 256 && ldx >= K && ldx % 4 == 0 && lddy >= N && pgt_aligned(X, 16) && pgt_aligned(W, 16) &&
                  (dX == nullptr || (lddx >= K && lddx % 4 == 0 && pgt_aligned(dX, 16))),
              "pgt_relu_linear_bwd_f32: rows must be 16-byte addressable (extent out of range or misaligned)");
  RoArgs g{};
  g.X = X; g.ldx = ldx; g.W = W; g.dY = dY; g.lddy = lddy; g.dX = dX; g.lddx = lddx; g.part = ws;
  g.M = (int)M; g.K = (int)K; g.N = (int)N; g.relu = relu ? 1 : 0;
  const int lpr = K <= 32 ? 8 : 16;
  const int64_t need = pgt_cdiv(M, (int64_t)(64 / lpr) * RO_U * 4);
  const int wgs = (int)(need < RO_WGS ? need : RO_WGS);
  if (lpr == 8) PGT_LAUNCH((relu_linear_bwd_kernel<8>), dim3((unsigned)wgs), dim3(256), stream, g);
  else PGT_LAUNCH((relu_linear_bwd_kernel<16>), dim3((unsigned)wgs), dim3(256), stream, g);
  const int n = (int)(N * K + N);
  PGT_LAUNCH(relu_linear_reduce_kernel, dim3((unsigned)pgt_cdiv(n, 64)), dim3(1024), stream, ws, wgs, n, (int)(N * K), dW, db);
  return pgt_check_launch("pgt_relu_linear_bwd_f32");
}
