// EvolveGCN weight evolution in one launch (evolvegcnh.py:78-102, evolvegcno.py:170-191).
//
// Per snapshot the reference runs, on an 8 x 8 state: TopKPooling (score = tanh(X p / |p|), top-k by a sort, gather, scale:
// ~10 torch launches), torch.nn.GRU over one time step (a cuDNN / MIOpen call: a handful of launches for 6 products of
// 8 x 8 by 8 x 8) and their autograd twins — per-launch latency is ALL of its cost (England-Covid, 129 nodes: the eager GPU
// epoch ties the CPU's).  Here the whole chain is one workgroup:
//   H variant (pool != 0):  s_i = tanh(X_i . p / |p|);  perm = the k rows with the largest s (ties: lowest index first;
//                           nan ranks above everything, as torch.sort(descending=True) has it);  xt_j = X[perm_j] * s[perm_j]
//   O variant (pool == 0):  xt = Wprev (the weight is input and hidden state)
//   GRU cell (torch.nn.GRU, gates ordered r | z | n):  r = sig(W_ir xt + b_ir + W_hr h + b_hr),  z likewise,
//                           n = tanh(W_in xt + b_in + r * (W_hn h + b_hn)),  W_t = (1 - z) * n + z * h,   h = Wprev [k, F]
// and the adjoint is one workgroup as well (gate chain, the six weight / bias gradients, d/dWprev, and through the
// selected rows and their scores d/dX and d/dp; the selection itself carries no gradient).  Plain fmaf dot products in
// index order: deterministic, no atomics.
#include "pgt_common.h"

namespace {

constexpr int EV_THREADS = 256;
constexpr int EV_MAX_F = 64;      // hidden width (= input width) of the GRU: the reference's examples use 8
constexpr int EV_MAX_K = 64;      // rows the pooling keeps (= the GRU's batch = F in the reference)

struct EvolveArgs {
  const float* X; int64_t ldx; int N;          // [N, F] node features (H variant)
  const float* p;                               // [F] TopKPooling projection (select.weight)
  const float* Wih; const float* Whh;           // [3F, F] each, rows r | z | n
  const float* bih; const float* bhh;           // [3F] each or null
  const float* Wprev;                           // [k, F] previous weight (GRU hidden state, batch = k)
  int F, k, pool;
  float* Wnew;                                  // [k, F]
  // saved for the adjoint
  int32_t* perm; float* score;                  // [k] selected rows and their tanh scores
  float* gates;                                 // [4][k][F]: r, z, n, hn_pre = W_hn h + b_hn
  float* xt;                                    // [k][F] GRU input
};

__device__ __forceinline__ float ev_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(EV_THREADS) void evolve_fwd_kernel(EvolveArgs a) {
  __shared__ float s_score[4096];
  __shared__ float s_xt[EV_MAX_K * EV_MAX_F];
  __shared__ int s_perm[EV_MAX_K];
  __shared__ float s_sel[EV_MAX_K];
  __shared__ float s_norm;
  const int tid = threadIdx.x, F = a.F, k = a.k;
  if (a.pool) {
    if (tid == 0) {                                       // |p| (F <= 64 terms, index order)
      float t = 0.f;
      for (int f = 0; f < F; ++f) t = fmaf(a.p[f], a.p[f], t);
      s_norm = sqrtf(t);
    }
    __syncthreads();
    const float nrm = s_norm;
    for (int i = tid; i < a.N; i += EV_THREADS) {
      const float* x = a.X + (int64_t)i * a.ldx;
      float d = 0.f;
      for (int f = 0; f < F; ++f) d = fmaf(x[f], a.p[f], d);
      s_score[i] = tanhf(d / nrm);
    }
    __syncthreads();
    // top-k by RANK: element i counts the elements ahead of it in the descending order (larger score; nan above everything,
    // as torch.sort(descending=True) has it; ties to the lower index) out of LDS — every thread scans all N scores, no
    // serial rounds; the k elements of rank < k write themselves to perm[rank]
    for (int i = tid; i < a.N; i += EV_THREADS) {
      const float v = s_score[i];
      const bool vn = v != v;
      int rank = 0;
      for (int j = 0; j < a.N; ++j) {
        const float u = s_score[j];
        const bool un = u != u;
        const bool ahead = un ? (!vn || j < i) : (!vn && (u > v || (u == v && j < i)));
        rank += ahead ? 1 : 0;
      }
      if (rank < k) { a.perm[rank] = i; a.score[rank] = v; s_perm[rank] = i; s_sel[rank] = v; }
    }
    __syncthreads();
    for (int e = tid; e < k * F; e += EV_THREADS) {
      const int j = e / F, f = e - j * F;
      const int r = s_perm[j];
      const float v = a.X[(int64_t)r * a.ldx + f] * s_sel[j];
      s_xt[e] = v;
      a.xt[e] = v;
    }
  } else {
    for (int e = tid; e < k * F; e += EV_THREADS) { s_xt[e] = a.Wprev[e]; a.xt[e] = a.Wprev[e]; }
  }
  __syncthreads();
  float* gr = a.gates;
  float* gz = a.gates + (int64_t)k * F;
  float* gn = a.gates + (int64_t)2 * k * F;
  float* ghn = a.gates + (int64_t)3 * k * F;
  for (int e = tid; e < k * F; e += EV_THREADS) {
    const int j = e / F, o = e - j * F;
    const float* x = s_xt + j * F;
    const float* h = a.Wprev + (int64_t)j * F;
    float ir = a.bih ? a.bih[o] : 0.f, iz = a.bih ? a.bih[F + o] : 0.f, in_ = a.bih ? a.bih[2 * F + o] : 0.f;
    float hr = a.bhh ? a.bhh[o] : 0.f, hz = a.bhh ? a.bhh[F + o] : 0.f, hn = a.bhh ? a.bhh[2 * F + o] : 0.f;
    for (int f = 0; f < F; ++f) {
      const float xv = x[f], hv = h[f];
      ir = fmaf(a.Wih[(int64_t)o * F + f], xv, ir);
      iz = fmaf(a.Wih[(int64_t)(F + o) * F + f], xv, iz);
      in_ = fmaf(a.Wih[(int64_t)(2 * F + o) * F + f], xv, in_);
      hr = fmaf(a.Whh[(int64_t)o * F + f], hv, hr);
      hz = fmaf(a.Whh[(int64_t)(F + o) * F + f], hv, hz);
      hn = fmaf(a.Whh[(int64_t)(2 * F + o) * F + f], hv, hn);
    }
    const float r = ev_sigmoid(ir + hr), z = ev_sigmoid(iz + hz);
    const float n = tanhf(in_ + r * hn);
    gr[e] = r; gz[e] = z; gn[e] = n; ghn[e] = hn;
    a.Wnew[e] = (1.f - z) * n + z * h[o];
  }
}

struct EvolveBwdArgs {
  const float* dWnew;                           // [k, F]
  const float* X; int64_t ldx; int N;
  const float* p; const float* Wih; const float* Whh; const float* Wprev;
  const int32_t* perm; const float* score; const float* gates; const float* xt;
  int F, k, pool;
  float* dX; int64_t lddx;                      // [N, F], zero-filled by the caller (H variant)
  float* dp;                                    // [F]
  float* dWih; float* dWhh;                     // [3F, F]
  float* dbih; float* dbhh;                     // [3F] or null
  float* dWprev;                                // [k, F]
};

__global__ __launch_bounds__(EV_THREADS) void evolve_bwd_kernel(EvolveBwdArgs a) {
  __shared__ float s_gi[3 * EV_MAX_K * EV_MAX_F];   // d pre-activations of the input side  [3][k][F]  (r | z | n)
  __shared__ float s_gh[3 * EV_MAX_K * EV_MAX_F];   // ... of the hidden side
  __shared__ float s_dxt[EV_MAX_K * EV_MAX_F];
  __shared__ float s_ds[EV_MAX_K];
  const int tid = threadIdx.x, F = a.F, k = a.k, kF = k * F;
  const float* gr = a.gates;
  const float* gz = a.gates + (int64_t)kF;
  const float* gn = a.gates + (int64_t)2 * kF;
  const float* ghn = a.gates + (int64_t)3 * kF;
  for (int e = tid; e < kF; e += EV_THREADS) {
    const float g = a.dWnew[e], r = gr[e], z = gz[e], n = gn[e], hn = ghn[e], h = a.Wprev[e];
    const float dn = g * (1.f - z), dz = g * (h - n);
    const float dn_pre = dn * (1.f - n * n);
    const float dr = dn_pre * hn;
    const float dr_pre = dr * r * (1.f - r), dz_pre = dz * z * (1.f - z);
    s_gi[e] = dr_pre; s_gi[kF + e] = dz_pre; s_gi[2 * kF + e] = dn_pre;
    s_gh[e] = dr_pre; s_gh[kF + e] = dz_pre; s_gh[2 * kF + e] = dn_pre * r;
  }
  __syncthreads();
  // weight / bias gradients: dW_ih[g F + o, f] = sum_j d gi[g, j, o] xt[j, f]; dW_hh likewise with h
  for (int e = tid; e < 3 * F * F; e += EV_THREADS) {
    const int row = e / F, f = e - row * F, g = row / F, o = row - g * F;
    float wi = 0.f, wh = 0.f;
    for (int j = 0; j < k; ++j) {
      wi = fmaf(s_gi[g * kF + j * F + o], a.xt[j * F + f], wi);
      wh = fmaf(s_gh[g * kF + j * F + o], a.Wprev[(int64_t)j * F + f], wh);
    }
    a.dWih[e] = wi;
    a.dWhh[e] = wh;
  }
  for (int e = tid; e < 3 * F; e += EV_THREADS) {
    const int g = e / F, o = e - g * F;
    float bi = 0.f, bh = 0.f;
    for (int j = 0; j < k; ++j) { bi += s_gi[g * kF + j * F + o]; bh += s_gh[g * kF + j * F + o]; }
    if (a.dbih) a.dbih[e] = bi;
    if (a.dbhh) a.dbhh[e] = bh;
  }
  // d xt = d gi W_ih ; d h = dWnew * z + d gh W_hh
  for (int e = tid; e < kF; e += EV_THREADS) {
    const int j = e / F, f = e - j * F;
    float dx = 0.f, dh = a.dWnew[e] * gz[e];
    for (int g = 0; g < 3; ++g)
      for (int o = 0; o < F; ++o) {
        dx = fmaf(s_gi[g * kF + j * F + o], a.Wih[(int64_t)(g * F + o) * F + f], dx);
        dh = fmaf(s_gh[g * kF + j * F + o], a.Whh[(int64_t)(g * F + o) * F + f], dh);
      }
    s_dxt[e] = dx;
    a.dWprev[e] = a.pool ? dh : dh + dx;          // O variant: the weight is also the input
  }
  __syncthreads();
  if (!a.pool) return;
  // through xt_j = X[perm_j] * s_j,  s_j = tanh(X[perm_j] . p / |p|)
  __shared__ float s_norm, s_dot[EV_MAX_K];
  if (tid == 0) {
    float t = 0.f;
    for (int f = 0; f < F; ++f) t = fmaf(a.p[f], a.p[f], t);
    s_norm = sqrtf(t);
  }
  for (int j = tid; j < k; j += EV_THREADS) {
    const int r = a.perm[j];
    float ds = 0.f, d = 0.f;
    if (r >= 0) {
      const float* x = a.X + (int64_t)r * a.ldx;
      for (int f = 0; f < F; ++f) { ds = fmaf(s_dxt[j * F + f], x[f], ds); d = fmaf(x[f], a.p[f], d); }
    }
    const float s = a.score[j];
    s_ds[j] = ds * (1.f - s * s);                 // d (pre-tanh score)
    s_dot[j] = d;
  }
  __syncthreads();
  const float nrm = s_norm;
  for (int e = tid; e < kF; e += EV_THREADS) {
    const int j = e / F, f = e - j * F;
    const int r = a.perm[j];
    if (r >= 0) a.dX[(int64_t)r * a.lddx + f] = s_dxt[e] * a.score[j] + s_ds[j] * a.p[f] / nrm;   // perm rows are distinct
  }
  for (int f = tid; f < F; f += EV_THREADS) {
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const int r = a.perm[j];
      if (r < 0) continue;
      const float x = a.X[(int64_t)r * a.ldx + f];
      acc += s_ds[j] * (x / nrm - s_dot[j] * a.p[f] / (nrm * nrm * nrm));
    }
    a.dp[f] = acc;
  }
}

}  // namespace

extern "C" int pgt_evolve_weight_f32(const float* X, int64_t ldx, int64_t N, const float* p, const float* Wih, const float* Whh,
                                     const float* bih, const float* bhh, const float* Wprev, int64_t F, int64_t k, int pool,
                                     float* Wnew, int32_t* perm, float* score, float* gates, float* xt, pgt_stream_t stream) {
  PGT_REQUIRE(F >= 1 && F <= EV_MAX_F && k >= 1 && k <= EV_MAX_K, "pgt_evolve_weight_f32: F and k must lie in [1, 64]");
  PGT_REQUIRE(Wih && Whh && Wprev && Wnew && gates && xt, "pgt_evolve_weight_f32: null pointer");
  PGT_REQUIRE((bih == nullptr) == (bhh == nullptr), "pgt_evolve_weight_f32: the two GRU biases go together");
  if (pool) {
    PGT_REQUIRE(X && p && perm && score, "pgt_evolve_weight_f32: null pointer (pooling inputs)");
    PGT_REQUIRE(N >= k && N <= 4096, "pgt_evolve_weight_f32: needs k <= N <= 4096 nodes (got %lld)", (long long)N);
  }
  EvolveArgs a{X, ldx, (int)N, p, Wih, Whh, bih, bhh, Wprev, (int)F, (int)k, pool, Wnew, perm, score, gates, xt};
  PGT_LAUNCH(evolve_fwd_kernel, dim3(1), dim3(EV_THREADS), stream, a);
  return pgt_check_launch("pgt_evolve_weight_f32");
}

extern "C" int pgt_evolve_weight_bwd_f32(const float* dWnew, const float* X, int64_t ldx, int64_t N, const float* p,
                                         const float* Wih, const float* Whh, const float* Wprev, const int32_t* perm,
                                         const float* score, const float* gates, const float* xt, int64_t F, int64_t k, int pool,
                                         float* dX, int64_t lddx, float* dp, float* dWih, float* dWhh, float* dbih, float* dbhh,
                                         float* dWprev, pgt_stream_t stream) {
  PGT_REQUIRE(F >= 1 && F <= EV_MAX_F && k >= 1 && k <= EV_MAX_K, "pgt_evolve_weight_bwd_f32: F and k must lie in [1, 64]");
  PGT_REQUIRE(dWnew && Wih && Whh && Wprev && gates && xt && dWih && dWhh && dWprev, "pgt_evolve_weight_bwd_f32: null pointer");
  if (pool) PGT_REQUIRE(X && p && perm && score && dX && dp, "pgt_evolve_weight_bwd_f32: null pointer (pooling operands)");
  EvolveBwdArgs a{dWnew, X, ldx, (int)N, p, Wih, Whh, Wprev, perm, score, gates, xt, (int)F, (int)k, pool,
                  dX, lddx, dp, dWih, dWhh, dbih, dbhh, dWprev};
  PGT_LAUNCH(evolve_bwd_kernel, dim3(1), dim3(EV_THREADS), stream, a);
  return pgt_check_launch("pgt_evolve_weight_bwd_f32");
}
