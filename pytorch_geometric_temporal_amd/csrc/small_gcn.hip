// One GCN layer on a small graph, straight from the edge list, in one launch per direction
// (GCNConv_Fixed_W.forward, evolvegcno.py:76-101 = evolvegcnh.py's conv_layer; gcn_norm as PyG has it).
//
// BASELINE configs[4] (examples/recurrent/evolvegcnh_example.py:38-50): 129 nodes, a NEW edge list of 836 - 2 158 edges every
// snapshot, 8 features.  Through the general path a new edge list costs the device preparation (radix sorts, degree and
// normalisation kernels: ~10 launches and a host check of the `info` words), then a product and an aggregation launch —
// for a graph whose every array fits one CU's LDS ten times over.  Here one workgroup does all of it:
//   edges -> LDS (endpoints 16-bit, range-checked);  h = X W in LDS;
//   add_remaining_self_loops: self-loop edges leave the list, node i's loop weighs what its LAST self-loop edge weighed,
//     else `fill` (1, or 2 for `improved`; 1 without edge weights);
//   destination lists by counting (LDS integer atomics: order-free) + a scan + a fill in EDGE ORDER (one thread per row
//     walks the edge list, eight endpoints per LDS read): sums run in the order scatter_add / index_add_ run them on the
//     reference's CPU path — deterministic, no float atomics;
//   deg_i = sum of its list's weights + the loop's;  dis = deg^-1/2 (inf -> 0);  coefficient (dis[src] w) dis[dst];
//   out_i = sum over the list of coef * h[src] (product rounded, then added: message then index_add_), the loop term last.
// The coefficients ([E] edges + [N] loops) are saved: the adjoint (one workgroup as well) builds the SOURCE lists the same
// way and returns d/dW = X^T (A^T G) and, when wanted, d/dX = (A^T G) W^T.
#include "pgt_common.h"

namespace {

constexpr int GS_THREADS = 512;
constexpr int GS_MAX_N = 512;        // nodes (one thread per row in the list phases)
constexpr int GS_MAX_E = 4096;       // edges
constexpr int GS_MAX_F = 64;         // feature widths
constexpr int GS_MAX_NF = 8192;      // N * out width (h in LDS)

struct GcnSmallArgs {
  const int64_t* ei; const float* ew; int E, N;
  float fill; int loops, normalize;
  const float* X; int64_t ldx; const float* W; int Fi, Fo;   // W [Fi, Fo]
  float* out;                                                // [N, Fo]
  float* coef;                                               // [E + N]
  int32_t* info;                                             // info[0] += edges with an endpoint outside [0, N) (skipped)
};

struct GcnSmallBwdArgs {
  const int64_t* ei; const float* coef; int E, N, loops;
  const float* G; int64_t ldg;                               // d/d out [N, Fo]
  const float* X; int64_t ldx; const float* W; int Fi, Fo;
  float* dW;                                                 // [Fi, Fo]
  float* dX; int64_t lddx;                                   // [N, Fi] or null
};

__device__ __forceinline__ float gs_inv_sqrt_or_zero(float d) {
  const float v = 1.0f / sqrtf(d);   // deg.pow(-0.5)
  return (v == INFINITY) ? 0.f : v;  // masked_fill_(== inf, 0)
}

// exclusive scan of cnt[0 .. n) into off[0 .. n] (n <= GS_THREADS); every thread of the workgroup calls it
__device__ __forceinline__ void gs_scan(const int* cnt, int* off, int* tmp, int n) {
  const int tid = threadIdx.x;
  tmp[tid] = tid < n ? cnt[tid] : 0;
  __syncthreads();
  for (int d = 1; d < GS_THREADS; d <<= 1) {
    const int t = tid >= d ? tmp[tid - d] : 0;
    __syncthreads();
    tmp[tid] += t;
    __syncthreads();
  }
  if (tid < n) off[tid + 1] = tmp[tid];
  if (tid == 0) off[0] = 0;
  __syncthreads();
}

// lists in edge order: row i collects the edges whose key endpoint is i (and that are still in the list: `other` != 0xffff
// marks removed / invalid edges through key = 0xffff)
__device__ __forceinline__ void gs_fill_lists(const unsigned short* key, int E, int N, const int* off, unsigned short* list) {
  const int i = threadIdx.x;
  if (i < N) {
    int k = off[i];
    const int E8 = E & ~7;
    for (int e = 0; e < E8; e += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(key + e);          // eight endpoints (the same address for every
      const unsigned w[4] = {v.x, v.y, v.z, v.w};                          // thread: an LDS broadcast)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((w[q] & 0xffffu) == (unsigned)i) list[k++] = (unsigned short)(e + 2 * q);
        if ((w[q] >> 16) == (unsigned)i) list[k++] = (unsigned short)(e + 2 * q + 1);
      }
    }
    for (int e = E8; e < E; ++e)
      if (key[e] == i) list[k++] = (unsigned short)e;
  }
}

__global__ __launch_bounds__(GS_THREADS) void gcn_small_fwd_kernel(GcnSmallArgs a) {
  __shared__ float s_h[GS_MAX_NF];
  __shared__ __attribute__((aligned(16))) unsigned short s_src[GS_MAX_E];
  __shared__ __attribute__((aligned(16))) unsigned short s_dst[GS_MAX_E];    // 0xffff: not in the list (self-loop / invalid)
  __shared__ float s_w[GS_MAX_E];
  __shared__ unsigned short s_list[GS_MAX_E];
  __shared__ int s_cnt[GS_MAX_N], s_off[GS_MAX_N + 1], s_tmp[GS_THREADS], s_loop[GS_MAX_N];
  __shared__ float s_dis[GS_MAX_N], s_lw[GS_MAX_N];
  const int tid = threadIdx.x, E = a.E, N = a.N, Fi = a.Fi, Fo = a.Fo;
  for (int i = tid; i < N; i += GS_THREADS) { s_cnt[i] = 0; s_loop[i] = -1; }
  __syncthreads();
  for (int e = tid; e < E; e += GS_THREADS) {
    const int64_t r = a.ei[e], c = a.ei[(int64_t)E + e];
    const bool bad = r < 0 || r >= N || c < 0 || c >= N;
    const bool self = !bad && a.loops && r == c;
    s_src[e] = bad ? (unsigned short)0xffff : (unsigned short)r;
    s_dst[e] = (bad || self) ? (unsigned short)0xffff : (unsigned short)c;
    s_w[e] = a.ew ? a.ew[e] : 1.f;
    if (bad) atomicAdd(a.info, 1);
    else if (self) atomicMax(&s_loop[r], e);                 // the last self-loop edge names the loop's weight
    else atomicAdd(&s_cnt[c], 1);
  }
  for (int idx = tid; idx < N * Fo; idx += GS_THREADS) {     // h = X W  (evolvegcno.py:92)
    const int i = idx / Fo, f = idx - i * Fo;
    const float* x = a.X + (int64_t)i * a.ldx;
    float acc = 0.f;
    for (int c = 0; c < Fi; ++c) acc = fmaf(x[c], a.W[c * Fo + f], acc);
    s_h[idx] = acc;
  }
  __syncthreads();
  gs_scan(s_cnt, s_off, s_tmp, N);
  gs_fill_lists(s_dst, E, N, s_off, s_list);
  __syncthreads();
  if (tid < N) {
    const int i = tid;
    float lw = 0.f;
    if (a.loops) lw = s_loop[i] >= 0 ? s_w[s_loop[i]] : a.fill;
    float deg = 0.f;
    for (int k = s_off[i]; k < s_off[i + 1]; ++k) deg += s_w[s_list[k]];
    if (a.loops) deg += lw;
    s_dis[i] = a.normalize ? gs_inv_sqrt_or_zero(deg) : 1.f;
    s_lw[i] = lw;
  }
  __syncthreads();
  for (int e = tid; e < E; e += GS_THREADS) {
    float c = 0.f;
    if (s_dst[e] != 0xffff) c = a.normalize ? pgt_mul_rn(pgt_mul_rn(s_dis[s_src[e]], s_w[e]), s_dis[s_dst[e]]) : s_w[e];
    a.coef[e] = c;
    s_w[e] = c;                                              // own element only
  }
  if (tid < N) {
    const float d = s_dis[tid];
    const float c = a.loops ? (a.normalize ? pgt_mul_rn(pgt_mul_rn(d, s_lw[tid]), d) : s_lw[tid]) : 0.f;
    a.coef[E + tid] = c;
    s_lw[tid] = c;
  }
  __syncthreads();
  for (int idx = tid; idx < N * Fo; idx += GS_THREADS) {
    const int i = idx / Fo, f = idx - i * Fo;
    float acc = 0.f;
    for (int k = s_off[i]; k < s_off[i + 1]; ++k) {
      const int e = s_list[k];
      acc = pgt_add_rn(acc, pgt_mul_rn(s_w[e], s_h[(int)s_src[e] * Fo + f]));
    }
    if (a.loops) acc = pgt_add_rn(acc, pgt_mul_rn(s_lw[i], s_h[idx]));
    a.out[idx] = acc;
  }
}

__global__ __launch_bounds__(GS_THREADS) void gcn_small_bwd_kernel(GcnSmallBwdArgs a) {
  __shared__ float s_gt[GS_MAX_NF];                          // A^T G
  __shared__ __attribute__((aligned(16))) unsigned short s_src[GS_MAX_E];   // 0xffff: zero coefficient (not in the list)
  __shared__ unsigned short s_dst[GS_MAX_E];
  __shared__ float s_c[GS_MAX_E];
  __shared__ unsigned short s_list[GS_MAX_E];
  __shared__ int s_cnt[GS_MAX_N], s_off[GS_MAX_N + 1], s_tmp[GS_THREADS];
  const int tid = threadIdx.x, E = a.E, N = a.N, Fi = a.Fi, Fo = a.Fo;
  for (int i = tid; i < N; i += GS_THREADS) s_cnt[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += GS_THREADS) {
    const int64_t r = a.ei[e], c = a.ei[(int64_t)E + e];
    const bool out = r < 0 || r >= N || c < 0 || c >= N || (a.loops && r == c);
    s_src[e] = out ? (unsigned short)0xffff : (unsigned short)r;
    s_dst[e] = out ? (unsigned short)0 : (unsigned short)c;
    s_c[e] = a.coef[e];
    if (!out) atomicAdd(&s_cnt[r], 1);
  }
  __syncthreads();
  gs_scan(s_cnt, s_off, s_tmp, N);
  gs_fill_lists(s_src, E, N, s_off, s_list);
  __syncthreads();
  for (int idx = tid; idx < N * Fo; idx += GS_THREADS) {     // (A^T G)_i = sum over the edges LEAVING i, edge order, loop last
    const int i = idx / Fo, f = idx - i * Fo;
    float acc = 0.f;
    for (int k = s_off[i]; k < s_off[i + 1]; ++k) {
      const int e = s_list[k];
      acc = fmaf(s_c[e], a.G[(int64_t)s_dst[e] * a.ldg + f], acc);
    }
    if (a.loops) acc = fmaf(a.coef[E + i], a.G[(int64_t)i * a.ldg + f], acc);
    s_gt[idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < Fi * Fo; idx += GS_THREADS) {    // d/dW = X^T (A^T G), rows in index order
    const int c = idx / Fo, f = idx - c * Fo;
    float acc = 0.f;
    for (int i = 0; i < N; ++i) acc = fmaf(a.X[(int64_t)i * a.ldx + c], s_gt[i * Fo + f], acc);
    a.dW[idx] = acc;
  }
  if (a.dX)
    for (int idx = tid; idx < N * Fi; idx += GS_THREADS) {   // d/dX = (A^T G) W^T
      const int i = idx / Fi, c = idx - i * Fi;
      float acc = 0.f;
      for (int f = 0; f < Fo; ++f) acc = fmaf(s_gt[i * Fo + f], a.W[c * Fo + f], acc);
      a.dX[(int64_t)i * a.lddx + c] = acc;
    }
}

}  // namespace

extern "C" int pgt_gcn_small_fits(int64_t N, int64_t E, int64_t Fi, int64_t Fo) {
  return N >= 1 && N <= GS_MAX_N && E >= 0 && E <= GS_MAX_E && Fi >= 1 && Fi <= GS_MAX_F && Fo >= 1 && Fo <= GS_MAX_F &&
         N * Fo <= GS_MAX_NF;
}

extern "C" int pgt_gcn_small_f32(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N, int improved,
                                 int add_self_loops, int normalize, const float* X, int64_t ldx, const float* W, int64_t Fi,
                                 int64_t Fo, float* out, float* coef, int32_t* info, pgt_stream_t stream) {
  PGT_REQUIRE(pgt_gcn_small_fits(N, E, Fi, Fo), "pgt_gcn_small_f32: N = %lld, E = %lld, widths %lld -> %lld outside N <= %d, "
              "E <= %d, widths <= %d, N * out <= %d", (long long)N, (long long)E, (long long)Fi, (long long)Fo, GS_MAX_N,
              GS_MAX_E, GS_MAX_F, GS_MAX_NF);
  PGT_REQUIRE((edge_index || E == 0) && X && W && out && coef && info, "pgt_gcn_small_f32: null pointer");
  PGT_REQUIRE(ldx >= Fi, "pgt_gcn_small_f32: row stride below the row width");
  GcnSmallArgs a{edge_index, edge_weight, (int)E, (int)N,
                 // PyG: self-loops first, THEN `None -> ones`: without weights every loop weighs 1 (improved has no effect)
                 (improved && edge_weight) ? 2.f : 1.f, (add_self_loops && normalize) ? 1 : 0, normalize ? 1 : 0,
                 X, ldx, W, (int)Fi, (int)Fo, out, coef, info};
  PGT_LAUNCH(gcn_small_fwd_kernel, dim3(1), dim3(GS_THREADS), stream, a);
  return pgt_check_launch("pgt_gcn_small_f32");
}

extern "C" int pgt_gcn_small_bwd_f32(const int64_t* edge_index, const float* coef, int64_t E, int64_t N, int add_self_loops,
                                     int normalize, const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W,
                                     int64_t Fi, int64_t Fo, float* dW, float* dX, int64_t lddx, pgt_stream_t stream) {
  PGT_REQUIRE(pgt_gcn_small_fits(N, E, Fi, Fo), "pgt_gcn_small_bwd_f32: N = %lld, E = %lld, widths %lld -> %lld outside the "
              "small-graph limits", (long long)N, (long long)E, (long long)Fi, (long long)Fo);
  PGT_REQUIRE((edge_index || E == 0) && coef && G && X && W && dW, "pgt_gcn_small_bwd_f32: null pointer");
  PGT_REQUIRE(ldg >= Fo && ldx >= Fi && (!dX || lddx >= Fi), "pgt_gcn_small_bwd_f32: row stride below the row width");
  GcnSmallBwdArgs a{edge_index, coef, (int)E, (int)N, (add_self_loops && normalize) ? 1 : 0, G, ldg, X, ldx, W, (int)Fi, (int)Fo,
                    dW, dX, lddx};
  PGT_LAUNCH(gcn_small_bwd_kernel, dim3(1), dim3(GS_THREADS), stream, a);
  return pgt_check_launch("pgt_gcn_small_bwd_f32");
}
