// Graph preparation on the device: COO (edge_index int64 [2,E], edge_weight f32 [E]) -> CSR operators.
//
// Replaces, once per distinct graph instead of once per conv call:
//   DConv        dcrnn.py:59-77   (dense N x N adjacency, row/col sums, reciprocals, nonzero() of adj^T)
//   BatchedDConv dcrnn.py:277-290 (scatter_add_ degrees, argsort of the reversed list)
//   GCNConv      PyG gcn_norm (add_remaining_self_loops, deg^-1/2 scaling)
//   ChebConv     PyG get_laplacian + 2L/lambda_max - I; astgcn.py:82-110
//
// Everything is O(E log E): stable LSD radix sorts (rocPRIM) give CSR rows whose slots keep the reference's edge
// order, so the aggregation kernel's sequential per-row sum reproduces index_add_'s CPU summation order.
// Degrees are row sums over those ordered slots (deterministic, no float atomics).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "pgt_common.h"
#ifndef PGT_EMU
#include <rocprim/device/device_radix_sort.hpp>
#endif

namespace {

struct Ws {
  uint64_t* k_in;
  uint64_t* k_out;
  int32_t* v_in;
  int32_t* perm;
  int32_t* l_dst;
  int32_t* l_src;
  float* l_val;
  int32_t* sigma;   // [E]
  int32_t* node_i;  // [N]
  float* node_f;    // [N]
  float* scal;      // [4]
  void* tmp;
  size_t tmp_bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_tmp_bytes(int64_t L) {
#ifdef PGT_EMU
  (void)L;
  return 256;
#else
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(L > 0 ? L : 1), 0u, 64u, (hipStream_t)0);
  return bytes + 256;
#endif
}

size_t ws_layout(int64_t E, int64_t N, char* base, Ws* w) {
  const int64_t L = E + 2 * N + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  char* p;
  p = take(L * 8); if (w) w->k_in = (uint64_t*)p;
  p = take(L * 8); if (w) w->k_out = (uint64_t*)p;
  p = take(L * 4); if (w) w->v_in = (int32_t*)p;
  p = take(L * 4); if (w) w->perm = (int32_t*)p;
  p = take(L * 4); if (w) w->l_dst = (int32_t*)p;
  p = take(L * 4); if (w) w->l_src = (int32_t*)p;
  p = take(L * 4); if (w) w->l_val = (float*)p;
  p = take((E + 1) * 4); if (w) w->sigma = (int32_t*)p;
  p = take((N + 1) * 4); if (w) w->node_i = (int32_t*)p;
  p = take((N + 1) * 4); if (w) w->node_f = (float*)p;
  p = take(64); if (w) w->scal = (float*)p;
  const size_t tb = sort_tmp_bytes(L);
  p = take(tb); if (w) { w->tmp = p; w->tmp_bytes = tb; }
  return off + 256;
}

int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

// (k_in, v_in) -> (k_out, perm), ascending, stable.
int sort_pairs(const Ws& w, int64_t L, int end_bit, pgt_stream_t stream) {
  if (L <= 0) return PGT_OK;
#ifdef PGT_EMU
  (void)end_bit; (void)stream;
  std::vector<int32_t> idx((size_t)L);
  for (int64_t i = 0; i < L; ++i) idx[(size_t)i] = (int32_t)i;
  std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return w.k_in[a] < w.k_in[b]; });
  for (int64_t i = 0; i < L; ++i) {
    w.k_out[i] = w.k_in[idx[(size_t)i]];
    w.perm[i] = w.v_in[idx[(size_t)i]];
  }
  return PGT_OK;
#else
  size_t bytes = w.tmp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(w.tmp, bytes, (const uint64_t*)w.k_in, w.k_out, (const int32_t*)w.v_in,
                                           w.perm, (size_t)L, 0u, (unsigned)end_bit, (hipStream_t)stream);
  if (e != hipSuccess) {
    pgt_set_error("graph prep: radix sort failed: %s", hipGetErrorString(e));
    return PGT_ERR_LAUNCH;
  }
  return PGT_OK;
#endif
}

// ------------------------------------------------------------------------------------------------ kernels

__global__ __launch_bounds__(256) void k_check_edges(const int64_t* __restrict__ ei, const float* __restrict__ w,
                                                      int64_t E, int64_t N, int32_t* info) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = ei[e], c = ei[E + e];
  if (r < 0 || r >= N || c < 0 || c >= N) atomicAdd(info + 2, 1);
  if (w != nullptr && w[e] == 0.f) atomicAdd(info + 1, 1);
}

// keys from the staged list: k_in[q] = l_dst[q] (N = dropped), v_in[q] = q
__global__ __launch_bounds__(256) void k_keys_from_list(const int32_t* __restrict__ l_dst, int64_t L,
                                                         uint64_t* k_in, int32_t* v_in) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= L) return;
  k_in[q] = (uint64_t)(uint32_t)l_dst[q];
  v_in[q] = (int32_t)q;
}

__global__ __launch_bounds__(256) void k_rowptr(const uint64_t* __restrict__ keys_sorted, int64_t L, int64_t N,
                                                 int32_t* rowptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > N) return;
  int64_t lo = 0, hi = L;  // first index with key >= i
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < (uint64_t)i) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

__global__ __launch_bounds__(256) void k_gather_slots(const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ l_src,
                                                       const float* __restrict__ l_val, int64_t L, int32_t* col,
                                                       float* val) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= L) return;
  const int32_t p = perm[q];
  col[q] = l_src[p];
  val[q] = l_val[p];
}

__global__ __launch_bounds__(256) void k_rowsum(const int32_t* __restrict__ rowptr, const float* __restrict__ val,
                                                 int64_t N, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float s = 0.f;
  for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) s += val[q];
  out[i] = s;
}

// stage (dst, src, val) from the raw COO. mode 0: dst = col, src = row ; mode 1: dst = row, src = col.
// drop_loops: self-loops are dropped (dst = N).  Out-of-range endpoints are always dropped.
__global__ __launch_bounds__(256) void k_stage_coo(const int64_t* __restrict__ ei, const float* __restrict__ w,
                                                    int64_t E, int64_t N, int mode, int drop_loops, int32_t* l_dst,
                                                    int32_t* l_src, float* l_val) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = ei[e], c = ei[E + e];
  const bool bad = (r < 0 || r >= N || c < 0 || c >= N) || (drop_loops && r == c);
  const int64_t d = mode == 0 ? c : r, s = mode == 0 ? r : c;
  l_dst[e] = bad ? (int32_t)N : (int32_t)d;
  l_src[e] = bad ? 0 : (int32_t)s;
  l_val[e] = w ? w[e] : 1.f;
}

// ---- DConv specifics
__global__ __launch_bounds__(256) void k_dconv_val_by_src(const int32_t* __restrict__ col, int64_t L,
                                                           const float* __restrict__ deg, float* val) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= L) return;
  val[q] = 1.0f / deg[col[q]];  // torch.reciprocal: 1/0 = inf (dcrnn.py:70-71)
}

__global__ __launch_bounds__(256) void k_dconv_val_by_row(const int32_t* __restrict__ rowptr, int64_t N,
                                                           const float* __restrict__ deg, float* val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float v = 1.0f / deg[i];
  for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) val[q] = v;
}

// info[3] += the slots of an operator whose coefficient is not finite (1 / deg of a node without out- / in-edges, dcrnn.py:71-77)
__global__ __launch_bounds__(256) void k_count_nonfinite(const float* __restrict__ val, int64_t L, int32_t* info) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= L) return;
  if (!(fabsf(val[q]) <= 3.0e38f)) atomicAdd(info + 3, 1);
}

__global__ __launch_bounds__(256) void k_dconv_sigma_keys(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                                           uint64_t* k_in, int32_t* v_in) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = ei[e], c = ei[E + e];
  const bool bad = (r < 0 || r >= N || c < 0 || c >= N);
  // sort_idx = reverse_edge_index[0] * num_nodes + reverse_edge_index[1]  (dcrnn.py:289) == row-major order of adj^T
  k_in[e] = bad ? (uint64_t)N * (uint64_t)N : (uint64_t)c * (uint64_t)N + (uint64_t)r;
  v_in[e] = (int32_t)e;
}

__global__ __launch_bounds__(256) void k_count_dups(const uint64_t* __restrict__ keys_sorted, int64_t E,
                                                     int32_t* info) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p + 1 >= E) return;
  if (keys_sorted[p] == keys_sorted[p + 1]) atomicAdd(info + 0, 1);
}

// reversed list (dcrnn.py:76-77,87 / :288-290): position p holds edge sigma[p] reversed, but its coefficient is
// norm_in[p] = deg_in_inv[row[p]] — taken at the ORIGINAL position p (dcrnn.py:74), reproduced on purpose.
__global__ __launch_bounds__(256) void k_dconv_stage_reverse(const int64_t* __restrict__ ei,
                                                              const int32_t* __restrict__ sigma, int64_t E,
                                                              int64_t N, const float* __restrict__ deg_in,
                                                              int transpose, int32_t* l_dst, int32_t* l_src,
                                                              float* l_val) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= E) return;
  const int32_t e = sigma[p];
  const int64_t r = ei[e], c = ei[E + e];
  const int64_t rp = ei[p];
  const bool bad = (r < 0 || r >= N || c < 0 || c >= N || rp < 0 || rp >= N);
  // propagate(reverse_edge_index): source = col[e], target = row[e]
  const int64_t d = transpose ? c : r, s = transpose ? r : c;
  l_dst[p] = bad ? (int32_t)N : (int32_t)d;
  l_src[p] = bad ? 0 : (int32_t)s;
  l_val[p] = bad ? 0.f : 1.0f / deg_in[rp];
}

// ---- GCN specifics
__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void k_gcn_loop_eid(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                                       int32_t* loop_eid) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = ei[e], c = ei[E + e];
  if (r == c && r >= 0 && r < N) atomicMax(loop_eid + r, (int32_t)e);  // last self-loop wins (index_put order)
}

__global__ __launch_bounds__(256) void k_gcn_stage_loops(const float* __restrict__ w,
                                                          const int32_t* __restrict__ loop_eid, int64_t E,
                                                          int64_t N, float fill, int32_t* l_dst, int32_t* l_src,
                                                          float* l_val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t e = loop_eid[i];
  l_dst[E + i] = (int32_t)i;
  l_src[E + i] = (int32_t)i;
  l_val[E + i] = (e >= 0) ? (w ? w[e] : 1.f) : fill;
}

__device__ __forceinline__ float inv_sqrt_or_zero(float d) {
  const float v = 1.0f / sqrtf(d);  // deg.pow(-0.5)
  return isinf(v) ? 0.f : v;        // masked_fill_(== inf, 0)
}

// val[q] <- (dis[src] * w) * dis[dst]   for the slots of row i (dst = i)
__global__ __launch_bounds__(256) void k_gcn_normalize(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int64_t N,
                                                        const float* __restrict__ deg, float* val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float di = inv_sqrt_or_zero(deg[i]);
  for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) val[q] = (inv_sqrt_or_zero(deg[col[q]]) * val[q]) * di;
}

// expand a CSR back into a staged list with the roles swapped (for the transposed operator)
__global__ __launch_bounds__(256) void k_stage_transpose(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ val, int64_t N, int64_t L,
                                                          int32_t* l_dst, int32_t* l_src, float* l_val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
    l_dst[q] = col[q];
    l_src[q] = (int32_t)i;
    l_val[q] = val[q];
  }
}

// slots past the live range of a CSR (q >= rowptr[N]) are dropped from the staged list
__global__ __launch_bounds__(256) void k_stage_drop_tail(const int32_t* __restrict__ rowptr, int64_t N, int64_t L,
                                                          int32_t* l_dst, int32_t* l_src, float* l_val) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= L || q < rowptr[N]) return;
  l_dst[q] = (int32_t)N;
  l_src[q] = 0;
  l_val[q] = 0.f;
}

// ---- Chebyshev specifics
// Laplacian values for the edge slots (self-loops already dropped in l_dst) given deg = scatter_add(w, row).
// Staged list on entry: mode-1 staging (dst = row, src = col, val = w).  normalization: 0 None, 1 sym, 2 rw.
__global__ __launch_bounds__(256) void k_cheb_edge_vals(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                                         const float* __restrict__ deg, int normalization,
                                                         int transposed_flow, int32_t* l_dst, int32_t* l_src,
                                                         float* l_val) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = ei[e], c = ei[E + e];
  const bool bad = (r < 0 || r >= N || c < 0 || c >= N) || r == c;
  if (bad) { l_dst[e] = (int32_t)N; l_src[e] = 0; l_val[e] = 0.f; return; }
  const float w = l_val[e];
  float v;
  if (normalization == 0) v = -w;
  else if (normalization == 1) v = -((inv_sqrt_or_zero(deg[r]) * w) * inv_sqrt_or_zero(deg[c]));
  else { float di = 1.0f / deg[r]; if (isinf(di)) di = 0.f; v = -(di * w); }
  l_val[e] = v;
  // ChebConv: propagate(edge_index): out[col] += norm * x[row];  ChebConvAttention: transposed list
  l_dst[e] = transposed_flow ? (int32_t)r : (int32_t)c;
  l_src[e] = transposed_flow ? (int32_t)c : (int32_t)r;
}

__global__ __launch_bounds__(256) void k_cheb_diag(int64_t E, int64_t N, const float* __restrict__ deg,
                                                    int normalization, int32_t* l_dst, int32_t* l_src,
                                                    float* l_val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  l_dst[E + i] = (int32_t)i;
  l_src[E + i] = (int32_t)i;
  l_val[E + i] = normalization == 0 ? deg[i] : 1.f;
}

// single workgroup: scal[0] = 2 * max(l_val[0:L]) over live entries
__global__ __launch_bounds__(256) void k_cheb_lambda_auto(const int32_t* __restrict__ l_dst,
                                                           const float* __restrict__ l_val, int64_t L, int64_t N,
                                                           float* scal) {
  __shared__ float red[256];
  float m = -INFINITY;
  for (int64_t q = threadIdx.x; q < L; q += 256)
    if (l_dst[q] < (int32_t)N) m = fmaxf(m, l_val[q]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scal[0] = 2.0f * red[0];
}

__global__ __launch_bounds__(256) void k_set_scalar(float* scal, float v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) scal[0] = v;
}

// (2 * v) / lambda, inf -> 0; variant 0: "-1" folded into the diagonal slots; variant 1: N extra "-1" slots.
// batch != nullptr: lambda of a slot = lam_vec[batch[row]] with row = edge_index[0] of the Laplacian's entry
// (astgcn.py:97-98 / PyG ChebConv.__norm__: `lambda_max[batch[edge_index[0]]]` after get_laplacian appended the diagonal);
// a label outside [0, G) is counted in info[3] and its slots become NaN.
__global__ __launch_bounds__(256) void k_cheb_scale(int64_t E, int64_t N, const float* __restrict__ scal,
                                                     int variant, int32_t* l_dst, int32_t* l_src, float* l_val,
                                                     const int64_t* __restrict__ batch,
                                                     const float* __restrict__ lam_vec, int64_t G, int32_t* info) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= E + 2 * N) return;
  if (q < E + N) {
    float lam = scal[0];
    if (batch != nullptr && l_dst[q] < (int32_t)N) {
      const int64_t row = (q >= E) ? (q - E) : (variant == 1 ? (int64_t)l_dst[q] : (int64_t)l_src[q]);
      const int64_t b = batch[row];
      if (b < 0 || b >= G) { atomicAdd(info + 3, 1); lam = NAN; }
      else lam = lam_vec[b];
    }
    float v = (2.0f * l_val[q]) / lam;
    if (isinf(v) && v > 0.f) v = 0.f;  // masked_fill_(edge_weight == inf, 0)
    if (q >= E && variant == 0) v -= 1.0f;
    l_val[q] = v;
  } else {
    const int64_t i = q - E - N;
    if (variant == 1) { l_dst[q] = (int32_t)i; l_src[q] = (int32_t)i; l_val[q] = -1.0f; }
    else { l_dst[q] = (int32_t)N; l_src[q] = 0; l_val[q] = 0.f; }
  }
}

// ------------------------------------------------------------------------------------------------ host

inline dim3 g1(int64_t n) { return dim3((unsigned)(n > 0 ? pgt_cdiv(n, 256) : 1)); }

// staged list (l_dst, l_src, l_val)[0:L] -> CSR `out` (stable by dst; dst == N dropped past rowptr[N])
int build_csr(const Ws& w, int64_t L, int64_t N, const pgt_csr& out, pgt_stream_t stream) {
  dim3 block(256);
  PGT_LAUNCH(k_keys_from_list, g1(L), block, stream, w.l_dst, L, w.k_in, w.v_in);
  if (int e = sort_pairs(w, L, bits_for((uint64_t)N), stream)) return e;
  PGT_LAUNCH(k_rowptr, g1(N + 1), block, stream, w.k_out, L, N, out.rowptr);
  PGT_LAUNCH(k_gather_slots, g1(L), block, stream, w.perm, w.l_src, w.l_val, L, out.col, out.val);
  return pgt_check_launch("graph prep: build_csr");
}

int transpose_csr(const Ws& w, const pgt_csr& src, int64_t L, int64_t N, const pgt_csr& out,
                  pgt_stream_t stream) {
  dim3 block(256);
  PGT_LAUNCH(k_stage_transpose, g1(N), block, stream, src.rowptr, src.col, src.val, N, L, w.l_dst, w.l_src,
             w.l_val);
  PGT_LAUNCH(k_stage_drop_tail, g1(L), block, stream, src.rowptr, N, L, w.l_dst, w.l_src, w.l_val);
  return build_csr(w, L, N, out, stream);
}

bool csr_ok(const pgt_csr& c) { return c.rowptr && c.col && c.val; }

int common_checks(const char* what, const int64_t* ei, int64_t E, int64_t N, void* ws, size_t ws_bytes, Ws* w) {
  PGT_REQUIRE(E >= 0 && N >= 0, "%s: negative size", what);
  PGT_REQUIRE(N > 0 || E == 0, "%s: edges given for an empty node set", what);
  PGT_REQUIRE(E == 0 || ei != nullptr, "%s: null edge_index", what);
  PGT_REQUIRE(N < ((int64_t)1 << 31) - 2 && E + 2 * N < ((int64_t)1 << 31) - 2, "%s: graph exceeds int32 indexing",
              what);
  const size_t need = ws_layout(E, N, nullptr, nullptr);
  if (ws == nullptr || ws_bytes < need) {
    pgt_set_error("%s: workspace too small (%zu < %zu)", what, ws_bytes, need);
    return PGT_ERR_WORKSPACE;
  }
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  ws_layout(E, N, base, w);
  return PGT_OK;
}

}  // namespace

extern "C" size_t pgt_prep_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  return ws_layout(E, N, nullptr, nullptr);
}

extern "C" int pgt_dconv_prep(const int64_t* ei, const float* ew, int64_t E, int64_t N,
                              const pgt_dconv_graph* out, void* ws, size_t ws_bytes, pgt_stream_t stream) {
  Ws w;
  if (int e = common_checks("pgt_dconv_prep", ei, E, N, ws, ws_bytes, &w)) return e;
  PGT_REQUIRE(out && csr_ok(out->fwd_o) && csr_ok(out->fwd_i) && csr_ok(out->bwd_o) && csr_ok(out->bwd_i) &&
                  out->deg_out && out->deg_in && out->info,
              "pgt_dconv_prep: null output buffer");
  dim3 block(256);
  (void)hipMemsetAsync(out->info, 0, 4 * sizeof(int32_t), (hipStream_t)stream);
  PGT_LAUNCH(k_check_edges, g1(E), block, stream, ei, ew, E, N, out->info);

  // P_o by target: dst = col, src = row (edge order kept) ; deg_in = scatter_add(w, col)
  PGT_LAUNCH(k_stage_coo, g1(E), block, stream, ei, ew, E, N, 0, 0, w.l_dst, w.l_src, w.l_val);
  if (int e = build_csr(w, E, N, out->fwd_o, stream)) return e;
  PGT_LAUNCH(k_rowsum, g1(N), block, stream, out->fwd_o.rowptr, out->fwd_o.val, N, out->deg_in);
  // P_o^T by source: dst = row, src = col ; deg_out = scatter_add(w, row)
  PGT_LAUNCH(k_stage_coo, g1(E), block, stream, ei, ew, E, N, 1, 0, w.l_dst, w.l_src, w.l_val);
  if (int e = build_csr(w, E, N, out->bwd_o, stream)) return e;
  PGT_LAUNCH(k_rowsum, g1(N), block, stream, out->bwd_o.rowptr, out->bwd_o.val, N, out->deg_out);
  // norm_out = deg_out_inv[row]  (dcrnn.py:73)
  PGT_LAUNCH(k_dconv_val_by_src, g1(E), block, stream, out->fwd_o.col, E, out->deg_out, out->fwd_o.val);
  PGT_LAUNCH(k_dconv_val_by_row, g1(N), block, stream, out->bwd_o.rowptr, N, out->deg_out, out->bwd_o.val);

  // sigma = argsort(col * N + row)
  PGT_LAUNCH(k_dconv_sigma_keys, g1(E), block, stream, ei, E, N, w.k_in, w.v_in);
  if (int e = sort_pairs(w, E, bits_for((uint64_t)N * (uint64_t)N), stream)) return e;
  PGT_LAUNCH(k_count_dups, g1(E), block, stream, w.k_out, E, out->info);
  if (E > 0) (void)hipMemcpyAsync(w.sigma, w.perm, (size_t)E * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  // P_i and its transpose
  PGT_LAUNCH(k_dconv_stage_reverse, g1(E), block, stream, ei, w.sigma, E, N, out->deg_in, 0, w.l_dst, w.l_src,
             w.l_val);
  if (int e = build_csr(w, E, N, out->fwd_i, stream)) return e;
  PGT_LAUNCH(k_dconv_stage_reverse, g1(E), block, stream, ei, w.sigma, E, N, out->deg_in, 1, w.l_dst, w.l_src,
             w.l_val);
  if (int e = build_csr(w, E, N, out->bwd_i, stream)) return e;
  // (P_o and P_i hold every coefficient of the four operators: the transposes carry the same values)
  if (E > 0) {
    PGT_LAUNCH(k_count_nonfinite, g1(E), block, stream, out->fwd_o.val, E, out->info);
    PGT_LAUNCH(k_count_nonfinite, g1(E), block, stream, out->fwd_i.val, E, out->info);
  }
  return pgt_check_launch("pgt_dconv_prep");
}

extern "C" int pgt_gcn_prep(const int64_t* ei, const float* ew, int64_t E, int64_t N, int improved,
                            int add_self_loops, const pgt_sym_graph* out, void* ws, size_t ws_bytes,
                            pgt_stream_t stream) {
  // PyG >= 2.3 runs add_remaining_self_loops BEFORE `edge_weight = ones` (gcn_norm): without edge weights the new
  // loops carry no attribute and end up with weight 1 like every other edge, i.e. `improved` only acts on weighted input
  const float fill = (improved && ew != nullptr) ? 2.0f : 1.0f;
  Ws w;
  if (int e = common_checks("pgt_gcn_prep", ei, E, N, ws, ws_bytes, &w)) return e;
  PGT_REQUIRE(out && csr_ok(out->fwd) && csr_ok(out->bwd) && out->deg && out->info,
              "pgt_gcn_prep: null output buffer");
  dim3 block(256);
  (void)hipMemsetAsync(out->info, 0, 4 * sizeof(int32_t), (hipStream_t)stream);
  PGT_LAUNCH(k_check_edges, g1(E), block, stream, ei, ew, E, N, out->info);
  int64_t L = E;
  PGT_LAUNCH(k_stage_coo, g1(E), block, stream, ei, ew, E, N, 0, add_self_loops ? 1 : 0, w.l_dst, w.l_src, w.l_val);
  if (add_self_loops) {
    PGT_LAUNCH(k_fill_i32, g1(N), block, stream, w.node_i, N, (int32_t)-1);
    PGT_LAUNCH(k_gcn_loop_eid, g1(E), block, stream, ei, E, N, w.node_i);
    PGT_LAUNCH(k_gcn_stage_loops, g1(N), block, stream, ew, w.node_i, E, N, fill, w.l_dst,
               w.l_src, w.l_val);
    L = E + N;
  }
  if (int e = build_csr(w, L, N, out->fwd, stream)) return e;
  PGT_LAUNCH(k_rowsum, g1(N), block, stream, out->fwd.rowptr, out->fwd.val, N, out->deg);
  PGT_LAUNCH(k_gcn_normalize, g1(N), block, stream, out->fwd.rowptr, out->fwd.col, N, out->deg, out->fwd.val);
  if (int e = transpose_csr(w, out->fwd, L, N, out->bwd, stream)) return e;
  return pgt_check_launch("pgt_gcn_prep");
}

namespace {
int cheb_prep_impl(const char* what, const int64_t* ei, const float* ew, int64_t E, int64_t N, int normalization,
                   float lambda_max, const int64_t* batch, const float* lam_vec, int64_t G, int variant,
                   const pgt_sym_graph* out, void* ws, size_t ws_bytes, pgt_stream_t stream) {
  Ws w;
  if (int e = common_checks(what, ei, E, N, ws, ws_bytes, &w)) return e;
  PGT_REQUIRE(out && csr_ok(out->fwd) && csr_ok(out->bwd) && out->deg && out->info, "%s: null output buffer", what);
  PGT_REQUIRE(normalization >= 0 && normalization <= 2, "%s: normalization must be 0 (None), 1 (sym) or 2 (rw)", what);
  PGT_REQUIRE(variant == 0 || variant == 1, "%s: variant must be 0 or 1", what);
  dim3 block(256);
  (void)hipMemsetAsync(out->info, 0, 4 * sizeof(int32_t), (hipStream_t)stream);
  PGT_LAUNCH(k_check_edges, g1(E), block, stream, ei, ew, E, N, out->info);
  // deg = scatter_add(w, row) over the non-loop edges (get_laplacian)
  PGT_LAUNCH(k_stage_coo, g1(E), block, stream, ei, ew, E, N, 1, 1, w.l_dst, w.l_src, w.l_val);
  if (int e = build_csr(w, E, N, out->bwd, stream)) return e;  // scratch use of out->bwd
  PGT_LAUNCH(k_rowsum, g1(N), block, stream, out->bwd.rowptr, out->bwd.val, N, out->deg);
  // restage with the Laplacian coefficients
  PGT_LAUNCH(k_stage_coo, g1(E), block, stream, ei, ew, E, N, 1, 1, w.l_dst, w.l_src, w.l_val);
  PGT_LAUNCH(k_cheb_edge_vals, g1(E), block, stream, ei, E, N, out->deg, normalization, variant, w.l_dst, w.l_src,
             w.l_val);
  PGT_LAUNCH(k_cheb_diag, g1(N), block, stream, E, N, out->deg, normalization, w.l_dst, w.l_src, w.l_val);
  if (batch != nullptr) {
    PGT_LAUNCH(k_set_scalar, dim3(1), block, stream, w.scal, 1.0f);     // (unused: every live slot has a label)
  } else if (isnan(lambda_max)) {
    if (normalization == 1) {
      PGT_LAUNCH(k_set_scalar, dim3(1), block, stream, w.scal, 2.0f);
    } else {
      PGT_LAUNCH(k_cheb_lambda_auto, dim3(1), block, stream, w.l_dst, w.l_val, E + N, N, w.scal);
    }
  } else {
    PGT_LAUNCH(k_set_scalar, dim3(1), block, stream, w.scal, lambda_max);
  }
  const int64_t L = E + 2 * N;
  PGT_LAUNCH(k_cheb_scale, g1(L), block, stream, E, N, w.scal, variant, w.l_dst, w.l_src, w.l_val, batch, lam_vec, G,
             out->info);
  if (int e = build_csr(w, L, N, out->fwd, stream)) return e;
  if (int e = transpose_csr(w, out->fwd, L, N, out->bwd, stream)) return e;
  return pgt_check_launch(what);
}
}  // namespace

extern "C" int pgt_cheb_prep(const int64_t* ei, const float* ew, int64_t E, int64_t N, int normalization,
                             float lambda_max, int variant, const pgt_sym_graph* out, void* ws, size_t ws_bytes,
                             pgt_stream_t stream) {
  PGT_REQUIRE(isnan(lambda_max) || lambda_max != 0.f, "pgt_cheb_prep: lambda_max must be non-zero");
  return cheb_prep_impl("pgt_cheb_prep", ei, ew, E, N, normalization, lambda_max, nullptr, nullptr, 0, variant, out, ws,
                        ws_bytes, stream);
}

extern "C" int pgt_cheb_prep_graphs(const int64_t* ei, const float* ew, int64_t E, int64_t N, int normalization,
                                    const int64_t* batch, const float* lambda_max, int64_t n_graphs, int variant,
                                    const pgt_sym_graph* out, void* ws, size_t ws_bytes, pgt_stream_t stream) {
  PGT_REQUIRE(n_graphs > 0 && lambda_max != nullptr, "pgt_cheb_prep_graphs: no lambda_max values");
  PGT_REQUIRE(N == 0 || batch != nullptr, "pgt_cheb_prep_graphs: null batch vector");
  return cheb_prep_impl("pgt_cheb_prep_graphs", ei, ew, E, N, normalization, NAN, batch, lambda_max, n_graphs, variant,
                        out, ws, ws_bytes, stream);
}
