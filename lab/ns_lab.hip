// LAB HARNESS (not shipped, not part of the library): times the N = 200 000, F = 64 aggregation kernels of
// csrc/spmm.hip directly through the C ABI, next to a plain float4 copy of the same size, with rotating buffers,
// and dumps an in-kernel timeline (PGT_TRACE) of the workgroups.   Build: see scripts/build_lab.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define PGT_TRACE 1
__device__ long long* g_trace_buf = nullptr;
#define PGT_TRACE_MARK(slot)                                                              \
  do {                                                                                    \
    if (g_trace_buf != nullptr && threadIdx.x == 0)                                       \
      g_trace_buf[(size_t)blockIdx.x * 16 + (slot)] = (long long)wall_clock64();          \
  } while (0)

void pgt_gemm_set_force_small(int) {}
void pgt_gemm_set_small_fill(int) {}
void pgt_gemm_set_tn_fullk(int) {}
void pgt_gemm_set_db(int) {}
void pgt_gemm_set_db64(int) {}
void pgt_slab_set_pairs(int) {}
void pgt_gemm_set_tn_pipe(int) {}
void pgt_gemm_set_skinny(int) {}
void pgt_gemm_set_dbp(int) {}
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/spmm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

// read-only stream: every lane sums its float4s; one store per workgroup (keeps the loads alive)
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = a[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1.2345e30f) out[blockIdx.x] = s;
}
// write-only stream
__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

struct Graph { std::vector<int> rp, col; std::vector<float> val; };

static Graph local_graph(int n, int deg, int window, bool uniform) {
  Graph g; g.rp.resize(n + 1); g.col.resize((size_t)n * deg); g.val.resize((size_t)n * deg);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < n; ++i) {
    g.rp[i] = i * deg;
    int c[64];
    for (int k = 0; k < deg; ++k) {
      int v;
      bool dup;
      do {
        if (uniform) v = (int)(rnd() % n);
        else { int off = (int)(rnd() % window) - window / 2; if (off >= 0) off += 1; v = ((i + off) % n + n) % n; }
        dup = false;
        for (int j = 0; j < k; ++j) dup |= c[j] == v;
      } while (dup);
      c[k] = v;
    }
    std::sort(c, c + deg);
    for (int k = 0; k < deg; ++k) { g.col[(size_t)i * deg + k] = c[k]; g.val[(size_t)i * deg + k] = 0.5f + (rnd() % 1000) / 1000.f; }
  }
  g.rp[n] = n * deg;
  return g;
}

int main(int argc, char** argv) {
  const int n = 200000, F = 64, PAIRS = 6;
  int deg = argc > 1 ? atoi(argv[1]) : 8;
  bool uniform = argc > 2 && atoi(argv[2]) != 0;
  Graph g = deg == 1 ? Graph() : local_graph(n, deg, 64, uniform);
  if (deg == 1) { g.rp.resize(n + 1); g.col.resize(n); g.val.assign(n, 1.f); for (int i = 0; i <= n; ++i) g.rp[i] = i; for (int i = 0; i < n; ++i) g.col[i] = i; }
  int *rp, *col; float* val;
  CK(hipMalloc(&rp, (n + 1) * 4)); CK(hipMalloc(&col, g.col.size() * 4)); CK(hipMalloc(&val, g.val.size() * 4));
  CK(hipMemcpy(rp, g.rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(col, g.col.data(), g.col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(val, g.val.data(), g.val.size() * 4, hipMemcpyHostToDevice));
  float *X[PAIRS], *Y[PAIRS];
  std::vector<float> hx((size_t)n * F);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  for (int p = 0; p < PAIRS; ++p) {
    CK(hipMalloc(&X[p], (size_t)n * F * 4)); CK(hipMalloc(&Y[p], (size_t)n * F * 4));
    CK(hipMemcpy(X[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double alg = 4.0 * (n + 1) + 8.0 * g.col.size() + 8.0 * n * F;
  auto timeit = [&](const char* name, auto fn, double bytes) {
    for (int i = 0; i < 2 * PAIRS; ++i) fn(i % PAIRS);
    CK(hipEventRecord(e0, st));
    const int reps = 10 * PAIRS;
    for (int i = 0; i < reps; ++i) fn(i % PAIRS);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps;
    printf("%-34s %8.2f us   %7.1f GB/s   (%.3f of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
    return us;
  };
  printf("deg %d %s, algorithmic %.1f MB\n", deg, uniform ? "uniform" : "local(+-32)", alg / 1e6);
  for (int blocks : {1024, 2048, 4096})
    timeit(blocks == 1024 ? "copy 51.2MB->51.2MB g=1024" : blocks == 2048 ? "copy g=2048" : "copy g=4096",
           [&](int p) { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)X[p], (float4*)Y[p], (size_t)n * F / 4); }, 8.0 * n * F);
  for (int blocks : {1024, 2048, 4096})
    timeit(blocks == 1024 ? "read-only 51.2MB g=1024" : blocks == 2048 ? "read-only g=2048" : "read-only g=4096",
           [&](int p) { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)X[p], Y[p], (size_t)n * F / 4); }, 4.0 * n * F);
  timeit("write-only 51.2MB g=2048", [&](int p) { hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, st, (float4*)Y[p], (size_t)n * F / 4); }, 4.0 * n * F);
  for (int qb : {7, 6, 4, 3}) {
    pgt_tune("spmm_quad", 1); pgt_tune("spmm_quad_blocks", qb);
    char nm[64]; snprintf(nm, 64, "quad persistent, %d wg/CU", qb);
    timeit(nm, [&](int p) { pgt_spmm_csr_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
  }
  pgt_tune("spmm_quad", 0);
  for (int rows : {64, 32}) {
    pgt_tune("spmm_tile_rows", rows);
    timeit(rows == 64 ? "plain tile TR=64" : "plain tile TR=32", [&](int p) { pgt_spmm_csr_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
  }
  pgt_tune("spmm_tile_rows", 32);
  for (int nt : {0, 1, 0, 1}) {
    pgt_tune("spmm_tile_nt", nt);
    timeit(nt ? "plain tile TR=32, streaming Y stores" : "plain tile TR=32, plain Y stores", [&](int p) { pgt_spmm_csr_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
  }
  pgt_tune("spmm_tile_nt", 0);
  pgt_tune("spmm_tile_rows", 64);
  for (int tpw : {1}) {
    pgt_tune("spmm_band_cu", 4); pgt_tune("spmm_wtile_tpw", tpw);
    char nm[64]; snprintf(nm, 64, "window tile TR=32 tiles/wg=%d", tpw);
    timeit(nm, [&](int p) { pgt_spmm_csr_band_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, 32, st); }, alg);
  }
  pgt_tune("spmm_wtile_tpw", 0);
  for (int cu : {3, 1}) {
    pgt_tune("spmm_band_cu", cu);
    char nm[64]; snprintf(nm, 64, cu >= 3 ? "window tile (band_cu=%d)" : "band per-CU workgroup x%d", cu);
    timeit(nm, [&](int p) { pgt_spmm_csr_band_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, 32, st); }, alg);
  }
  pgt_tune("spmm_band_cu", 0);
  for (int k : {3}) for (int ch : {1}) {
    pgt_tune("spmm_band_blocks", k); pgt_tune("spmm_band_xcd", ch);
    char nm[64]; snprintf(nm, 64, "band halo=32 k=%d xcd=%d", k, ch);
    timeit(nm, [&](int p) { pgt_spmm_csr_band_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, 32, st); }, alg);
  }
  pgt_tune("spmm_band_blocks", 3); pgt_tune("spmm_band_xcd", 1);
  // ---- timeline of one band launch and one plain launch
  long long* tr; const size_t TRN = 4096 * 16;
  CK(hipMalloc(&tr, TRN * 8));
  std::vector<long long> h(TRN);
  for (int which = 0; which < 2; ++which) {
    CK(hipMemset(tr, 0, TRN * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &tr, sizeof(tr)));
    CK(hipDeviceSynchronize());
    pgt_tune("spmm_band_cu", 1);
    if (which == 0) pgt_spmm_csr_band_f32(rp, col, val, n, X[3], F, Y[3], F, nullptr, 0, 1.f, 0.f, F, 32, st);
    else pgt_spmm_csr_f32(rp, col, val, n, X[4], F, Y[4], F, nullptr, 0, 1.f, 0.f, F, st);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), tr, TRN * 8, hipMemcpyDeviceToHost));
    long long t0 = -1, t1 = 0; int nb = 0;
    for (size_t b = 0; b < 4096; ++b) if (h[b * 16]) { nb++; if (t0 < 0 || h[b * 16] < t0) t0 = h[b * 16]; for (int s = 0; s < 16; ++s) t1 = std::max(t1, h[b * 16 + s]); }
    printf("%s: %d blocks traced, span %.2f us (100 MHz wall clock)\n", which == 0 ? "band" : "plain", nb, (t1 - t0) / 100.0);
    // per-slot mean offset from kernel start and mean duration
    for (int s = 0; s < 16; ++s) {
      double sum = 0, mx = 0, mn = 1e18; int c = 0;
      for (size_t b = 0; b < 4096; ++b) if (h[b * 16] && h[b * 16 + s]) { double v = (h[b * 16 + s] - t0) / 100.0; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); c++; }
      if (c) printf("   mark %2d: n=%4d  mean %7.2f us  min %7.2f  max %7.2f\n", s, c, sum / c, mn, mx);
    }
    // histogram of block start times
    int hist[16] = {0};
    for (size_t b = 0; b < 4096; ++b) if (h[b * 16]) { int k = (int)((h[b * 16] - t0) / 100.0 / 3.0); hist[std::min(k, 15)]++; }
    printf("   block starts per 3us bin:");
    for (int k = 0; k < 16; ++k) printf(" %d", hist[k]);
    printf("\n");
  }
  long long* nul = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &nul, sizeof(nul)));
  return 0;
}
