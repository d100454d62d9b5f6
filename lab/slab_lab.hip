// LAB HARNESS (not shipped): takes the LDS-resident diffusion-stack kernel (csrc/dconv_slab.hip, whole-sample forward) apart
// at the benchmark shape: launch time with stores / gathers / prefetch loads removed, and a per-workgroup phase timeline.
//   ./lab/slab_lab [B = 1024] [C = 66]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ int g_lab_skip = 0;
#define PGT_LAB_SKIP(bit) ((g_lab_skip & (bit)) != 0)
__device__ long long* g_trace_buf = nullptr;
constexpr int TR_ITERS = 8, TR_SLOTS = 5;
#define PGT_TRACE_MARK2(iter, slot)                                                                                   \
  do {                                                                                                                \
    if (g_trace_buf != nullptr && threadIdx.x == 0 && (iter) < TR_ITERS)                                              \
      g_trace_buf[((size_t)blockIdx.x * TR_ITERS + (iter)) * TR_SLOTS + (slot)] = (long long)wall_clock64();          \
  } while (0)

#define LAB_HAS_SLAB
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/dconv_slab.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, C = argc > 2 ? atoi(argv[2]) : 66, N = 207, E = 1515, K = 3;
  const int quad = argc > 3 ? atoi(argv[3]) : 1;     // 1: quad-layout kernels (C = 64 / 66), 0: pair layout
  // a METR-LA-like operator pair: every row 4 .. 11 slots, random sources
  std::vector<int32_t> rp(N + 1, 0), col;
  std::vector<float> val;
  uint32_t seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (int i = 0; i < N; ++i) {
    int len = (i == N - 1) ? E - rp[i] : 4 + (int)(rnd() % 8);
    if (rp[i] + len > E - (N - 1 - i) * 4) len = std::max(1, E - (N - 1 - i) * 4 - rp[i]);
    rp[i + 1] = rp[i] + len;
    for (int q = 0; q < len; ++q) { col.push_back((int32_t)(rnd() % N)); val.push_back(1.f / len); }
  }
  const int nnz = rp[N];
  int32_t *d_rp, *d_col; float* d_val;
  CK(hipMalloc(&d_rp, (N + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 4));
  CK(hipMemcpy(d_rp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), nnz * 4, hipMemcpyHostToDevice));
  const size_t seg = (size_t)B * N * C;
  float* TS;
  CK(hipMalloc(&TS, 5 * seg * 4));
  std::vector<float> h(seg);
  for (size_t i = 0; i < seg; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  CK(hipMemcpy(TS, h.data(), seg * 4, hipMemcpyHostToDevice));
  pgt_csr op{d_rp, d_col, d_val};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pgt_tune("slab_split", 0);     // the whole-sample kernels
  pgt_tune("slab_quad", quad);
  auto run = [&]() {
    int rc = pgt_dconv_stack_slab_f32(&op, &op, nnz, nnz, N, B, C, K, TS, (int64_t)seg, st);
    if (rc) { printf("launch failed: %s\n", pgt_last_error()); exit(1); }
  };
  auto timeit = [&](const char* name, int skip) {
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_skip), &skip, sizeof(int)));
    for (int i = 0; i < 3; ++i) run();
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = 5.0 * 4 * seg;
    printf("%-52s %8.2f us   (%.3f of 8 TB/s on the full launch's bytes)\n", name, us, bytes / us / 1e3 / 8000.0);
  };
  printf("B = %d, C = %d, N = %d, nnz = %d, %s layout\n", B, C, N, nnz, quad ? "quad" : "pair");
  timeit("whole kernel", 0);
  timeit("no global stores", 1);
  timeit("no gathers (copies instead)", 2);
  timeit("no prefetch loads", 4);
  timeit("no stores, no gathers", 3);
  timeit("no stores, no gathers, no prefetch", 7);
  timeit("no gathers, no prefetch (stores only)", 6);
  // phase timeline of the whole kernel: wall_clock64 ticks (100 MHz) at [loop top, T0 in LDS, hop 1 done, T1i in LDS, hop 2 done]
  const int nwg = std::min(B, 256);
  long long* d_tr;
  CK(hipMalloc(&d_tr, (size_t)nwg * TR_ITERS * TR_SLOTS * 8));
  CK(hipMemset(d_tr, 0, (size_t)nwg * TR_ITERS * TR_SLOTS * 8));
  int zero = 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_skip), &zero, sizeof(int)));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &d_tr, sizeof(d_tr)));
  run();
  CK(hipStreamSynchronize(st));
  std::vector<long long> tr((size_t)nwg * TR_ITERS * TR_SLOTS);
  CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
  long long t_min = 1LL << 62;
  for (auto v : tr) if (v > 0) t_min = std::min(t_min, v);
  const int iters = std::min(TR_ITERS, (B + nwg - 1) / nwg);
  for (int wg : {0, 1, 7, 100, 255}) {
    if (wg >= nwg) continue;
    printf("wg %3d:", wg);
    for (int it = 0; it < iters; ++it) {
      printf("  |");
      for (int s = 0; s < TR_SLOTS; ++s) printf(" %6.2f", (tr[((size_t)wg * TR_ITERS + it) * TR_SLOTS + s] - t_min) * 0.01);
    }
    printf("   (us since the first mark)\n");
  }
  // mean phase lengths over all workgroups and iterations
  double ph[TR_SLOTS] = {0}; int cnt = 0;
  for (int wg = 0; wg < nwg; ++wg)
    for (int it = 0; it < iters; ++it) {
      const long long* p = &tr[((size_t)wg * TR_ITERS + it) * TR_SLOTS];
      if (p[4] == 0) continue;
      for (int s = 1; s < TR_SLOTS; ++s) ph[s] += (p[s] - p[s - 1]) * 0.01;
      if (it + 1 < iters && p[TR_SLOTS] != 0) ph[0] += (p[TR_SLOTS] - p[4]) * 0.01;
      ++cnt;
    }
  printf("mean phase (us): wait+LDS store %.2f | hop 1 %.2f | LDS swap %.2f | hop 2 %.2f | loop back %.2f   over %d samples\n",
         ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, ph[0] / cnt, cnt);
  return 0;
}
