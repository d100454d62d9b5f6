// LAB HARNESS (not shipped): the PRODUCT ELLW kernel (csrc/spmm.hip through the C ABI) at the north-star shape, issued
// as plain stream launches and as one hipGraph, next to the lab kernel's numbers (lab/ellw_lab.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define LAB_HAS_SPMM
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/spmm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define PK(x) do { int rc_ = (x); if (rc_ != 0) { printf("pgt error %d: %s at %d\n", rc_, pgt_last_error(), __LINE__); exit(1); } } while (0)

struct Graph { std::vector<int> rp, col; std::vector<float> val; };
static Graph local_graph(int n, int deg, int window) {
  Graph g; g.rp.resize(n + 1); g.col.resize((size_t)n * deg); g.val.resize((size_t)n * deg);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  std::vector<float> degout(n, 0.f), w((size_t)n * deg);
  for (int i = 0; i < n; ++i) {
    g.rp[i] = i * deg;
    int c[64];
    for (int k = 0; k < deg; ++k) {
      int v; bool dup;
      do { int off = (int)(rnd() % window) - window / 2; if (off >= 0) off += 1; v = ((i + off) % n + n) % n; dup = false; for (int j = 0; j < k; ++j) dup |= c[j] == v; } while (dup);
      c[k] = v;
    }
    std::sort(c, c + deg);
    for (int k = 0; k < deg; ++k) { g.col[(size_t)i * deg + k] = c[k]; w[(size_t)i * deg + k] = 0.5f + (rnd() % 1000) / 1000.f; degout[c[k]] += w[(size_t)i * deg + k]; }
  }
  g.rp[n] = n * deg;
  for (size_t q = 0; q < g.col.size(); ++q) g.val[q] = 1.f / degout[g.col[q]];
  return g;
}

int main(int argc, char** argv) {
  const int n = 200000, F = 64, PAIRS = 6;
  const int deg = argc > 1 ? atoi(argv[1]) : 8;
  Graph g = local_graph(n, deg, 64);
  const int64_t nnz = (int64_t)g.col.size();
  int *rp, *col; float* val;
  CK(hipMalloc(&rp, (n + 1) * 4)); CK(hipMalloc(&col, nnz * 4)); CK(hipMalloc(&val, nnz * 4));
  CK(hipMemcpy(rp, g.rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(col, g.col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(val, g.val.data(), nnz * 4, hipMemcpyHostToDevice));
  float *X[PAIRS], *Y[PAIRS];
  std::vector<float> hx((size_t)n * F);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  for (int p = 0; p < PAIRS; ++p) {
    CK(hipMalloc(&X[p], (size_t)n * F * 4)); CK(hipMalloc(&Y[p], (size_t)n * F * 4));
    CK(hipMemcpy(X[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double alg = 4.0 * (n + 1) + 8.0 * nnz + 8.0 * n * F;
  auto timeit = [&](const char* name, auto fn) {
    for (int i = 0; i < 2 * PAIRS; ++i) fn(i % PAIRS);
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      const int reps = 10 * PAIRS;
      for (int i = 0; i < reps; ++i) fn(i % PAIRS);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms * 1e3 / reps); sum += ms * 1e3 / reps;
    }
    printf("%-64s %7.2f us (mean %6.2f)  %.3f of 8 TB/s\n", name, best, sum / 3, alg / best / 1e3 / 8000);
    fflush(stdout);
  };
  // build the layout through the C ABI
  for (int cfg : {1, 2}) {
  pgt_tune("spmm_ellw_cfg", cfg);
  pgt_ellw op; memset(&op, 0, sizeof(op));
  op.halo = 32;
  PK(pgt_ellw_plan(n, 32, deg, 1, &op.tile_rows, &op.width, &op.config, &op.n_tiles, &op.far_rows));
  const size_t total = (size_t)op.n_tiles * op.tile_rows * op.width;
  uint16_t* slots; float *vals, *scale; int32_t* info;
  CK(hipMalloc(&slots, total * 2)); CK(hipMalloc(&vals, total * 4)); CK(hipMalloc(&scale, n * 4)); CK(hipMalloc(&info, 16));
  PK(pgt_ellw_build(rp, col, val, n, nnz, &op, slots, vals, scale, nullptr, nullptr, info, st));
  int32_t hinfo[4]; CK(hipMemcpyAsync(hinfo, info, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  printf("cfg %d deg %d: plan: %lld tiles of %d rows x %d slots; far %d, scale mismatches %d, overflow rows %d; algorithmic %.1f MB\n",
         cfg, deg, (long long)op.n_tiles, op.tile_rows, op.width, hinfo[0], hinfo[1], hinfo[2], alg / 1e6);
  op.slots = slots;
  for (int mode = 0; mode < 2; ++mode) {
    op.scale = mode == 0 ? scale : nullptr;
    op.vals = mode == 0 ? nullptr : vals;
    char nm[96];
    snprintf(nm, 96, "cfg %d product kernel, mode %d (%s), stream launches", cfg, mode, mode == 0 ? "scale table" : "per-slot vals");
    timeit(nm, [&](int p) { PK(pgt_spmm_ellw_f32(&op, rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st)); });
    // the same 60 launches as one hipGraph (what bench.py replays)
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 60; ++i) PK(pgt_spmm_ellw_f32(&op, rp, col, val, n, X[i % PAIRS], F, Y[i % PAIRS], F, nullptr, 0, 1.f, 0.f, F, st));
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms * 1e3 / 60); sum += ms * 1e3 / 60;
    }
    snprintf(nm, 96, "cfg %d product kernel, mode %d, 60 launches as one hipGraph", cfg, mode);
    printf("%-64s %7.2f us (mean %6.2f)  %.3f of 8 TB/s\n", nm, best, sum / 3, alg / best / 1e3 / 8000);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  }
  op.scale = scale; op.vals = nullptr;
  timeit("  same, mode 0, epilogue 2*P*X - T", [&](int p) { PK(pgt_spmm_ellw_f32(&op, rp, col, val, n, X[p], F, Y[p], F, X[(p + 1) % PAIRS], F, 2.f, -1.f, F, st)); });
  CK(hipFree(slots)); CK(hipFree(vals)); CK(hipFree(scale)); CK(hipFree(info));
  }
  pgt_tune("spmm_ellw_cfg", 0);
  // CSR row tiles for reference
  timeit("CSR row tiles (pgt_spmm_csr_f32)", [&](int p) { PK(pgt_spmm_csr_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st)); });
  return 0;
}
