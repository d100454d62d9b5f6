// LAB HARNESS (not shipped): times pgt_gemm_f32 / pgt_gemm_tn_acc_f32 on the DCRNN shapes and dumps a workgroup timeline.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define PGT_TRACE 1
__device__ long long* g_trace_buf = nullptr;
#define PGT_TRACE_MARK(slot)                                                              \
  do {                                                                                    \
    if (g_trace_buf != nullptr && threadIdx.x == 0)                                       \
      g_trace_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = (long long)wall_clock64(); \
  } while (0)

__device__ int g_lab_dbg = 0;
#define PGT_LAB_KT(kt) (g_lab_dbg == 1 ? 0 : (kt))
#define PGT_LAB_SKIP_EPI() (g_lab_dbg == 2)
#define LAB_HAS_GEMM
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/gemm.hip"
// gemm.hip routes tall products to csrc/gemm_bx.hip; this harness links without it (its own copy of the kernels below)
int pgt_gemm_bx_launch(const PgtGemmArgs&, pgt_stream_t) { return 0; }
int pgt_gemm_bx_tn_plan(const PgtTnArgs&, int64_t*) { return 0; }
int pgt_gemm_bx_tn_launch(const PgtTnArgs&, pgt_stream_t) { return PGT_ERR_INVALID; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pure matrix-pipe loop: W wavefronts per SIMD, 4 independent accumulators, no memory traffic (achievable MFMA rate)
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  pgt_f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = PGT_MFMA_32x32x2(a, b, acc[i]);
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 211968, S = 5, C = 66, K = S * C;
  float *A, *W, *Cout, *G, *dW, *bias;
  CK(hipMalloc(&A, (size_t)S * M * C * 4)); CK(hipMalloc(&W, (size_t)640 * 128 * 4)); CK(hipMalloc(&Cout, (size_t)M * 128 * 4));
  CK(hipMalloc(&G, (size_t)S * M * C * 4)); CK(hipMalloc(&dW, (size_t)K * 128 * 4)); CK(hipMalloc(&bias, 128 * 4));
  std::vector<float> h((size_t)S * M * C);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)640 * 128 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Cout, h.data(), (size_t)M * 128 * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 512)); CK(hipMemset(dW, 0, (size_t)K * 128 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto fn, double flop) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps;
    printf("%-40s %9.2f us  %7.2f TF  (%.3f of 157.3)\n", name, us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  };
  for (int wps : {1, 2, 4}) {
    const int iters = 4096;
    auto fn = [&]() { hipLaunchKernelGGL(mfma_peak_kernel, dim3(256 * wps), dim3(256), 0, st, bias, iters); };
    for (int i = 0; i < 20; ++i) fn();     // also warms the clocks up before anything is timed
    char nm[80]; snprintf(nm, 80, "pure MFMA 32x32x2 f32, %d wave/SIMD", wps);
    timeit(nm, fn, 256.0 * wps * 4 * iters * 4 * 4096.0);
  }
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  pgt_tune("gemm_db64", argc > 4 ? atoi(argv[4]) : 1);
  if (argc > 5) pgt_tune("gemm_db", atoi(argv[5]));
  const int dbg = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_dbg), &dbg, sizeof(int)));
  if (only == 9) {   // one-tile vs persistent deferred-store schedule on the shapes of the training step
    float *ZR, *H, *XHR;
    CK(hipMalloc(&ZR, (size_t)M * 128 * 4)); CK(hipMalloc(&H, (size_t)M * 64 * 4)); CK(hipMalloc(&XHR, (size_t)M * C * 4));
    CK(hipMemset(H, 0, (size_t)M * 64 * 4));
    for (int stg : {0, 1, 0, 1}) {
      pgt_tune("gemm_dbp", stg);
      char nm[80];
      snprintf(nm, 80, "dbp %d  NT [M,64]->256 cols (66-seg)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 64, 0, 1, 64, W, 1, 64, G + 2, C, (int64_t)M * C, 64, nullptr, M, 256, 0, st); }, 2.0 * M * 64 * 256);
      snprintf(nm, 80, "dbp %d  NT [M,128]->256 cols (66-seg)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 128, 0, 1, 128, W, 1, 128, G + 2, C, (int64_t)M * C, 64, nullptr, M, 256, 0, st); }, 2.0 * M * 128 * 256);
      snprintf(nm, 80, "dbp %d  NT [M,64]->256 cols (64-seg, aligned)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 64, 0, 1, 64, W, 1, 64, G, 64, (int64_t)M * 64, 64, nullptr, M, 256, 0, st); }, 2.0 * M * 64 * 256);
      snprintf(nm, 80, "dbp %d  NT [M,128]->256 cols (64-seg, aligned)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 128, 0, 1, 128, W, 1, 128, G, 64, (int64_t)M * 64, 64, nullptr, M, 256, 0, st); }, 2.0 * M * 128 * 256);
      snprintf(nm, 80, "dbp %d  NT [M,64]->320 cols (64-seg, aligned)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 64, 0, 1, 64, W, 1, 64, G, 64, (int64_t)M * 64, 64, nullptr, M, 320, 0, st); }, 2.0 * M * 64 * 320);
      snprintf(nm, 80, "dbp %d  NT [M,64]->256 cols (plain 256-wide rows)", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 64, 0, 1, 64, W, 1, 64, G, 256, 0, 256, nullptr, M, 256, 0, st); }, 2.0 * M * 64 * 256);
      snprintf(nm, 80, "dbp %d  NT [M,128]->64 cols", stg);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, 128, 0, 1, 128, W, 1, 128, G + 2, C, (int64_t)M * C, 64, nullptr, M, 64, 0, st); }, 2.0 * M * 128 * 64);
      snprintf(nm, 80, "dbp %d  NN [M,330]->128", stg);
      timeit(nm, [&]() { pgt_gemm_f32(A, C, (int64_t)M * C, S, C, W, 128, 1, Cout, 128, 0, 128, bias, M, 128, 0, st); }, 2.0 * M * K * 128);
      snprintf(nm, 80, "dbp %d  NN+zr [M,330]->128 fused", stg);
      timeit(nm, [&]() { pgt_gemm_gru_zr_f32(A, C, (int64_t)M * C, S, C, W, 128, 1, bias, ZR, H, 64, XHR, C, 2, M, 64, st); }, 2.0 * M * K * 128);
      snprintf(nm, 80, "dbp %d  NN+h [M,330]->64 fused", stg);
      timeit(nm, [&]() { pgt_gemm_gru_h_f32(A, C, (int64_t)M * C, S, C, W, 64, 1, bias, Cout, ZR, H, 64, XHR, 64, nullptr, nullptr, 0, M, 64, st); }, 2.0 * M * K * 64);
    }
    return 0;
  }
  for (int N : {128, 64}) {
    char nm[80];
    for (int db : {0, 1}) {
      if (only >= 0 && db != only) continue;
      pgt_tune("gemm_db", db ? (argc > 5 ? atoi(argv[5]) : 1) : 0);
      snprintf(nm, 80, "NN  [M,330]x[330,%d] seg A  db=%d", N, db);
      timeit(nm, [&]() { pgt_gemm_f32(A, C, (int64_t)M * C, S, C, W, N, 1, Cout, N, 0, N, bias, M, N, 0, st); }, 2.0 * M * K * N);
      snprintf(nm, 80, "NT  [M,%d]x[%d,330] -> seg C  db=%d", N, N, db);
      timeit(nm, [&]() { pgt_gemm_f32(Cout, N, 0, 1, N, W, 1, N, G, C, (int64_t)M * C, C, nullptr, M, K, 0, st); }, 2.0 * M * K * N);
      // second layer: five 128-wide terms viewed inside the same buffers (M/2 rows so the extents fit)
      snprintf(nm, 80, "NN  [M/2,640]x[640,%d] seg A  db=%d", N, db);
      timeit(nm, [&]() { pgt_gemm_f32(A, 128, (int64_t)(M / 2) * 128, 5, 128, W, N, 1, Cout, N, 0, N, bias, M / 2, N, 0, st); }, 2.0 * (M / 2) * 640 * N);
    }
    pgt_tune("gemm_db", 1);
    snprintf(nm, 80, "TN  dW[330,%d] whole-K", N);
    pgt_tune("gemm_tn_fullk", 1);
    timeit(nm, [&]() { pgt_gemm_tn_acc_f32(A, C, (int64_t)M * C, S, C, Cout, N, dW, N, bias, M, N, st); }, 2.0 * M * K * N);
    snprintf(nm, 80, "TN  dW[330,%d] k-tiled", N);
    pgt_tune("gemm_tn_fullk", 0);
    timeit(nm, [&]() { pgt_gemm_tn_acc_f32(A, C, (int64_t)M * C, S, C, Cout, N, dW, N, bias, M, N, st); }, 2.0 * M * K * N);
    pgt_tune("gemm_tn_fullk", 1);
  }
  {  // weight gradient at the size of the training step: all 12 time steps in one launch
    const int64_t Mb = 12 * (int64_t)M;
    float *Ab, *Gb;
    CK(hipMalloc(&Ab, (size_t)S * Mb * C * 4)); CK(hipMalloc(&Gb, (size_t)Mb * 128 * 4));
    for (int j = 0; j < 12; ++j) {
      CK(hipMemcpy(Gb + (size_t)j * M * 128, Cout, (size_t)M * 128 * 4, hipMemcpyDeviceToDevice));
      for (int q = 0; q < S; ++q) CK(hipMemcpy(Ab + ((size_t)q * Mb + (size_t)j * M) * C, A + (size_t)q * M * C, (size_t)M * C * 4, hipMemcpyDeviceToDevice));
    }
    for (int N : {128, 64}) for (int pipe : {0, 1}) {
      pgt_tune("gemm_tn_pipe", pipe);
      char nm[80]; snprintf(nm, 80, "TN  dW[330,%d] M=12x  pipe=%d", N, pipe);
      timeit(nm, [&]() { pgt_gemm_tn_acc_f32(Ab, C, Mb * C, S, C, Gb, N, dW, N, bias, Mb, N, st); }, 2.0 * Mb * K * N);
    }
    pgt_tune("gemm_tn_pipe", 1);
    CK(hipFree(Ab)); CK(hipFree(Gb));
  }
  // timeline of one NN launch (N = 128)
  long long* tr; const size_t TRN = 8192 * 4;
  CK(hipMalloc(&tr, TRN * 8)); CK(hipMemset(tr, 0, TRN * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &tr, sizeof(tr)));
  CK(hipDeviceSynchronize());
  pgt_gemm_f32(A, C, (int64_t)M * C, S, C, W, 128, 1, Cout, 128, 0, 128, bias, M, 128, 0, st);
  CK(hipDeviceSynchronize());
  std::vector<long long> t(TRN);
  CK(hipMemcpy(t.data(), tr, TRN * 8, hipMemcpyDeviceToHost));
  long long t0 = -1, t1 = 0; int nb = 0; double dur = 0;
  for (size_t b = 0; b < 8192; ++b) if (t[b * 4]) { nb++; if (t0 < 0 || t[b * 4] < t0) t0 = t[b * 4]; t1 = std::max(t1, t[b * 4 + 1]); dur += (t[b * 4 + 1] - t[b * 4]) / 100.0; }
  printf("NN timeline: %d workgroups, span %.1f us, mean workgroup duration %.1f us\n", nb, (t1 - t0) / 100.0, dur / nb);
  int hs[24] = {0}, he[24] = {0};
  for (size_t b = 0; b < 8192; ++b) if (t[b * 4]) { hs[std::min(23, (int)((t[b * 4] - t0) / 100.0 / 12.0))]++; he[std::min(23, (int)((t[b * 4 + 1] - t0) / 100.0 / 12.0))]++; }
  printf("starts per 12us:"); for (int i = 0; i < 24; ++i) printf(" %d", hs[i]); printf("\nends   per 12us:"); for (int i = 0; i < 24; ++i) printf(" %d", he[i]); printf("\n");
  return 0;
}
