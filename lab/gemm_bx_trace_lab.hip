// LAB HARNESS (not shipped): per-wavefront timeline of the shipped K-split split-bf16 kernels (csrc/gemm_bx.hip) at the
// training step's shapes: 330 -> 128 with the z | r gate epilogue, 330 -> 64 with the candidate epilogue.
//   ./lab/gemm_bx_trace_lab [M = 211968]
// Marks (wall_clock64, 10 ns): consumer wavefront 0: loop top | k-loop done | barrier passed | before operand wait | operands
// landed | gate math done | stores issued;  producer wavefront 4: loop top | k-loop + conversions done | part_seen passed |
// partial sums written | barrier passed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ long long* g_bx_trace = nullptr;
constexpr int TR_FIRST = 6, TR_ITERS = 6, TR_SLOTS = 8;
#define BX_TRACE(slot)                                                                                                  \
  do {                                                                                                                  \
    if (g_bx_trace != nullptr && lane == 0 && (wave == 0 || wave == 4) && n_iter >= TR_FIRST && n_iter < TR_FIRST + TR_ITERS) \
      g_bx_trace[(((size_t)blockIdx.x * 2 + (wave >> 2)) * TR_ITERS + (n_iter - TR_FIRST)) * TR_SLOTS + (slot)] =       \
          (long long)wall_clock64();                                                                                    \
  } while (0)

// compile-time: a run-time switch in front of the hand-issued loads changes the kernel it is supposed to measure
// (scripts/build_lab.sh builds one binary per mask: lab/gemm_bx_trace_lab_<mask>)
#ifndef BX_SKIP
#define BX_SKIP 0
#endif
#define BX_LAB_SKIP(bit) ((BX_SKIP & (bit)) != 0)

#define LAB_HAS_GEMM
#define LAB_HAS_GEMM_BX
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/gemm.hip"
#include "../pytorch_geometric_temporal_amd/csrc/gemm_bx.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 211968;
  const int S = 5, C = 66, O = 64, Fin = 2;
  // argv[2] = 1: the side outputs (H * R into the candidate's stack slot; H' into the next step's) go to 256-byte aligned
  // [M, 64] rows instead of columns 2 .. 65 of 264-byte rows
  const bool aligned_side = argc > 2 && atoi(argv[2]) == 1;
  const int side_ld = aligned_side ? 64 : C, side_fin = aligned_side ? 0 : Fin;
  float *TS, *W, *bias, *zr, *H, *xhr, *ht, *out0, *out1;
  CK(hipMalloc(&TS, (size_t)S * M * C * 4)); CK(hipMalloc(&W, (size_t)S * C * 2 * O * 4)); CK(hipMalloc(&bias, 2 * O * 4));
  CK(hipMalloc(&zr, (size_t)M * 2 * O * 4)); CK(hipMalloc(&H, (size_t)M * O * 4)); CK(hipMalloc(&xhr, (size_t)M * C * 4));
  CK(hipMalloc(&ht, (size_t)M * O * 4)); CK(hipMalloc(&out0, (size_t)M * O * 4)); CK(hipMalloc(&out1, (size_t)M * C * 4));
  std::vector<float> h((size_t)S * M * C);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f;
  CK(hipMemcpy(TS, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)S * C * 2 * O * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(H, h.data(), (size_t)M * O * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 2 * O * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_zr = [&]() {
    int rc = pgt_gemm_gru_zr_f32(TS, C, M * C, S, C, W, 2 * O, 1, bias, zr, H, O, xhr, side_ld, side_fin, M, O, st);
    if (rc) { printf("zr launch failed: %s\n", pgt_last_error()); exit(1); }
  };
  auto run_h = [&]() {
    int rc = pgt_gemm_gru_h_f32(TS, C, M * C, S, C, W, O, 1, bias, ht, zr, H, O, out0, O, nullptr, out1 + side_fin, side_ld, M, O, st);
    if (rc) { printf("h launch failed: %s\n", pgt_last_error()); exit(1); }
  };
  auto timeit = [&](const char* name, auto&& fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-84s %8.2f us per launch\n", name, ms * 1e3 / 20);
  };
  {
    char nm[160];
    snprintf(nm, sizeof nm, "skip mask %2d%s | 330 -> 128 + z | r gates", BX_SKIP, aligned_side ? " aligned side rows" : ""); timeit(nm, run_zr);
    snprintf(nm, sizeof nm, "skip mask %2d%s | 330 -> 64 + candidate gate", BX_SKIP, aligned_side ? " aligned side rows" : ""); timeit(nm, run_h);
  }
  if (BX_SKIP != 0 || aligned_side) return 0;

  const int nwg = 256;
  long long* d_tr;
  const size_t ntr = (size_t)nwg * 2 * TR_ITERS * TR_SLOTS;
  CK(hipMalloc(&d_tr, ntr * 8));
  const char* names[2] = {"330 -> 128 + z | r gates", "330 -> 64 + candidate gate"};
  for (int which = 0; which < 2; ++which) {
    CK(hipMemset(d_tr, 0, ntr * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_trace), &d_tr, sizeof(d_tr)));
    if (which == 0) run_zr(); else run_h();
    CK(hipStreamSynchronize(st));
    long long* nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bx_trace), &nul, sizeof(nul)));
    std::vector<long long> tr(ntr);
    CK(hipMemcpy(tr.data(), d_tr, ntr * 8, hipMemcpyDeviceToHost));
    printf("\n== %s: timeline of iterations %d .. %d (us since the workgroup's first mark)\n", names[which], TR_FIRST, TR_FIRST + TR_ITERS - 1);
    for (int wg : {0, 3, 100, 255}) {
      long long t0 = 1LL << 62;
      for (int role = 0; role < 2; ++role)
        for (int it = 0; it < TR_ITERS; ++it)
          for (int s = 0; s < TR_SLOTS; ++s) {
            long long v = tr[(((size_t)wg * 2 + role) * TR_ITERS + it) * TR_SLOTS + s];
            if (v > 0) t0 = std::min(t0, v);
          }
      for (int role = 0; role < 2; ++role) {
        printf("wg %3d %s:", wg, role ? "producer" : "consumer");
        for (int it = 0; it < TR_ITERS; ++it) {
          printf("  |");
          for (int s = 0; s < (role ? 5 : 7); ++s) {
            long long v = tr[(((size_t)wg * 2 + role) * TR_ITERS + it) * TR_SLOTS + s];
            printf(" %6.2f", v > 0 ? (v - t0) * 0.01 : -1.0);
          }
        }
        printf("\n");
      }
    }
    // mean phase lengths over all workgroups
    double cph[7] = {0}, pph[7] = {0};
    int cn = 0, pn = 0;
    for (int wg = 0; wg < nwg; ++wg)
      for (int it = 0; it + 1 < TR_ITERS; ++it) {
        const long long* c = &tr[(((size_t)wg * 2 + 0) * TR_ITERS + it) * TR_SLOTS];
        const long long* p = &tr[(((size_t)wg * 2 + 1) * TR_ITERS + it) * TR_SLOTS];
        if (c[0] > 0 && c[6] > 0 && c[TR_SLOTS] > 0) {
          for (int s = 1; s < 7; ++s) cph[s] += ((c[s] > 0 ? c[s] : c[s - 1]) - (c[s - 1] > 0 ? c[s - 1] : c[s])) * 0.01;
          cph[0] += (c[TR_SLOTS] - c[0]) * 0.01;
          ++cn;
        }
        if (p[0] > 0 && p[4] > 0 && p[TR_SLOTS] > 0) {
          for (int s = 1; s < 5; ++s) pph[s] += (p[s] - p[s - 1]) * 0.01;
          pph[0] += (p[TR_SLOTS] - p[0]) * 0.01;
          ++pn;
        }
      }
    if (cn) printf("consumer mean (us): iteration %.2f = k-loop %.2f | barrier %.2f | partials + nan check %.2f | operand wait %.2f | gate math %.2f | stores + next operands %.2f\n",
                   cph[0] / cn, cph[1] / cn, cph[2] / cn, cph[3] / cn, cph[4] / cn, cph[5] / cn, cph[6] / cn);
    if (pn) printf("producer mean (us): iteration %.2f = k-loop + conversions %.2f | part_seen %.2f | partials written %.2f | barrier %.2f\n",
                   pph[0] / pn, pph[1] / pn, pph[2] / pn, pph[3] / pn, pph[4] / pn);
  }
  return 0;
}
