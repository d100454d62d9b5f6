// LAB HARNESS (not shipped): the one-workgroup DCRNN sequence kernels (csrc/seq_small.hip) at the reference's own configuration —
// BatchedDCRNN(2, 2, K = 3), 207 nodes / 1 515 edges, T = 12 — launch times forward / backward for a few batch sizes and the phase
// timeline of workgroup 0 (wall_clock64 ticks, 100 MHz) per time step:
//   forward slots  0 staged | 1 hops (z|r stack) | 2 z|r product + save | 3 H*R + hops (candidate stack) | 4 candidate + blend + save
//   backward slots 0 gate adjoints + stack load | 1 candidate product adjoint | 2 adjoint hops | 3 d(HR) + stack load | 4 z|r product
//                  adjoint | 5 adjoint hops
//   ./lab/seq_small_lab [B = 64] [Fin = 2] [O = 2] [K = 3]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ long long* g_seq_trace = nullptr;
constexpr int SEQ_STEPS = 12, SEQ_SLOTS = 6;
#define PGT_SEQ_MARK(step, slot)                                                                                   \
  do {                                                                                                             \
    if (g_seq_trace != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && (step) < SEQ_STEPS)                        \
      g_seq_trace[(step) * SEQ_SLOTS + (slot)] = (long long)wall_clock64();                                        \
  } while (0)

#define LAB_HAS_SEQ
#include "lab_stubs.h"
#include "../pytorch_geometric_temporal_amd/csrc/pgt_core.hip"
#include "../pytorch_geometric_temporal_amd/csrc/seq_small.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, Fin = argc > 2 ? atoi(argv[2]) : 2, O = argc > 3 ? atoi(argv[3]) : 2;
  const int K = argc > 4 ? atoi(argv[4]) : 3, N = 207, E = 1515, T = 12, C = Fin + O, S = 2 * K - 1;
  std::vector<int32_t> rp(N + 1, 0), col;
  std::vector<float> val;
  uint32_t seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (int i = 0; i < N; ++i) {                       // a METR-LA-like operator: every row 4 .. 11 slots, random sources
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
  pgt_csr op{d_rp, d_col, d_val};
  auto dev_rand = [&](size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)(rnd() % 2000) / 1000.f - 1.f);
    float* d; CK(hipMalloc(&d, std::max<size_t>(n, 1) * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  };
  const size_t nx = (size_t)B * T * N * Fin, no = (size_t)B * T * N * O;
  float* X = dev_rand(nx, 1.f);
  float* Wzr = dev_rand((size_t)S * C * 2 * O, 0.3f); float* bzr = dev_rand(2 * O, 0.1f);
  float* Wh = dev_rand((size_t)S * C * O, 0.3f); float* bh = dev_rand(O, 0.1f);
  float* dOut = dev_rand(no, 1.f);
  float *out, *save, *dX, *dWpart;
  const size_t per_step = (size_t)pgt_dcrnn_seq_small_save_floats(N, Fin, O, K), nW = (size_t)S * C * 3 * O + 3 * O;
  CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&save, (size_t)B * T * per_step * 4)); CK(hipMalloc(&dX, nx * 4));
  CK(hipMalloc(&dWpart, (size_t)B * nW * 4)); CK(hipMemset(dWpart, 0, (size_t)B * nW * 4));
  if (!pgt_dcrnn_seq_small_fits(N, nnz, nnz, Fin, O, K)) { printf("does not fit\n"); return 1; }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto fwd = [&]() {
    int rc = pgt_dcrnn_seq_small_f32(&op, &op, nnz, nnz, N, X, (int64_t)T * N * Fin, (int64_t)N * Fin, nullptr, Wzr, bzr, Wh, bh, B, T, Fin,
                                     O, K, out, (int64_t)T * N * O, (int64_t)N * O, save, st);
    if (rc) { printf("forward failed: %s\n", pgt_last_error()); exit(1); }
  };
  auto bwd = [&]() {
    int rc = pgt_dcrnn_seq_small_bwd_f32(&op, &op, nnz, nnz, N, dOut, (int64_t)T * N * O, (int64_t)N * O, out, (int64_t)T * N * O,
                                         (int64_t)N * O, nullptr, save, Wzr, Wh, B, T, Fin, O, K, dX, (int64_t)T * N * Fin,
                                         (int64_t)N * Fin, nullptr, dWpart, st);
    if (rc) { printf("backward failed: %s\n", pgt_last_error()); exit(1); }
  };
  auto timeit = [&](const char* name, auto&& run) {
    for (int i = 0; i < 3; ++i) run();
    CK(hipEventRecord(e0, st));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %8.2f us per launch = %6.2f us per time step\n", name, ms * 1e3 / reps, ms * 1e3 / reps / T);
  };
  printf("B = %d, BatchedDCRNN(%d, %d, K = %d), N = %d, nnz = %d, T = %d\n", B, Fin, O, K, N, nnz, T);
  timeit("forward", fwd);
  timeit("backward", bwd);
  long long* d_tr;
  CK(hipMalloc(&d_tr, SEQ_STEPS * SEQ_SLOTS * 8));
  for (int dir = 0; dir < 2; ++dir) {
    CK(hipMemset(d_tr, 0, SEQ_STEPS * SEQ_SLOTS * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_seq_trace), &d_tr, sizeof(d_tr)));
    if (dir == 0) fwd(); else bwd();
    CK(hipStreamSynchronize(st));
    long long* none = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_seq_trace), &none, sizeof(none)));
    std::vector<long long> tr(SEQ_STEPS * SEQ_SLOTS);
    CK(hipMemcpy(tr.data(), d_tr, tr.size() * 8, hipMemcpyDeviceToHost));
    const int slots = dir == 0 ? 5 : 6;
    printf("%s, workgroup 0: us spent up to each mark, per step (first column = since the previous step's last mark)\n", dir == 0 ? "forward" : "backward");
    for (int t = 1; t < T && t < SEQ_STEPS; ++t) {
      printf("  step %2d:", t);
      long long prev = tr[(t - 1) * SEQ_SLOTS + slots - 1];
      for (int sl = 0; sl < slots; ++sl) { printf(" %6.2f", (tr[t * SEQ_SLOTS + sl] - prev) / 100.0); prev = tr[t * SEQ_SLOTS + sl]; }
      printf("\n");
    }
  }
  return 0;
}
