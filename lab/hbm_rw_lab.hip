// LAB HARNESS (not shipped): what HBM gives one MI355X for a given READ : WRITE mix, with the Infinity Cache (256 MiB) out of the
// picture — every launch touches a fresh window of a buffer many times its size, so nothing it reads was written recently and
// nothing it writes is overwritten while still cached.  Question behind it (round 5): the fused T-GCN cell forward kernel moves
// 59 MB in and 200 MB out per launch and three differently built kernels all take 75 - 87 us; is that the kernels or the mix?
//   ./lab/hbm_rw_lab [window_MB=256]
// modes: streams of 16-byte accesses per lane, `R` read streams and `W` write streams of equal size per launch (R : W bytes),
// 2048 workgroups x 256 threads, grid-stride.  Prints TB/s of (read + written) bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int W>
__global__ __launch_bounds__(256) void rw_kernel(const f4* __restrict__ src, f4* __restrict__ dst, int64_t n4, float* sink) {
  // stream s of the reads covers src[s * n4 ...], of the writes dst[s * n4 ...]
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int s = 0; s < R; ++s) acc += src[(int64_t)s * n4 + i];
#pragma unroll
    for (int s = 0; s < W; ++s) { f4 v = acc; v.x += (float)s; dst[(int64_t)s * n4 + i] = v; }
  }
  if (W == 0 && acc.x == 123.456f) *sink = acc.y;          // keep the reads alive
}

template <int R, int W>
static void run(const char* name, char* buf, size_t total, size_t window, float* sink) {
  // one launch: R + W streams of `per` bytes each inside a window; consecutive launches take consecutive windows
  const size_t per = window / (R + W) / 4096 * 4096;
  const int64_t n4 = (int64_t)(per / 16);
  const int n_win = (int)(total / window);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int w) {
    char* base = buf + (size_t)(w % n_win) * window;
    hipLaunchKernelGGL((rw_kernel<R, W>), dim3(2048), dim3(256), 0, 0, (const f4*)base, (f4*)(base + (size_t)R * per), n4, sink);
  };
  for (int w = 0; w < n_win; ++w) launch(w);
  CK(hipDeviceSynchronize());
  const int reps = 3 * n_win;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) launch(w);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, mb = (double)(R + W) * per / 1e6;
  printf("{\"mix\": \"%s\", \"read_MB\": %.1f, \"write_MB\": %.1f, \"us\": %.1f, \"TBs\": %.3f}\n", name, R * per / 1e6, W * per / 1e6, us,
         mb / us);
}

// Write-only, three store patterns over [rows][64] fp32 tiles of 32 rows (what an MFMA epilogue has to choose between):
//   0: whole rows, 16 bytes per lane (an instruction = 4 complete 256-byte rows) — what staging through LDS buys
//   1: the accumulator layout as it is, 4 bytes per lane (an instruction = two 128-byte row pieces, rows 4 apart; csrc/gemm_bx.hip)
//   2: lane = row, 16 bytes per lane (an instruction = 32 pieces of 32 bytes; the first row-per-lane T-GCN forward kernel)
template <int PAT>
__global__ __launch_bounds__(256) void store_pattern_kernel(float* __restrict__ dst, int64_t n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
    float* tile = dst + t * 32 * 64;
    if (PAT == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f4 v = {(float)j, 1.f, 2.f, (float)lane};
        *reinterpret_cast<f4*>(tile + (4 * j + (lane >> 4)) * 64 + (lane & 15) * 4) = v;
      }
    } else if (PAT == 1) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + half * 32 + lo] = (float)(r + lane);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f4 v = {(float)q, 1.f, 2.f, (float)lane};
        *reinterpret_cast<f4*>(tile + lo * 64 + 8 * q + 4 * hi) = v;
      }
    }
  }
}

template <int PAT>
static void run_pattern(const char* name, char* buf, size_t total, size_t window) {
  const int64_t n_tiles = (int64_t)(window / (32 * 64 * 4));
  const int n_win = (int)(total / window);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int w) {
    hipLaunchKernelGGL((store_pattern_kernel<PAT>), dim3(1024), dim3(256), 0, 0, (float*)(buf + (size_t)(w % n_win) * window), n_tiles);
  };
  for (int w = 0; w < n_win; ++w) launch(w);
  CK(hipDeviceSynchronize());
  const int reps = 3 * n_win;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) launch(w);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("{\"store_pattern\": \"%s\", \"write_MB\": %.1f, \"us\": %.1f, \"TBs\": %.3f}\n", name, window / 1e6, us, window / us / 1e6);
}

int main(int argc, char** argv) {
  const size_t window = (size_t)(argc > 1 ? atoll(argv[1]) : 256) << 20;
  const size_t total = (size_t)16 << 30;                      // 16 GiB: 64 windows of 256 MiB
  char* buf;
  float* sink;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, total));
  run<1, 0>("read only", buf, total, window, sink);
  run<0, 1>("write only", buf, total, window, sink);
  run<1, 1>("copy 1:1", buf, total, window, sink);
  run<3, 1>("3:1", buf, total, window, sink);
  run<1, 3>("1:3", buf, total, window, sink);
  run<1, 4>("1:4 (the forward cell: 59 MB in, 200 MB out)", buf, total, window, sink);
  run<4, 1>("4:1 (the adjoint cell: 259 MB in, 51 MB out)", buf, total, window, sink);
  run<2, 1>("2:1", buf, total, window, sink);
  run<1, 2>("1:2", buf, total, window, sink);
  run_pattern<0>("whole rows, 16 B per lane", buf, total, window);
  run_pattern<1>("accumulator layout, 4 B per lane (two 128-byte pieces per instruction)", buf, total, window);
  run_pattern<2>("lane = row, 16 B per lane (32 pieces of 32 bytes per instruction)", buf, total, window);
  return 0;
}
