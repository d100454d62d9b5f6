// LAB HARNESS (not shipped): the sliced-ELL / LDS-window aggregation schedule for the north-star shape
// (N = 200 000, F = 64, in-degree 8), swept over tile heights, workgroup sizes and load/store policies next to plain
// copy / read / write streams of the same buffers.   Build: scripts/build_lab.sh
//
//   Y[i,:] = sum_j  scale[c_ij] * X[c_ij,:]          (MODE 0: P_o of DConv, val = 1/deg_out[source], dcrnn.py:70-73)
//   Y[i,:] = sum_j  val_ij * X[c_ij,:]               (MODE 1: any operator)
//
// Operator layout ("ELLW"): rows are cut into tiles of TR rows; every row of a tile has W slots (padding slots point
// at a zero row); a slot is a 16-bit offset of the source row inside the tile's window [r0 - H, r0 + TR + H), or
// 0xFFFF for a source outside it (fetched through the CSR the operator was built from).  Everything a workgroup
// needs is addressable from blockIdx alone: window rows, slots and Y rows are requested in ONE memory phase.
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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ long long* g_trace = nullptr;
#define MARK(slot) do { if (TRACE && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 8 + (slot)] = (long long)wall_clock64(); } while (0)

__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7;
  return x * q + (x < r ? x : r) + (b >> 3);
}

template <int NT>
__device__ __forceinline__ f4 ld4(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  else return *reinterpret_cast<const f4*>(p);
}
template <int NT>
__device__ __forceinline__ void st4(float* p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}

// THREADS threads, G = THREADS/16 row groups of 16 lanes (one 256-byte row each).  WRMAX: window rows the LDS holds.
template <int THREADS, int WRMAX, int TRMAX, int MODE, int NTL, int NTS, int TRACE>
__global__ __launch_bounds__(THREADS) void ellw_kernel(
    const uint16_t* __restrict__ slots, const float* __restrict__ vals, const float* __restrict__ scale,
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    int n, int TR, int H, const float* __restrict__ X, float* __restrict__ Y, int xcd) {
  constexpr int W = 8;
  constexpr int G = THREADS / 16;
  constexpr int XPT = (WRMAX + G - 1) / G;
  constexpr int RPG = (TRMAX + G - 1) / G;
  __shared__ f4 s_x[(WRMAX + 1) * 16];
  const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int tile = xcd ? xcd_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int r0 = tile * TR, w0 = r0 - H, WR = TR + 2 * H;
  const int nr = (n - r0 < TR) ? (n - r0) : TR;
  const float* Xl = X + l16 * 4;
  MARK(0);
  // ---- one memory phase: window rows, their scales, the slot vectors of the rows this group will produce
  f4 xw[XPT];
  float sc[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    int wr = rg + G * i;
    wr = wr < WR ? wr : WR - 1;
    int r = w0 + wr;
    r = r < 0 ? 0 : (r < n ? r : n - 1);
    xw[i] = ld4<NTL>(Xl + (unsigned)(r * 64));
    if constexpr (MODE == 0) sc[i] = scale[r];
  }
  u4 sv[RPG];
#pragma unroll
  for (int k = 0; k < RPG; ++k) {
    int r = rg + G * k;
    r = r < nr ? r : nr - 1;
    sv[k] = *reinterpret_cast<const u4*>(slots + (unsigned)((r0 + r) * W));
  }
  // ---- window -> LDS (pre-scaled in MODE 0: the product is rounded once, like norm * x_j in the reference)
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int wr = rg + G * i;
    if (wr < WR) {
      f4 v = xw[i];
      if constexpr (MODE == 0) v = v * sc[i];
      s_x[wr * 16 + l16] = v;
    }
  }
  if (tid < 16) s_x[WR * 16 + tid] = (f4){0.f, 0.f, 0.f, 0.f};
  MARK(1);
  __syncthreads();
  MARK(2);
  // ---- gather out of the window, slot order, one rounded add per slot
#pragma unroll
  for (int k = 0; k < RPG; ++k) {
    const int r = rg + G * k;
    if (r < nr) {
      const unsigned d[8] = {sv[k].x & 0xffffu, sv[k].x >> 16, sv[k].y & 0xffffu, sv[k].y >> 16,
                             sv[k].z & 0xffffu, sv[k].z >> 16, sv[k].w & 0xffffu, sv[k].w >> 16};
      f4 x[8];
      float vv[8];
      if constexpr (MODE == 1) {
        const f4 va = *reinterpret_cast<const f4*>(vals + (unsigned)((r0 + r) * W));
        const f4 vb = *reinterpret_cast<const f4*>(vals + (unsigned)((r0 + r) * W) + 4);
        vv[0] = va.x; vv[1] = va.y; vv[2] = va.z; vv[3] = va.w; vv[4] = vb.x; vv[5] = vb.y; vv[6] = vb.z; vv[7] = vb.w;
      }
      bool far = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        far |= d[j] == 0xffffu;
        x[j] = s_x[(d[j] == 0xffffu ? (unsigned)WR : d[j]) * 16 + l16];
      }
      if (far) {   // rare: a source row outside the window (wrap-around, long-range edge) comes through the CSR
        const int q0 = rowptr[r0 + r];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (d[j] == 0xffffu) {
            const int c = col[q0 + j];
            f4 xx = *reinterpret_cast<const f4*>(Xl + (unsigned)(c * 64));
            if constexpr (MODE == 0) {
              const float s = scale[c];
              xx.x = __fmul_rn(xx.x, s); xx.y = __fmul_rn(xx.y, s); xx.z = __fmul_rn(xx.z, s); xx.w = __fmul_rn(xx.w, s);
            } else {
              vv[j] = val[q0 + j];
            }
            x[j] = xx;
          }
      }
      f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 0) {
          acc.x = __fadd_rn(acc.x, x[j].x); acc.y = __fadd_rn(acc.y, x[j].y);
          acc.z = __fadd_rn(acc.z, x[j].z); acc.w = __fadd_rn(acc.w, x[j].w);
        } else {
          acc.x = fmaf(vv[j], x[j].x, acc.x); acc.y = fmaf(vv[j], x[j].y, acc.y);
          acc.z = fmaf(vv[j], x[j].z, acc.z); acc.w = fmaf(vv[j], x[j].w, acc.w);
        }
      }
      st4<NTS>(Y + (unsigned)((r0 + r) * 64 + l16 * 4), acc);
    }
  }
  MARK(3);
}


// persistent variant: one workgroup per CU walks `tpw` tiles; the next tile's window / scales / slots are requested
// right after the barrier that publishes the current window, so they are in flight while the current tile is gathered
// and stored.  Slot vectors of the current tile live in LDS (the registers hold the next tile's).
template <int THREADS, int WRMAX, int TRMAX, int NTS, int GSTEP>
__global__ __launch_bounds__(THREADS) void ellw_persist_kernel(
    const uint16_t* __restrict__ slots, const float* __restrict__ scale,
    const int* __restrict__ rowptr, const int* __restrict__ col,
    int n, int TR, int H, const float* __restrict__ X, float* __restrict__ Y, int n_tiles, int tpw) {
  constexpr int W = 8;
  constexpr int G = THREADS / 16;
  constexpr int XPT = (WRMAX + G - 1) / G;
  constexpr int RPG = (TRMAX + G - 1) / G;
  __shared__ f4 s_x[(WRMAX + 1) * 16];
  __shared__ u4 s_sv[TRMAX];
  const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int WR = TR + 2 * H;
  const float* Xl = X + l16 * 4;
  // XCD x = blockIdx % 8 owns a contiguous range of tiles; its workgroups take them round-robin
  const int nwx = (int)(gridDim.x >> 3), x = (int)(blockIdx.x & 7u), l = (int)(blockIdx.x >> 3);
  const int q = n_tiles >> 3, rem = n_tiles & 7;
  const int t_lo = x * q + (x < rem ? x : rem), t_hi = t_lo + q + (x < rem ? 1 : 0);
  f4 xw[XPT];
  float sc[XPT];
  u4 sv[RPG];
  auto fetch = [&](int tile) {
    const int r0 = tile * TR, w0 = r0 - H;
    const int nr = (n - r0 < TR) ? (n - r0) : TR;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int wr = rg + G * i;
      wr = wr < WR ? wr : WR - 1;
      int r = w0 + wr;
      r = r < 0 ? 0 : (r < n ? r : n - 1);
      xw[i] = *reinterpret_cast<const f4*>(Xl + (unsigned)(r * 64));
      sc[i] = scale[r];
    }
#pragma unroll
    for (int k = 0; k < RPG; ++k) {
      int r = rg + G * k;
      r = r < nr ? r : nr - 1;
      sv[k] = *reinterpret_cast<const u4*>(slots + (unsigned)((r0 + r) * W));
    }
  };
  int tile = t_lo + l;
  if (tile >= t_hi) return;
  fetch(tile);
  if (tid < 16) s_x[WR * 16 + tid] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < tpw; ++it) {
    const int r0 = tile * TR;
    const int nr = (n - r0 < TR) ? (n - r0) : TR;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int wr = rg + G * i;
      if (wr < WR) s_x[wr * 16 + l16] = xw[i] * sc[i];
    }
#pragma unroll
    for (int k = 0; k < RPG; ++k) {
      const int r = rg + G * k;
      if (r < TR) s_sv[r] = sv[k];
    }
    __syncthreads();
    const int next = tile + nwx;
    const bool more = (it + 1 < tpw) && next < t_hi;
    if (more) fetch(next);
#pragma unroll 1
    for (int k = 0; k < RPG; ++k) {
      const int r = rg + G * k;
      if (r < nr) {
        const u4 s4 = s_sv[r];
        const unsigned d[8] = {s4.x & 0xffffu, s4.x >> 16, s4.y & 0xffffu, s4.y >> 16,
                               s4.z & 0xffffu, s4.z >> 16, s4.w & 0xffffu, s4.w >> 16};
        f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j0 = 0; j0 < 8; j0 += GSTEP) {
          f4 xx[GSTEP];
          bool far = false;
#pragma unroll
          for (int j = 0; j < GSTEP; ++j) {
            far |= d[j0 + j] == 0xffffu;
            xx[j] = s_x[(d[j0 + j] == 0xffffu ? (unsigned)WR : d[j0 + j]) * 16 + l16];
          }
          if (far) {
            const int q0 = rowptr[r0 + r];
#pragma unroll
            for (int j = 0; j < GSTEP; ++j)
              if (d[j0 + j] == 0xffffu) {
                const int c = col[q0 + j0 + j];
                f4 v = *reinterpret_cast<const f4*>(Xl + (unsigned)(c * 64));
                const float s = scale[c];
                v.x = __fmul_rn(v.x, s); v.y = __fmul_rn(v.y, s); v.z = __fmul_rn(v.z, s); v.w = __fmul_rn(v.w, s);
                xx[j] = v;
              }
          }
#pragma unroll
          for (int j = 0; j < GSTEP; ++j) {
            acc.x = __fadd_rn(acc.x, xx[j].x); acc.y = __fadd_rn(acc.y, xx[j].y);
            acc.z = __fadd_rn(acc.z, xx[j].z); acc.w = __fadd_rn(acc.w, xx[j].w);
          }
        }
        st4<NTS>(Y + (unsigned)((r0 + r) * 64 + l16 * 4), acc);
      }
    }
    if (!more) break;
    __syncthreads();
    tile = next;
  }
}

__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy_nt_kernel(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
// one tile-sized chunk per workgroup, all loads in flight before the first store (the shape of the ELLW kernel without
// window, slots and gather): the ceiling of that launch shape
template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void copy_tile_kernel(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  f4 v[PER];
  const size_t base = (size_t)blockIdx.x * THREADS * PER + threadIdx.x;
#pragma unroll
  for (int i = 0; i < PER; ++i) { const size_t j = base + (size_t)i * THREADS; v[i] = a[j < n ? j : n - 1]; }
#pragma unroll
  for (int i = 0; i < PER; ++i) { const size_t j = base + (size_t)i * THREADS; if (j < n) __builtin_nontemporal_store(v[i], b + j); }
}

struct Graph { std::vector<int> rp, col; std::vector<float> val, scale; };

static Graph local_graph(int n, int deg, int window) {
  Graph g; g.rp.resize(n + 1); g.col.resize((size_t)n * deg); g.val.resize((size_t)n * deg); g.scale.resize(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  std::vector<float> degout(n, 0.f);
  std::vector<float> w((size_t)n * deg);
  for (int i = 0; i < n; ++i) {
    g.rp[i] = i * deg;
    int c[64];
    for (int k = 0; k < deg; ++k) {
      int v; bool dup;
      do {
        int off = (int)(rnd() % window) - window / 2; if (off >= 0) off += 1;
        v = ((i + off) % n + n) % n;
        dup = false;
        for (int j = 0; j < k; ++j) dup |= c[j] == v;
      } while (dup);
      c[k] = v;
    }
    std::sort(c, c + deg);
    for (int k = 0; k < deg; ++k) { g.col[(size_t)i * deg + k] = c[k]; w[(size_t)i * deg + k] = 0.5f + (rnd() % 1000) / 1000.f; degout[c[k]] += w[(size_t)i * deg + k]; }
  }
  g.rp[n] = n * deg;
  for (int i = 0; i < n; ++i) g.scale[i] = degout[i] > 0 ? 1.f / degout[i] : 0.f;
  for (size_t q = 0; q < g.col.size(); ++q) g.val[q] = g.scale[g.col[q]];
  return g;
}

int main(int argc, char** argv) {
  const int n = 200000, F = 64, PAIRS = 6, deg = 8;
  Graph g = local_graph(n, deg, 64);
  int *rp, *col; float *val, *scale;
  CK(hipMalloc(&rp, (n + 1) * 4)); CK(hipMalloc(&col, g.col.size() * 4)); CK(hipMalloc(&val, g.val.size() * 4)); CK(hipMalloc(&scale, n * 4));
  CK(hipMemcpy(rp, g.rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(col, g.col.data(), g.col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(val, g.val.data(), g.val.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(scale, g.scale.data(), n * 4, hipMemcpyHostToDevice));
  float *X[PAIRS], *Y[PAIRS];
  std::vector<float> hx((size_t)n * F);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  for (int p = 0; p < PAIRS; ++p) {
    CK(hipMalloc(&X[p], (size_t)n * F * 4)); CK(hipMalloc(&Y[p], (size_t)n * F * 4));
    CK(hipMemcpy(X[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  // host reference (MODE 0 rounding: product rounded, then sequential adds)
  std::vector<float> ref((size_t)n * F);
  for (int i = 0; i < n; ++i)
    for (int f = 0; f < F; ++f) {
      float a = 0.f;
      for (int q = g.rp[i]; q < g.rp[i + 1]; ++q) { volatile float p = g.scale[g.col[q]] * hx[(size_t)g.col[q] * F + f]; a = a + p; }
      ref[(size_t)i * F + f] = a;
    }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double alg = 4.0 * (n + 1) + 8.0 * g.col.size() + 8.0 * n * F;
  auto timeit = [&](const char* name, auto fn, double bytes) {
    for (int i = 0; i < 2 * PAIRS; ++i) fn(i % PAIRS);
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      const int reps = 10 * PAIRS;
      for (int i = 0; i < reps; ++i) fn(i % PAIRS);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double us = ms * 1e3 / reps;
      best = std::min(best, us); sum += us;
    }
    printf("%-58s %7.2f us (mean %6.2f)  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, best, sum / 3, bytes / best / 1e3, bytes / best / 1e3 / 8000);
    fflush(stdout);
    return best;
  };
  printf("N=%d F=%d deg=%d local(+-32), algorithmic %.1f MB\n", n, F, deg, alg / 1e6);
  for (int blocks : {1024, 2048, 4096, 8192}) {
    char nm[64]; snprintf(nm, 64, "copy grid-stride g=%d", blocks);
    timeit(nm, [&](int p) { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, (const f4*)X[p], (f4*)Y[p], (size_t)n * F / 4); }, 8.0 * n * F);
  }
  timeit("copy nt g=2048", [&](int p) { hipLaunchKernelGGL(copy_nt_kernel, dim3(2048), dim3(256), 0, st, (const f4*)X[p], (f4*)Y[p], (size_t)n * F / 4); }, 8.0 * n * F);
  {
    const size_t n4 = (size_t)n * F / 4;
    timeit("copy tile 256thr x 8 f4 (32 KB/wg)", [&](int p) { hipLaunchKernelGGL((copy_tile_kernel<256, 8>), dim3((n4 + 2047) / 2048), dim3(256), 0, st, (const f4*)X[p], (f4*)Y[p], n4); }, 8.0 * n * F);
    timeit("copy tile 512thr x 4 f4 (32 KB/wg)", [&](int p) { hipLaunchKernelGGL((copy_tile_kernel<512, 4>), dim3((n4 + 2047) / 2048), dim3(512), 0, st, (const f4*)X[p], (f4*)Y[p], n4); }, 8.0 * n * F);
    timeit("copy tile 512thr x 8 f4 (64 KB/wg)", [&](int p) { hipLaunchKernelGGL((copy_tile_kernel<512, 8>), dim3((n4 + 4095) / 4096), dim3(512), 0, st, (const f4*)X[p], (f4*)Y[p], n4); }, 8.0 * n * F);
    timeit("copy tile 256thr x 16 f4 (64 KB/wg)", [&](int p) { hipLaunchKernelGGL((copy_tile_kernel<256, 16>), dim3((n4 + 4095) / 4096), dim3(256), 0, st, (const f4*)X[p], (f4*)Y[p], n4); }, 8.0 * n * F);
  }

  // ---- ELLW operator for a given (TR, H): slots (+ vals), built on the host here (the library builds it on the device)
  uint16_t* d_slots = nullptr; float* d_vals = nullptr;
  size_t cap = 0;
  auto build = [&](int TR, int H) {
    const int nt = (n + TR - 1) / TR, WR = TR + 2 * H;
    std::vector<uint16_t> s((size_t)nt * TR * 8);
    std::vector<float> v((size_t)nt * TR * 8, 0.f);
    size_t nfar = 0;
    for (int t = 0; t < nt; ++t)
      for (int r = 0; r < TR; ++r) {
        const int row = t * TR + r, w0 = t * TR - H;
        for (int j = 0; j < 8; ++j) {
          uint16_t d = (uint16_t)WR;
          if (row < n && g.rp[row] + j < g.rp[row + 1]) {
            const int c = g.col[g.rp[row] + j];
            v[((size_t)row) * 8 + j] = g.val[g.rp[row] + j];
            if (c - w0 >= 0 && c - w0 < WR) d = (uint16_t)(c - w0); else { d = 0xffff; nfar++; }
          }
          s[((size_t)row) * 8 + j] = d;
        }
      }
    if (s.size() > cap) {
      if (d_slots) { CK(hipFree(d_slots)); CK(hipFree(d_vals)); }
      cap = s.size() + 4096;
      CK(hipMalloc(&d_slots, cap * 2)); CK(hipMalloc(&d_vals, cap * 4));
    }
    CK(hipMemcpy(d_slots, s.data(), s.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vals, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return nfar;
  };
  std::vector<float> hy((size_t)n * F);
  auto check = [&](float* y, int mode) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; size_t nbad = 0;
    for (size_t i = 0; i < hy.size(); ++i) { double d = fabs((double)hy[i] - ref[i]); worst = std::max(worst, d); nbad += (mode == 0 ? hy[i] != ref[i] : d > 1e-5); }
    return std::make_pair(worst, nbad);
  };

#define RUN(THREADS, WRMAX, TRMAX, MODE, NTL, NTS, TRv, xcd)                                                              \
  do {                                                                                                                    \
    const int TR_ = (TRv), H_ = 32;                                                                                       \
    if (TR_ <= TRMAX && TR_ + 2 * H_ <= WRMAX) {                                                                          \
      const size_t nfar = build(TR_, H_);                                                                                 \
      const int nt = (n + TR_ - 1) / TR_;                                                                                 \
      CK(hipMemsetAsync(Y[0], 0xff, (size_t)n * F * 4, st));                                                              \
      hipLaunchKernelGGL((ellw_kernel<THREADS, WRMAX, TRMAX, MODE, NTL, NTS, 0>), dim3(nt), dim3(THREADS), 0, st, d_slots, \
                         d_vals, scale, rp, col, val, n, TR_, H_, X[0], Y[0], xcd);                                       \
      auto ck = check(Y[0], MODE);                                                                                        \
      char nm[128];                                                                                                       \
      snprintf(nm, 128, "ellw thr=%d WRMAX=%d TR=%d mode=%d ntl=%d nts=%d xcd=%d tiles=%d far=%zu err=%.1e/%zu", THREADS,  \
               WRMAX, TR_, MODE, NTL, NTS, xcd, nt, nfar, ck.first, ck.second);                                           \
      timeit(nm, [&](int p) { hipLaunchKernelGGL((ellw_kernel<THREADS, WRMAX, TRMAX, MODE, NTL, NTS, 0>), dim3(nt),        \
                                                 dim3(THREADS), 0, st, d_slots, d_vals, scale, rp, col, val, n, TR_, H_,   \
                                                 X[p], Y[p], xcd); }, alg);                                               \
    }                                                                                                                     \
  } while (0)

  // third lab round: lab kernel vs PRODUCT kernel (C ABI) on the same box, same buffers
  pgt_ellw op; memset(&op, 0, sizeof(op));
  op.halo = 32;
  if (pgt_ellw_plan(n, 32, deg, 1, &op.tile_rows, &op.width, &op.config, &op.n_tiles, &op.far_rows)) { printf("plan: %s\n", pgt_last_error()); return 1; }
  const size_t total = (size_t)op.n_tiles * op.tile_rows * op.width;
  uint16_t* pslots; float *pvals, *pscale; int32_t* pinfo;
  CK(hipMalloc(&pslots, total * 2)); CK(hipMalloc(&pvals, total * 4)); CK(hipMalloc(&pscale, n * 4)); CK(hipMalloc(&pinfo, 16));
  if (pgt_ellw_build(rp, col, val, n, (int64_t)g.col.size(), &op, pslots, pvals, pscale, nullptr, nullptr, pinfo, st)) { printf("build: %s\n", pgt_last_error()); return 1; }
  int32_t hinfo[4]; CK(hipMemcpyAsync(hinfo, pinfo, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  printf("product plan: %lld tiles of %d rows x %d slots; far %d, scale mismatches %d\n", (long long)op.n_tiles, op.tile_rows, op.width, hinfo[0], hinfo[1]);
  op.slots = pslots;
  for (int round = 0; round < 2; ++round) {
    RUN(1024, 456, 392, 0, 0, 1, 392, 1);
    op.scale = pscale; op.vals = nullptr;
    CK(hipMemsetAsync(Y[0], 0xff, (size_t)n * F * 4, st));
    pgt_spmm_ellw_f32(&op, rp, col, val, n, X[0], F, Y[0], F, nullptr, 0, 1.f, 0.f, F, st);
    { auto ck = check(Y[0], 0); printf("product mode 0 check: err %.1e, %zu mismatching\n", ck.first, ck.second); }
    timeit("PRODUCT pgt_spmm_ellw_f32 mode 0 (scale table)", [&](int p) { pgt_spmm_ellw_f32(&op, rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
    RUN(1024, 456, 392, 1, 0, 1, 392, 1);
    op.scale = nullptr; op.vals = pvals;
    timeit("PRODUCT pgt_spmm_ellw_f32 mode 1 (per-slot vals)", [&](int p) { pgt_spmm_ellw_f32(&op, rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
    timeit("PRODUCT pgt_spmm_csr_f32 (CSR row tiles)", [&](int p) { pgt_spmm_csr_f32(rp, col, val, n, X[p], F, Y[p], F, nullptr, 0, 1.f, 0.f, F, st); }, alg);
  }

  return 0;
}
