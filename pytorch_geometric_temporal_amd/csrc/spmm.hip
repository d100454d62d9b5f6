// CSR aggregation kernels (the fused replacement of PyG's propagate: gather -> norm*x_j -> scatter-add).
//
//   Y[i,:] = alpha * sum_{q in row i} val[q] * X[col[q],:]  +  beta * T[i,:]
//
// Two launch shapes, both deterministic (per-row sequential accumulation in slot order, no atomics):
//
//  * spmm_tile_kernel<VEC,LPR>  — F <= 64*VEC floats per row.  A 256-thread workgroup owns a tile of
//    TR = 64 consecutive rows.  The tile's rowptr slice and its col/val slots are staged into LDS with
//    coalesced loads; then each group of LPR lanes walks one row, broadcasting (col,val) out of LDS and
//    issuing VEC-wide coalesced reads of the neighbour's feature row (LPR*VEC*4 bytes contiguous).
//    For F = 64: LPR = 16 lanes x float4 = one 256-byte row per group, 4 rows per wavefront.
//    Tiles are handed to XCDs in contiguous ranges (blockIdx -> XCD is round-robin on MI355X) so that
//    a locality-ordered graph keeps its neighbour rows in one XCD's 4 MiB L2.
//
//  * spmm_wide_kernel<VEC>      — F > 64*VEC (node-major batches: F = B*C).  One wavefront per
//    (row, 64*VEC-float chunk); row index is wave-uniform so (col,val) come through the scalar path and the
//    neighbour read is a fully coalesced 1 KiB (VEC = 4) burst.
//
// Algorithmic bytes per launch: 4(N+1) + 8*nnz + 4*N*F (read X once) + 4*N*F (write Y) [+ 4*N*F for T].
#include <string.h>

#include "pgt_common.h"

namespace {

constexpr int CAP_PER_ROW = 24;  // LDS-staged slots per tile = 24 * TR (12 KiB at TR = 64); larger tiles read global

// A/B knobs (pgt_tune); the defaults are the shipped configuration
int g_tile_xcd = 1;    // hand tiles to XCDs in contiguous ranges
int g_tile_rows = 64;  // rows per tile for the F = 64 fast path (32 | 64 | 128)
int g_unroll = 8;      // neighbour loads in flight per lane group (4 | 8)
int g_wide_xcd = 1;    // XCD-slab block mapping of the wide kernel
int g_band_blocks = 4;  // band kernel: resident workgroups per CU the chunking aims at
int g_band_xcd = 1;     // band kernel: contiguous chunk ranges per XCD

template <int VEC>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

// blockIdx -> tile so that each XCD (block b runs on XCD b % 8) owns a contiguous range of tiles.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7;
  return x * q + (x < r ? x : r) + (b >> 3);
}

template <int VEC, int LPR, int TR, int U>
__global__ __launch_bounds__(256) void spmm_tile_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int64_t ldx, float* Y, int64_t ldy, const float* T,
    int64_t ldt, float alpha, float beta, int F, int xcd_remap) {
  constexpr int CAP = CAP_PER_ROW * TR;
  __shared__ int s_rp[TR + 1];
  __shared__ int s_col[CAP];
  __shared__ float s_val[CAP];

  const int tid = threadIdx.x;
  const int tile = xcd_remap ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int r0 = tile * TR;
  const int nr = (n_rows - r0 < TR) ? (n_rows - r0) : TR;

  if (tid <= nr) s_rp[tid] = rowptr[r0 + tid];
  __syncthreads();
  const int e0 = s_rp[0];
  const int nnz = s_rp[nr] - e0;
  const bool staged = nnz <= CAP;
  if (staged) {
    for (int q = tid; q < nnz; q += 256) {
      s_col[q] = col[e0 + q];
      s_val[q] = val[e0 + q];
    }
  }
  __syncthreads();

  constexpr int GROUPS = 256 / LPR;
  const int g = tid / LPR;
  const int f = (tid % LPR) * VEC;
  if (f >= F) return;  // no barriers below

  for (int r = g; r < nr; r += GROUPS) {
    const int a = s_rp[r] - e0, b = s_rp[r + 1] - e0;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    int q = a;
    if (staged) {
      // U independent neighbour-row loads are issued before the first fma consumes one (memory-level parallelism);
      // the fma chain itself stays in slot order.
      for (; q + U <= b; q += U) {
        int c[U];
        float v[U];
        float x[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) { c[u] = s_col[q + u]; v[u] = s_val[q + u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) ldv<VEC>(X + (int64_t)c[u] * ldx + f, x[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v[u], x[u][i], acc[i]);
      }
      if (U > 4) {
        for (; q + 4 <= b; q += 4) {
          int c[4];
          float v[4];
          float x[4][VEC];
#pragma unroll
          for (int u = 0; u < 4; ++u) { c[u] = s_col[q + u]; v[u] = s_val[q + u]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) ldv<VEC>(X + (int64_t)c[u] * ldx + f, x[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v[u], x[u][i], acc[i]);
        }
      }
      for (; q < b; ++q) {
        const int c0 = s_col[q];
        const float v0 = s_val[q];
        float x0[VEC];
        ldv<VEC>(X + (int64_t)c0 * ldx + f, x0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v0, x0[i], acc[i]);
      }
    } else {
      for (; q < b; ++q) {
        const int c0 = col[e0 + q];
        const float v0 = val[e0 + q];
        float x0[VEC];
        ldv<VEC>(X + (int64_t)c0 * ldx + f, x0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v0, x0[i], acc[i]);
      }
    }
    float out[VEC];
    if (T != nullptr) {
      float t[VEC];
      ldv<VEC>(T + (int64_t)(r0 + r) * ldt + f, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] = alpha * acc[i] + beta * t[i];
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] = alpha * acc[i];
    }
    stv<VEC>(Y + (int64_t)(r0 + r) * ldy + f, out);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// spmm_band64_kernel<RING> — F = 64 floats per row on a locality-ordered (banded) operator.
//
// Why: the per-row gather of the tile kernel moves deg x 256 B through the CU's vector L1 for every 256 B it
// writes; on MI355X a CU sustains only ~64 cache lines in flight, so at in-degree 8 the launch is bound by that
// queue (measured: HBM traffic == algorithmic bytes, 3.2 TB/s) and not by HBM.  Here a workgroup owns a contiguous
// chunk of rows and slides a RING-row window of X through LDS (RING x 256 B, rows [s0 - H, s0 + 64 + H) resident
// while rows [s0, s0 + 64) are produced, H = (RING - 64) / 2).  Every X row goes through the vector memory path
// once per chunk (plus 2H halo rows per chunk, L2 hits) with fully coalesced 256-B reads, and the deg-fold gather
// is served by ds_read_b128 (one 256-B row per 16-lane group: conflict-free, 256 B/clk/CU).  Neighbours outside
// the resident window (the wrap-around rows, or a graph that is not banded) fall back to a global read, so the
// kernel is correct for any operator; the host only selects it when most slots are within the halo.
// (col, val) never touch LDS: each 16-lane group loads up to 16 slots of its row with one coalesced read and
// broadcasts them with ds_bpermute (__shfl, width 16).  Accumulation is sequential in slot order: deterministic.
template <int RING>
__global__ __launch_bounds__(256) void spmm_band64_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt,
    float alpha, float beta, int rows_per_chunk, int xcd_remap) {
  constexpr int S = 64, H = (RING - S) / 2;
  __shared__ float4 s_x[RING * 16];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, l16 = lane & 15;
  const int chunk = xcd_remap ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int c0 = chunk * rows_per_chunk;
  if (c0 >= n_rows) return;  // whole workgroup
  const int c1 = (c0 + rows_per_chunk < n_rows) ? c0 + rows_per_chunk : n_rows;

  const int rg = tid >> 4;  // 16 row-groups of 16 lanes stage 16 rows per pass
  // leading part of the first window: rows [c0 - H, c0 + H)
  {
    const int a = (c0 - H > 0) ? c0 - H : 0;
    const int b = (c0 + H < n_rows) ? c0 + H : n_rows;
    for (int r = a + rg; r < b; r += 16)
      s_x[(r & (RING - 1)) * 16 + l16] = *reinterpret_cast<const float4*>(X + (int64_t)r * ldx + l16 * 4);
  }
  for (int s0 = c0; s0 < c1; s0 += S) {
    {  // rows [s0 + H, s0 + S + H): four independent 256-B row reads per 16-lane group, then the LDS writes
      const int a = s0 + H;
      float4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = a + rg + 16 * i;
        const int rc = r < n_rows ? r : n_rows - 1;  // clamped: the load is unconditional, the LDS write is not
        t[i] = *reinterpret_cast<const float4*>(X + (int64_t)rc * ldx + l16 * 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = a + rg + 16 * i;
        if (r < n_rows) s_x[(r & (RING - 1)) * 16 + l16] = t[i];
      }
    }
    const int send = (s0 + S < c1) ? s0 + S : c1;  // rows [s0, send) this step
    const int w_lo = (s0 - H > 0) ? s0 - H : 0;
    const int w_hi = (s0 + S + H < n_rows) ? s0 + S + H : n_rows;
    // rowptr[s0 .. s0 + 64] for the whole step, two coalesced reads per wave
    const int i0 = (s0 + lane < n_rows) ? s0 + lane : n_rows;
    const int i1 = (s0 + lane + 1 < n_rows) ? s0 + lane + 1 : n_rows;
    const int rp0 = rowptr[i0], rp1 = rowptr[i1];
    __syncthreads();

    // 16 row-quads per step, dealt round-robin to the 4 waves; each 16-lane group of a wave owns one row of the quad
#pragma unroll 1
    for (int qi = wave; qi * 4 < send - s0; qi += 4) {
      const int rl = qi * 4 + g;  // row within the step (< 64)
      const int row = s0 + rl;
      const int a = __shfl(rp0, rl), b = __shfl(rp1, rl);
      const int n = (row < send) ? b - a : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q0 = 0; __ballot(q0 < n) != 0ull; q0 += 16) {
        const bool mine = q0 + l16 < n;
        const int mc = mine ? col[a + q0 + l16] : 0;
        const float mv = mine ? val[a + q0 + l16] : 0.f;
        // wave-uniform: does any live slot of this 16-slot chunk point outside the resident window?
        const bool any_far = __ballot(mine && (mc < w_lo || mc >= w_hi)) != 0ull;
        for (int u0 = 0; u0 < 16 && __ballot(q0 + u0 < n) != 0ull; u0 += 8) {
          int c[8];
          float v[8];
          float4 x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { c[u] = __shfl(mc, u0 + u, 16); v[u] = __shfl(mv, u0 + u, 16); }
          // every ring slot is mapped LDS, so the read is unconditional (a dead slot's value is discarded below)
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = s_x[(c[u] & (RING - 1)) * 16 + l16];
          if (any_far) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (q0 + u0 + u < n && (c[u] < w_lo || c[u] >= w_hi))
                x[u] = *reinterpret_cast<const float4*>(X + (int64_t)c[u] * ldx + l16 * 4);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool live = q0 + u0 + u < n;  // select, not multiply-by-zero: a dead slot must not inject NaN
            acc.x = live ? fmaf(v[u], x[u].x, acc.x) : acc.x;
            acc.y = live ? fmaf(v[u], x[u].y, acc.y) : acc.y;
            acc.z = live ? fmaf(v[u], x[u].z, acc.z) : acc.z;
            acc.w = live ? fmaf(v[u], x[u].w, acc.w) : acc.w;
          }
        }
      }
      if (row < send) {
        float4 o;
        if (T != nullptr) {
          const float4 t = *reinterpret_cast<const float4*>(T + (int64_t)row * ldt + l16 * 4);
          o = make_float4(alpha * acc.x + beta * t.x, alpha * acc.y + beta * t.y, alpha * acc.z + beta * t.z,
                          alpha * acc.w + beta * t.w);
        } else {
          o = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
        }
        *reinterpret_cast<float4*>(Y + (int64_t)row * ldy + l16 * 4) = o;
      }
    }
    __syncthreads();
  }
}

// slots of a CSR operator whose source lies within +-32 / +-96 rows of the destination (selects the band kernel)
__global__ __launch_bounds__(256) void csr_locality_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int n_rows,
                                                           int32_t* out) {
  const int row = (int)(blockIdx.x * 256 + threadIdx.x);
  int n32 = 0, n96 = 0;
  if (row < n_rows) {
    for (int q = rowptr[row]; q < rowptr[row + 1]; ++q) {
      const int d = col[q] - row;
      n32 += (d >= -32 && d <= 32) ? 1 : 0;
      n96 += (d >= -96 && d <= 96) ? 1 : 0;
    }
  }
  if (n32) atomicAdd(&out[0], n32);
  if (n96) atomicAdd(&out[1], n96);
}

template <int VEC, int U>
__global__ __launch_bounds__(256) void spmm_wide_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int64_t ldx, float* Y, int64_t ldy, const float* T,
    int64_t ldt, float alpha, float beta, int F, int nchunks, int nrowgroups, int xcd_map) {
  const int wave = PGT_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  int chunk, rowgroup;
  if (xcd_map) {
    // XCD x (= blockIdx % 8 on MI355X) owns the column chunks c = 8*j + x and sweeps ALL rows of one chunk before
    // moving to the next: the 64*VEC-float column slab of X it gathers from (n_rows KiB at VEC = 4) stays in that
    // XCD's private L2, so every neighbour re-read after the first is an L2 hit and HBM sees X exactly once.
    const int x = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3);
    chunk = (local / nrowgroups) * 8 + x;
    rowgroup = local % nrowgroups;
    if (chunk >= nchunks) return;
  } else {
    chunk = (int)(blockIdx.x % (unsigned)nchunks);
    rowgroup = (int)(blockIdx.x / (unsigned)nchunks);
  }
  const int row = rowgroup * 4 + wave;  // wave-uniform: rowptr / col / val below are scalar (SMEM) loads
  if (row >= n_rows) return;
  const int f = (chunk * 64 + lane) * VEC;
  const bool active = f < F;
  const int a = rowptr[row], b = rowptr[row + 1];
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const float* Xf = X + (active ? f : 0);
  int q = a;
  for (; q + U <= b; q += U) {
    int c[U];
    float v[U];
    float x[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) { c[u] = col[q + u]; v[u] = val[q + u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) ldv<VEC>(Xf + (int64_t)c[u] * ldx, x[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v[u], x[u][i], acc[i]);
  }
  for (; q < b; ++q) {
    const int c0 = col[q];
    const float v0 = val[q];
    float x0[VEC];
    ldv<VEC>(Xf + (int64_t)c0 * ldx, x0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(v0, x0[i], acc[i]);
  }
  if (!active) return;
  float out[VEC];
  if (T != nullptr) {
    float t[VEC];
    ldv<VEC>(T + (int64_t)row * ldt + f, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = alpha * acc[i] + beta * t[i];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = alpha * acc[i];
  }
  stv<VEC>(Y + (int64_t)row * ldy + f, out);
}

// ChebConvAttention hop-1: coefficient val[q] * S[b, row, col[q]] (dense [B,N,N] attention gathered at the
// edges instead of the reference's [B,E] temporary); rows node-major [N][B][C].
__global__ __launch_bounds__(256) void spmm_att_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ S, int n_rows, int B, int C, const float* __restrict__ X,
    float* __restrict__ Y) {
  // one thread per (row, b, c) element
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n_rows * B * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int b = (int)((idx / C) % B);
  const int row = (int)(idx / ((int64_t)B * C));
  const int a = rowptr[row], e = rowptr[row + 1];
  const float* Srow = S + ((int64_t)b * n_rows + row) * n_rows;
  float acc = 0.f;
  for (int q = a; q < e; ++q) {
    const int j = col[q];
    const float w = val[q] * Srow[j];
    acc = fmaf(w, X[((int64_t)j * B + b) * C + c], acc);
  }
  Y[idx] = acc;
}

template <int VEC>
int launch_spmm(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const float* X,
                int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt, float alpha, float beta,
                int64_t F, pgt_stream_t stream) {
  const int64_t Fv = F / VEC;
  const int n = (int)n_rows, Fi = (int)F;
  dim3 block(256);
  if (Fv <= 64) {
#define PGT_SPMM_CASE(L, TR_, U_)                                                                             \
  PGT_LAUNCH((spmm_tile_kernel<VEC, L, TR_, U_>), dim3((unsigned)pgt_cdiv(n_rows, TR_)), block, stream, rowptr, \
             col, val, n, X, ldx, Y, ldy, T, ldt, alpha, beta, Fi, g_tile_xcd)
    if (Fv <= 4) { PGT_SPMM_CASE(4, 64, 4); }
    else if (Fv <= 8) { PGT_SPMM_CASE(8, 64, 4); }
    else if (Fv <= 16) {
      if (VEC == 4) {  // the F = 64 fast path carries the A/B variants
        const int key = g_tile_rows * 10 + g_unroll;
        if (key == 324) { PGT_SPMM_CASE(16, 32, 4); }
        else if (key == 328) { PGT_SPMM_CASE(16, 32, 8); }
        else if (key == 644) { PGT_SPMM_CASE(16, 64, 4); }
        else if (key == 1284) { PGT_SPMM_CASE(16, 128, 4); }
        else if (key == 1288) { PGT_SPMM_CASE(16, 128, 8); }
        else { PGT_SPMM_CASE(16, 64, 8); }
      } else {
        PGT_SPMM_CASE(16, 64, 4);
      }
    }
    else if (Fv <= 32) { PGT_SPMM_CASE(32, 64, 4); }
    else { PGT_SPMM_CASE(64, 64, 4); }
#undef PGT_SPMM_CASE
  } else {
    const int nchunks = (int)pgt_cdiv(Fv, 64);
    const int nrowgroups = (int)pgt_cdiv(n_rows, 4);
    const int xcd_map = (nchunks >= 16 && g_wide_xcd) ? 1 : 0;
    const int64_t nblocks = (int64_t)nrowgroups * (xcd_map ? pgt_cdiv(nchunks, 8) * 8 : nchunks);
    PGT_REQUIRE(nblocks < (int64_t)1 << 31, "pgt_spmm_csr_f32: grid too large");
    dim3 grid((unsigned)nblocks);
    if (g_unroll == 4) {
      PGT_LAUNCH((spmm_wide_kernel<VEC, 4>), grid, block, stream, rowptr, col, val, n, X, ldx, Y, ldy, T, ldt, alpha,
                 beta, Fi, nchunks, nrowgroups, xcd_map);
    } else {
      PGT_LAUNCH((spmm_wide_kernel<VEC, 8>), grid, block, stream, rowptr, col, val, n, X, ldx, Y, ldy, T, ldt, alpha,
                 beta, Fi, nchunks, nrowgroups, xcd_map);
    }
  }
  return pgt_check_launch("pgt_spmm_csr_f32");
}

}  // namespace

int pgt_spmm_tune(const char* key, int value) {
  if (strcmp(key, "spmm_tile_xcd") == 0) { g_tile_xcd = value; return 1; }
  if (strcmp(key, "spmm_tile_rows") == 0) { g_tile_rows = value; return 1; }
  if (strcmp(key, "spmm_unroll") == 0) { g_unroll = value; return 1; }
  if (strcmp(key, "spmm_wide_xcd") == 0) { g_wide_xcd = value; return 1; }
  if (strcmp(key, "spmm_band_blocks") == 0) { g_band_blocks = value > 0 ? value : 1; return 1; }
  if (strcmp(key, "spmm_band_xcd") == 0) { g_band_xcd = value; return 1; }
  return 0;
}

extern "C" int pgt_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T,
                                int64_t ldt, float alpha, float beta, int64_t F, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && F >= 0, "pgt_spmm_csr_f32: negative size");
  if (n_rows == 0 || F == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && X && Y, "pgt_spmm_csr_f32: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) - 128 && F < ((int64_t)1 << 31), "pgt_spmm_csr_f32: size exceeds int32 indexing");
  PGT_REQUIRE(ldx >= F && ldy >= F && (T == nullptr || ldt >= F), "pgt_spmm_csr_f32: row stride smaller than F");
  PGT_REQUIRE(Y != X, "pgt_spmm_csr_f32: Y must not alias X");
  // widest vector width every row start is aligned for
  auto ok = [&](int v) {
    const size_t a = (size_t)v * 4;
    return F % v == 0 && ldx % v == 0 && ldy % v == 0 && pgt_aligned(X, a) && pgt_aligned(Y, a) &&
           (T == nullptr || (ldt % v == 0 && pgt_aligned(T, a)));
  };
  if (ok(4)) return launch_spmm<4>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  if (ok(2)) return launch_spmm<2>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  return launch_spmm<1>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
}

static int spmm_validate(const char* who, const int32_t* rowptr, int64_t n_rows, const float* X, int64_t ldx,
                         float* Y, int64_t ldy, const float* T, int64_t ldt, int64_t F) {
  (void)who;
  PGT_REQUIRE(rowptr && X && Y, "pgt_spmm_csr_f32: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) - 128 && F < ((int64_t)1 << 31), "pgt_spmm_csr_f32: size exceeds int32 indexing");
  PGT_REQUIRE(ldx >= F && ldy >= F && (T == nullptr || ldt >= F), "pgt_spmm_csr_f32: row stride smaller than F");
  PGT_REQUIRE(Y != X, "pgt_spmm_csr_f32: Y must not alias X");
  return PGT_OK;
}

extern "C" int pgt_spmm_csr_band_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                     const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T,
                                     int64_t ldt, float alpha, float beta, int64_t F, int32_t halo,
                                     pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && F >= 0, "pgt_spmm_csr_f32: negative size");
  if (n_rows == 0 || F == 0) return PGT_OK;
  if (int rc = spmm_validate("pgt_spmm_csr_band_f32", rowptr, n_rows, X, ldx, Y, ldy, T, ldt, F)) return rc;
  PgtVecPick vp;
  vp.width(F); vp.operand(X, ldx); vp.operand(Y, ldy); vp.operand(T, ldt);
  if (halo <= 0 || halo > 96 || F != 64 || vp.v != 4)
    return pgt_spmm_csr_f32(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  const int ring = halo <= 32 ? 128 : 256;
  // chunking: aim at g_band_blocks resident workgroups per CU (LDS admits 4 at RING = 128, 2 at RING = 256)
  const int per_cu = ring == 128 ? (g_band_blocks < 4 ? g_band_blocks : 4) : (g_band_blocks < 2 ? g_band_blocks : 2);
  int64_t nblk = 256 * (int64_t)per_cu;
  int64_t rpc = pgt_cdiv(pgt_cdiv(n_rows, nblk), 16) * 16;
  if (rpc < 64) rpc = 64;
  nblk = pgt_cdiv(n_rows, rpc);
  dim3 grid((unsigned)nblk), block(256);
  if (ring == 128) {
    PGT_LAUNCH((spmm_band64_kernel<128>), grid, block, stream, rowptr, col, val, (int)n_rows, X, ldx, Y, ldy, T, ldt,
               alpha, beta, (int)rpc, g_band_xcd);
  } else {
    PGT_LAUNCH((spmm_band64_kernel<256>), grid, block, stream, rowptr, col, val, (int)n_rows, X, ldx, Y, ldy, T, ldt,
               alpha, beta, (int)rpc, g_band_xcd);
  }
  return pgt_check_launch("pgt_spmm_csr_band_f32");
}

extern "C" int pgt_csr_locality(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t* out2,
                                pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0, "pgt_csr_locality: negative size");
  PGT_REQUIRE(out2 != nullptr, "pgt_csr_locality: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) - 256, "pgt_csr_locality: size exceeds int32 indexing");
  if (hipMemsetAsync(out2, 0, 2 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
    pgt_set_error("pgt_csr_locality: memset failed");
    return PGT_ERR_LAUNCH;
  }
  if (n_rows == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && col, "pgt_csr_locality: null pointer");
  PGT_LAUNCH(csr_locality_kernel, dim3((unsigned)pgt_cdiv(n_rows, 256)), dim3(256), stream, rowptr, col, (int)n_rows,
             out2);
  return pgt_check_launch("pgt_csr_locality");
}

extern "C" int pgt_spmm_csr_att_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                    const float* S, int64_t n_rows, int64_t B, int64_t C, const float* X,
                                    float* Y, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && B >= 0 && C >= 0, "pgt_spmm_csr_att_f32: negative size");
  if (n_rows == 0 || B == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && S && X && Y, "pgt_spmm_csr_att_f32: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) && B < ((int64_t)1 << 31) && C < ((int64_t)1 << 31),
              "pgt_spmm_csr_att_f32: size exceeds int32 indexing");
  const int64_t total = n_rows * B * C;
  PGT_REQUIRE(pgt_cdiv(total, 256) < ((int64_t)1 << 31), "pgt_spmm_csr_att_f32: grid too large");
  dim3 grid((unsigned)pgt_cdiv(total, 256)), block(256);
  PGT_LAUNCH(spmm_att_kernel, grid, block, stream, rowptr, col, val, S, (int)n_rows, (int)B, (int)C, X, Y);
  return pgt_check_launch("pgt_spmm_csr_att_f32");
}
