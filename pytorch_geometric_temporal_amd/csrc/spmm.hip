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

#include <type_traits>

#include "pgt_common.h"

namespace {

constexpr int CAP_PER_ROW = 24;  // LDS-staged slots per tile = 24 * TR (12 KiB at TR = 64); larger tiles read global

// A/B knobs (pgt_tune); the defaults are the shipped configuration
int g_tile_xcd = 1;    // hand tiles to XCDs in contiguous ranges
int g_tile_nt = 1;     // pgt_tune("spmm_tile_nt"): non-temporal stores of the aggregated rows: 1 = when Y exceeds the L2s
                       // (>= 32 MiB: 33.5 vs 34.4 us at N = 200 k, F = 64), 2 = always, 0 = never
int g_tile_rows = 32;  // rows per tile for the F = 64 fast path (32 | 64 | 128); 32: finer tail, measured best
int g_unroll = 8;      // neighbour loads in flight per lane group (4 | 8)
int g_wide_xcd = 1;    // XCD-slab block mapping of the wide kernel
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

// streaming (non-temporal) form of stv: the aggregated rows are not re-read by the launch that writes them, so they
// need not displace the X rows that neighbouring tiles still gather from L2
template <int VEC>
__device__ __forceinline__ void stv_stream(float* __restrict__ p, const float (&v)[VEC]) {
#if defined(PGT_EMU)
  stv<VEC>(p, v);
#else
  if constexpr (VEC == 4) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
  } else if constexpr (VEC == 2) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 t = {v[0], v[1]};
    __builtin_nontemporal_store(t, reinterpret_cast<v2*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
#endif
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
    int64_t ldt, float alpha, float beta, int F, int xcd_remap, int skip_len) {
  constexpr int CAP = CAP_PER_ROW * TR;
  __shared__ int s_rp[TR + 1];
  __shared__ int s_col[CAP];
  __shared__ float s_val[CAP];

  const int tid = threadIdx.x;
  const int tile = (xcd_remap & 1) ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const bool stream_y = (xcd_remap & 2) != 0;
  const int r0 = tile * TR;
  const int nr = (n_rows - r0 < TR) ? (n_rows - r0) : TR;

  PGT_TRACE_MARK(0);
  if (tid <= nr) s_rp[tid] = rowptr[r0 + tid];
  __syncthreads();
  PGT_TRACE_MARK(1);
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
  PGT_TRACE_MARK(2);

  constexpr int GROUPS = 256 / LPR;
  const int g = tid / LPR;
  const int f = (tid % LPR) * VEC;
  if (f >= F) return;  // no barriers below

  for (int r = g; r < nr; r += GROUPS) {
    const int a = s_rp[r] - e0, b = s_rp[r + 1] - e0;
    if (skip_len > 0 && b - a > skip_len) continue;   // a long row: spmm_long_rows_kernel produces it (pgt_spmm_csr_long_f32)
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
    if (stream_y) stv_stream<VEC>(Y + (int64_t)(r0 + r) * ldy + f, out);
    else stv<VEC>(Y + (int64_t)(r0 + r) * ldy + f, out);
  }
  PGT_TRACE_MARK(3);
}

// ---------------------------------------------------------------------------------------------------------------
// spmm_ellw64_kernel<MODE> — F = 64 on a locality-ordered operator in the ELLW layout (pgt_ellw, include/pgt_hip.h).
//
// Why: the CSR row tiles pay a dependent chain per tile — rowptr -> (col, val) -> neighbour rows -> store, two
// barriers — and re-read every neighbour row through the vector L1 (N = 200 000, in-degree 8: 32.4 - 34 us, 0.43 of
// 8 TB/s, although FETCH + WRITE equal the algorithmic bytes).  Here everything a workgroup needs is addressable from
// blockIdx alone, so window rows, slot block and (MODE 0) source scales are requested in ONE memory phase:
//   * rows are cut into tiles of TR rows; a tile's sources lie in the window [r0 - H, r0 + TR + H) of X, which is
//     loaded once with coalesced 16-byte reads and parked in LDS (1 + 2H/TR reads of X through L2, 1x from HBM);
//   * every row has W slots (W = longest row rounded up to 8; padding slots point at a zero row behind the window); a
//     slot is the 16-bit offset of its source row inside the window, or 0xFFFF for a source outside it (wrap-around,
//     long-range edge), which is fetched through the CSR the operator was built from — any operator is handled;
//   * MODE 0 (val[q] == scale[col[q]] for every slot: P_o of DConv, dcrnn.py:70-73): the per-slot coefficient stream
//     is dropped; a window row is multiplied by its scale once when it enters LDS and the gather is a chain of rounded
//     adds in slot order — the same roundings as the reference's `norm * x_j` followed by scatter-add;
//     MODE 1: per-slot coefficients, fmaf chain in slot order (bit-identical to the CSR kernels);
//   * one 1024-thread workgroup per CU holds up to 456 window rows (114 KB of LDS): the host picks TR so that the
//     tiles fill whole rounds of the 256 CUs (N = 200 000, H = 32: 511 tiles of 392 rows, halo re-reads 16 %).
// Measured (lab/ellw_lab, N = 200 000, in-degree 8, rotating buffers): 21.2 us = 0.685 of 8 TB/s (MODE 0), 22.6 us
// (MODE 1); a float4 copy of the same X -> Y on the same box: 18.8 us.  Streaming (non-temporal) stores of Y matter:
// 27.6 us without them.
// Two launch shapes (ELLW_CFG):
//   A: 1024 threads, 456 window rows (130 - 155 KB of LDS): one workgroup per CU, the smallest halo overhead; load phase
//      and gather phase of a CU do not overlap.  Best at in-degree <= 8 (W = 8).
//   B:  512 threads, 240 window rows (67 - 79 KB): two workgroups per CU, one gathers out of LDS while the other's
//      memory phase is in flight.  For wider rows (W >= 16), where the LDS gather is as long as the memory phase.
// FAR0 / FAR1: LDS rows for out-of-window sources in source-scale / per-slot mode (what the LDS budget leaves)
//   C: the launch shape of a RENUMBERED operator (pgt_ellw.order, tile_order.hip): 1024 threads, a 400-row window with NO
//      halo (position in a patch order says nothing about adjacency, so a halo would be 14 % wasted reads) and the LDS it
//      frees given to the table of outside rows (a grown patch of 392 mesh nodes names 113 of them on average, up to ~180).
struct EllwCfgA { static constexpr int THREADS = 1024, WRMAX = 456, SLOTS = 392 * 16, FAR0 = 128, FAR1 = 32, HMIN = 1; };
struct EllwCfgB { static constexpr int THREADS = 512, WRMAX = 240, SLOTS = 176 * 16, FAR0 = 48, FAR1 = 12, HMIN = 1; };
struct EllwCfgC { static constexpr int THREADS = 1024, WRMAX = 400, SLOTS = 400 * 8, FAR0 = 208, FAR1 = 160, HMIN = 0; };
constexpr int ELLW_WMAX = 32;
constexpr unsigned ELLW_ROW_LEFT_OUT = 0xfffeu;   // first slot of a row the layout leaves out (longer than its width)

int g_ellw = 1;        // pgt_tune("spmm_ellw"): 0 = pgt_spmm_ellw_f32 runs the CSR kernels instead (A/B)
int g_ellw_rows = 0;   // pgt_tune("spmm_ellw_rows"): test hook, caps the planned tile height (0 = no cap)
int g_ellw_cus = 0;    // pgt_tune("spmm_ellw_cus"): test hook, CU count the plan balances for (0 = the device's)
int g_ellw_cfg = 0;    // pgt_tune("spmm_ellw_cfg"): launch shape pgt_ellw_plan picks: 0 = by row width, 1 = A, 2 = B

// W8C: slot vectors (8 slots) per row when known at compile time (1, 2), 0 = runtime
// PERM: the layout lives in a renumbered row space — layout row p is row order[p] of X / Y / T (slots, scale, far_col,
// rowptr / col are all in layout numbering); every X row is still one coalesced 256-byte read and every Y row one
// 256-byte streaming store, just not at consecutive addresses
// HUB: the operator has a few rows of thousands of slots (hubs) that the layout leaves out (ELLW_ROW_LEFT_OUT).  Their slots
// are cut into pieces of at most 2 G = 128 (64) (source, coefficient) pairs, piece t rides with tile t: the pairs are addressable
// from blockIdx like the table of outside rows, their X rows are requested behind them in the shadow of the window loads, every
// lane group multiplies its one or two rows, the four groups of a wavefront add up through the lanes and each wavefront leaves ONE
// partial row in the workspace; ellw_hub_combine_kernel (a second, tiny launch) adds a hub's partial rows in a fixed order.  A hub
// on a workgroup of its own is a chain of ten dependent round trips (10.7 us per launch); here the chain hides in a memory phase
// that is waited for anyway.
struct EllwHub {
  const int32_t* col;    // [pieces * 2 G] source row, -1 = unused entry
  const float* val;      // [pieces * 2 G]
  float* partial;        // [pieces * (THREADS / 64) * 64]
  int pieces;            // tiles [0, pieces) carry one piece each
};

template <int MODE, class CFG, int W8C, bool PERM = false, bool HUB = false>
__global__ __launch_bounds__(CFG::THREADS) void spmm_ellw64_kernel(
    const uint16_t* __restrict__ slots, const float* __restrict__ vals, const float* __restrict__ scale,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val, int n_rows,
    int TR, int H, int W, const float* __restrict__ X, int ldx, float* Y, int ldy, const float* T, int ldt, float alpha,
    float beta, int flags, const int32_t* __restrict__ far_col, const int32_t* __restrict__ order, EllwHub hub,
    const int32_t* __restrict__ far_src) {
  constexpr int THREADS = CFG::THREADS, WRMAX = CFG::WRMAX, SLOTS = CFG::SLOTS;
  constexpr int FARMAX = MODE == 0 ? CFG::FAR0 : CFG::FAR1;   // LDS rows behind the zero row for out-of-window sources
  constexpr int G = THREADS / 16;                      // row groups of 16 lanes: one 256-byte row each
  constexpr int XPT = (WRMAX + G - 1) / G;             // window rows per group
  constexpr int RPG = (WRMAX - 2 * CFG::HMIN + G - 1) / G;   // output rows per group (TR <= WRMAX - 2 H, H >= HMIN)
  constexpr int NSV = (SLOTS / 8 + THREADS - 1) / THREADS;   // slot vectors per thread
  __shared__ pgt_f4 s_x[(WRMAX + 1 + FARMAX) * 16];
  __shared__ pgt_u4 s_slots[SLOTS / 8];
  __shared__ pgt_f4 s_vals[MODE == 1 ? SLOTS / 4 : 1];
  const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int tile = (flags & 1) ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const bool stream_y = (flags & 2) != 0;
  const int r0 = tile * TR, w0 = r0 - H, WR = TR + 2 * H;
  const int nr = (n_rows - r0 < TR) ? (n_rows - r0) : TR;
  const int W8 = W8C > 0 ? W8C : (W >> 3);             // slot vectors (8 x u16) per row
  // rows wider than 64 floats (node-major batches, F = B * C): blockIdx.y picks a 64-float column chunk; the window of
  // a chunk is the 256-byte segment of each window row, the slot block is shared by all chunks of the tile
  const int chunk0 = (int)blockIdx.y * 64;
  X += chunk0; Y += chunk0;
  if (T != nullptr) T += chunk0;
  const float* Xl = X + l16 * 4;
  // ---- one memory phase: window rows, their source scales, the tile's slot block (and coefficient block)
  // Out-of-window sources (wrap-around rows, long-range edges): the layout lists up to FARMAX of them per tile
  // (far_col, -1 = unused entry) and their slots already point at the LDS rows behind the zero row, so the gather below
  // is uniform.  The list is requested FIRST (loads return in order: its consumer, the far-row loads, then waits for
  // nothing younger) and the far rows ride in the shadow of the window loads.
  constexpr int FPT = (FARMAX + G - 1) / G;            // far rows per lane group
  int fc[FPT];
  int fx[PERM ? FPT : 1];                              // PERM: the outside rows' X rows (the caller's numbering), straight from the layout
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int k = rg + G * i;
    fc[i] = (far_col != nullptr && k < FARMAX) ? far_col[(size_t)tile * FARMAX + k] : -1;
    if constexpr (PERM) fx[i] = (far_src != nullptr && k < FARMAX) ? far_src[(size_t)tile * FARMAX + k] : -1;
  }
  constexpr int HU = 2;                                // hub pairs per lane group
  int hc[HUB ? HU : 1];
  float hv[HUB ? HU : 1];
  const bool hub_tile = HUB && tile < hub.pieces;
  if constexpr (HUB) {
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      hc[u] = -1; hv[u] = 0.f;
      if (hub_tile) {
        const size_t e = (size_t)tile * (G * HU) + rg + G * u;
        hc[u] = hub.col[e]; hv[u] = hub.val[e];
      }
    }
  }
  pgt_f4 xw[XPT];
  float sc[XPT];
  int xrow[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    int wr = rg + G * i;
    wr = wr < WR ? wr : WR - 1;                        // unconditional clamped loads: a static number in flight
    int r = w0 + wr;
    r = r < 0 ? 0 : (r < n_rows ? r : n_rows - 1);
    xrow[i] = r;
    if constexpr (PERM) xrow[i] = order[r];
    sc[i] = MODE == 0 ? scale[r] : 1.f;
  }
  int yrow[PERM ? RPG : 1];                            // PERM: where this lane group's output rows go
  if constexpr (PERM) {
    // (a renumbered layout has no halo: the window rows ARE the tile's rows, output row k of this lane group is its window row k)
    static_assert(!PERM || RPG <= XPT, "one order entry per window row serves the output row too");
#pragma unroll
    for (int k = 0; k < RPG; ++k) yrow[k] = xrow[k < XPT ? k : 0];
  }
#pragma unroll
  for (int i = 0; i < XPT; ++i) xw[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(xrow[i] * ldx));
  const int nvec = TR * W8;                            // <= SLOTS / 8
  pgt_u4 sv[NSV];
  pgt_f4 va[MODE == 1 ? 2 * NSV : 1];
  {
    const pgt_u4* sp = reinterpret_cast<const pgt_u4*>(slots + (size_t)tile * TR * W8 * 8);
#pragma unroll
    for (int i = 0; i < NSV; ++i) { const int v = tid + THREADS * i; sv[i] = sp[v < nvec ? v : nvec - 1]; }
    if constexpr (MODE == 1) {
      const pgt_f4* vp = reinterpret_cast<const pgt_f4*>(vals + (size_t)tile * TR * W8 * 8);
#pragma unroll
      for (int i = 0; i < 2 * NSV; ++i) { const int v = tid + THREADS * i; va[i] = vp[v < 2 * nvec ? v : 2 * nvec - 1]; }
    }
  }
  pgt_f4 xfar[FPT];
  float sfar[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {                      // far rows: second (short) hop behind far_col, in the shadow of the window
    xfar[i] = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    sfar[i] = 1.f;
    if (fc[i] >= 0 && fc[i] < n_rows) {
      int fr = fc[i];
      // (far_col -> order -> X row is a chain of three round trips where the window's own is two: with far_src it is two as well)
      if constexpr (PERM) fr = far_src != nullptr ? fx[i] : order[fr];
      xfar[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(fr * ldx));
      if constexpr (MODE == 0) sfar[i] = scale[fc[i]];
    }
  }
  pgt_f4 xh[HUB ? HU : 1];
  if constexpr (HUB) {
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      xh[u] = pgt_mk4(0.f, 0.f, 0.f, 0.f);
      if (hc[u] >= 0) xh[u] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(hc[u] * ldx));
    }
  }
  pgt_f4 tcur = pgt_mk4(0.f, 0.f, 0.f, 0.f);
  if (T != nullptr && rg < nr) tcur = *reinterpret_cast<const pgt_f4*>(T + (unsigned)((PERM ? yrow[0] : r0 + rg) * ldt) + l16 * 4);
  // ---- window -> LDS.  MODE 0: scaled on the way in (the product is rounded once, like norm * x_j in the reference)
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int wr = rg + G * i;
    if (wr < WR) {
      pgt_f4 v = xw[i];
      if constexpr (MODE == 0) v = pgt_mk4(pgt_mul_rn(v.x, sc[i]), pgt_mul_rn(v.y, sc[i]), pgt_mul_rn(v.z, sc[i]), pgt_mul_rn(v.w, sc[i]));
      s_x[wr * 16 + l16] = v;
    }
  }
  if (tid < 16) s_x[WR * 16 + tid] = pgt_mk4(0.f, 0.f, 0.f, 0.f);     // the row padding slots point at
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int k = rg + G * i;
    if (fc[i] >= 0) {
      pgt_f4 v = xfar[i];
      if constexpr (MODE == 0) v = pgt_mk4(pgt_mul_rn(v.x, sfar[i]), pgt_mul_rn(v.y, sfar[i]), pgt_mul_rn(v.z, sfar[i]), pgt_mul_rn(v.w, sfar[i]));
      s_x[(WR + 1 + k) * 16 + l16] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < NSV; ++i) { const int v = tid + THREADS * i; if (v < nvec) s_slots[v] = sv[i]; }
  if constexpr (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 2 * NSV; ++i) { const int v = tid + THREADS * i; if (v < 2 * nvec) s_vals[v] = va[i]; }
  }
  if constexpr (HUB) {
    if (hub_tile) {                                      // (uniform)
      pgt_f4 hacc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < HU; ++u)                       // select, not multiply-by-zero: an unused entry must not inject NaN
        if (hc[u] >= 0) hacc = pgt_mk4(fmaf(hv[u], xh[u].x, hacc.x), fmaf(hv[u], xh[u].y, hacc.y), fmaf(hv[u], xh[u].z, hacc.z), fmaf(hv[u], xh[u].w, hacc.w));
#pragma unroll
      for (int m = 16; m < 64; m <<= 1)                  // the four lane groups of the wavefront: (g0 + g1) + (g2 + g3)
        hacc = pgt_mk4(hacc.x + __shfl_xor(hacc.x, m), hacc.y + __shfl_xor(hacc.y, m), hacc.z + __shfl_xor(hacc.z, m), hacc.w + __shfl_xor(hacc.w, m));
      if ((tid & 63) < 16)
        *reinterpret_cast<pgt_f4*>(hub.partial + ((size_t)tile * (THREADS / 64) + (tid >> 6)) * 64 + l16 * 4) = hacc;
    }
  }
  __syncthreads();
  // ---- gather out of the window: rows rg, rg + G, ... ; sequential chain in slot order
  auto do_row = [&](const int r, pgt_f4& tc, const int yr, const int yr_next) {
    pgt_f4 tnext = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    if (T != nullptr && r + G < nr) tnext = *reinterpret_cast<const pgt_f4*>(T + (unsigned)(yr_next * ldt) + l16 * 4);
    if ((s_slots[r * W8].x & 0xffffu) == ELLW_ROW_LEFT_OUT) {   // a hub: not this launch's row (Y and an aliased T stay untouched)
      tc = tnext;
      return;
    }
    pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    auto chunk = [&](const int c8) {
      const pgt_u4 s4 = s_slots[r * W8 + c8];
      const unsigned d[8] = {s4.x & 0xffffu, s4.x >> 16, s4.y & 0xffffu, s4.y >> 16,
                             s4.z & 0xffffu, s4.z >> 16, s4.w & 0xffffu, s4.w >> 16};
      float vv[8];
      if constexpr (MODE == 1) {
        const pgt_f4 v0 = s_vals[(r * W8 + c8) * 2], v1 = s_vals[(r * W8 + c8) * 2 + 1];
        vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
      }
      pgt_f4 x[8];
      bool far = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        far |= d[j] == 0xffffu;
        x[j] = s_x[(d[j] == 0xffffu ? (unsigned)WR : d[j]) * 16 + l16];
      }
      if (far) {   // rare: a source row outside the window comes through the CSR (slot j of the row = CSR slot j)
        const int q0 = rowptr[r0 + r] + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (d[j] == 0xffffu) {
            const int cj = col[q0 + j];
            int xj = cj;
            if constexpr (PERM) xj = order[cj];
            pgt_f4 xx = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(xj * ldx));
            if constexpr (MODE == 0) {
              const float s = scale[cj];
              xx = pgt_mk4(pgt_mul_rn(xx.x, s), pgt_mul_rn(xx.y, s), pgt_mul_rn(xx.z, s), pgt_mul_rn(xx.w, s));
            }
            x[j] = xx;
          }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 0) {
          acc = pgt_mk4(pgt_add_rn(acc.x, x[j].x), pgt_add_rn(acc.y, x[j].y), pgt_add_rn(acc.z, x[j].z), pgt_add_rn(acc.w, x[j].w));
        } else {
          acc = pgt_mk4(fmaf(vv[j], x[j].x, acc.x), fmaf(vv[j], x[j].y, acc.y), fmaf(vv[j], x[j].z, acc.z), fmaf(vv[j], x[j].w, acc.w));
        }
      }
    };
    // padding sits at the end of a row: a slot vector whose FIRST slot is padding (the zero row) is all padding, and so
    // are the vectors behind it — rows shorter than the layout's width stop early (out-degree spread of P_i: longest
    // row 20 slots -> width 24, mean 8)
    if constexpr (W8C > 0) {
      chunk(0);
#pragma unroll
      for (int c8 = 1; c8 < W8C; ++c8)
        if ((s_slots[r * W8 + c8].x & 0xffffu) != (unsigned)WR) chunk(c8);
    } else {
      chunk(0);
      for (int c8 = 1; c8 < W8 && (s_slots[r * W8 + c8].x & 0xffffu) != (unsigned)WR; ++c8) chunk(c8);
    }
    float out[4];
    if (T != nullptr) {
      out[0] = alpha * acc.x + beta * tc.x; out[1] = alpha * acc.y + beta * tc.y;
      out[2] = alpha * acc.z + beta * tc.z; out[3] = alpha * acc.w + beta * tc.w;
    } else {
      out[0] = alpha * acc.x; out[1] = alpha * acc.y; out[2] = alpha * acc.z; out[3] = alpha * acc.w;
    }
    float* yp = Y + (unsigned)(yr * ldy) + l16 * 4;
    if (stream_y) stv_stream<4>(yp, out);
    else stv<4>(yp, out);
    tc = tnext;
  };
  if constexpr (PERM) {
#pragma unroll
    for (int k = 0; k < RPG; ++k) {
      const int r = rg + G * k;
      if (r < nr) do_row(r, tcur, yrow[k], yrow[k + 1 < RPG ? k + 1 : k]);
    }
  } else if constexpr (W8C == 1) {
    // narrow rows: the row loop is unrolled so the LDS reads of several rows are in flight together
#pragma unroll
    for (int k = 0; k < RPG; ++k) {
      const int r = rg + G * k;
      if (r < nr) do_row(r, tcur, r0 + r, r0 + r + G);
    }
  } else {
    for (int r = rg; r < nr; r += G) do_row(r, tcur, r0 + r, r0 + r + G);
  }
}

// Y[hub row] from the partial rows its pieces left (spmm_ellw64_kernel<..., HUB>): one workgroup per hub, 64 lane groups take
// the partial rows round-robin in rising order (eight loads in flight each: a loop of one load and one add per trip was 25
// dependent round trips = 8 us), the group sums meet in LDS in group order — a fixed order: deterministic.
__global__ __launch_bounds__(1024) void ellw_hub_combine_kernel(const int32_t* __restrict__ hub_rows, int rows_per_hub,
                                                                const float* __restrict__ partial, float* Y, int ldy,
                                                                const float* T, int ldt, float alpha, float beta) {
  __shared__ pgt_f4 s_sum[64 * 16];
  const int tid = threadIdx.x, l16 = tid & 15, g = tid >> 4;
  const int hub = (int)blockIdx.x;
  const int row = hub_rows[hub];
  const float* p = partial + (size_t)hub * rows_per_hub * 64 + l16 * 4;
  pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 8;
  for (int r0 = g; r0 < rows_per_hub; r0 += 64 * U) {
    pgt_f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 64 * u;
      v[u] = *reinterpret_cast<const pgt_f4*>(p + (size_t)(r < rows_per_hub ? r : r0) * 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r0 + 64 * u < rows_per_hub) acc = pgt_mk4(acc.x + v[u].x, acc.y + v[u].y, acc.z + v[u].z, acc.w + v[u].w);
  }
  s_sum[g * 16 + l16] = acc;
  __syncthreads();
  if (g == 0) {
    pgt_f4 tt = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    if (T != nullptr) tt = *reinterpret_cast<const pgt_f4*>(T + (unsigned)(row * ldt) + l16 * 4);
    pgt_f4 out = s_sum[l16];
#pragma unroll 8
    for (int gg = 1; gg < 64; ++gg) {
      const pgt_f4 v = s_sum[gg * 16 + l16];
      out = pgt_mk4(out.x + v.x, out.y + v.y, out.z + v.z, out.w + v.w);
    }
    float o[4] = {alpha * out.x, alpha * out.y, alpha * out.z, alpha * out.w};
    if (T != nullptr) {
      o[0] = alpha * out.x + beta * tt.x; o[1] = alpha * out.y + beta * tt.y; o[2] = alpha * out.z + beta * tt.z; o[3] = alpha * out.w + beta * tt.w;
    }
    stv<4>(Y + (unsigned)(row * ldy) + l16 * 4, o);
  }
}

// scale[col[q]] = val[q]: the candidate per-source coefficient table of MODE 0 (every writer of one entry stores the
// same bits when the operator is source-scaled; ellw_build_kernel verifies it)
__global__ __launch_bounds__(256) void ellw_scale_scatter_kernel(const int32_t* __restrict__ col,
                                                                 const float* __restrict__ val, int nnz, float* scale) {
  const int q = (int)(blockIdx.x * 256 + threadIdx.x);
  if (q < nnz) scale[col[q]] = val[q];
}

// one thread per (row, slot): slot block + coefficient block of the ELLW layout from the CSR operator.
// info[0] += slots outside their tile's window, info[1] += slots whose val differs from scale[col] (bitwise),
// info[2] += rows longer than W (their tail is NOT represented: the caller must not use the operator),
// info[3] += out-of-window slots beyond the tile's far_max LDS rows (kept as 0xFFFF)
__global__ __launch_bounds__(256) void ellw_build_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, int n_rows, int TR, int H, int W,
                                                         int n_tiles, const float* __restrict__ scale, uint16_t* slots,
                                                         float* vals, int32_t* info, int32_t* far_col, int32_t* far_cnt,
                                                         int far_max) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n_tiles * TR * W;
  if (idx >= total) return;
  const int row = (int)(idx / W), j = (int)(idx % W);
  const int tile = row / TR, w0 = tile * TR - H, WR = TR + 2 * H;
  unsigned d = (unsigned)WR;   // padding: the zero row
  float v = 0.f;
  if (row < n_rows) {
    const int a = rowptr[row], len = rowptr[row + 1] - a;
    if (len > W) {
      // a row the layout cannot hold is LEFT OUT: its first slot says so (ELLW_ROW_LEFT_OUT) and the window kernel neither
      // gathers nor stores it — the caller produces it some other way (a hub: pgt_spmm_csr_rows_f32) or rejects the layout
      if (j == 0) { atomicAdd(&info[2], 1); d = ELLW_ROW_LEFT_OUT; }
    } else if (j < len) {
      const int c = col[a + j];
      v = val[a + j];
      if (c >= w0 && c < w0 + WR && c >= 0 && c < n_rows) d = (unsigned)(c - w0);
      else {
        // out of the window: one of the tile's far_max LDS rows behind the zero row if one is left (the slot then points
        // at it and the kernel prefetches X[c] into it), otherwise 0xFFFF = fetched through the CSR inside the gather
        // (the table is a small hash set: every slot that names the same source shares ONE LDS row — a tile of a mesh
        // numbered along a space-filling curve names ~85 distinct outside rows through ~230 slots)
        d = 0xffffu;
        atomicAdd(&info[0], 1);
        if (far_cnt != nullptr) {
          int32_t* tab = far_col + (int64_t)tile * far_max;
          unsigned h = ((unsigned)c * 2654435761u) % (unsigned)far_max;
          int k = -1;
          for (int probe = 0; probe < far_max; ++probe) {
            const int old = atomicCAS(&tab[h], -1, c);
            if (old == -1) { atomicAdd(&far_cnt[tile], 1); k = (int)h; break; }
            if (old == c) { k = (int)h; break; }
            h = h + 1 == (unsigned)far_max ? 0u : h + 1;
          }
          if (k >= 0) d = (unsigned)(WR + 1 + k);
          else atomicAdd(&info[3], 1);
        }
      }
      if (scale != nullptr) {
        unsigned bv, bs;
        const float s = scale[c];
        memcpy(&bv, &v, 4); memcpy(&bs, &s, 4);
        if (bv != bs) atomicAdd(&info[1], 1);
      }
    }
  }
  slots[idx] = (uint16_t)d;
  if (vals != nullptr) vals[idx] = v;
}


// Rows far longer than their neighbours (hubs): in the tile kernel one lane group walks a row's slots one dependent
// chain after the other, so a 2 000-slot row costs 660 us and a 20 000-slot row 6.4 ms at N = 200 000 (the whole launch
// otherwise takes 20 - 30 us).  Here one 1024-thread workgroup owns ONE long row: its lane groups (lpr lanes x VEC floats
// = one feature row each) take the slots round-robin, eight neighbour rows in flight per group, and the partial sums
// meet in LDS where group 0 adds them in group order — deterministic, and a pairwise-style sum that is closer to the
// exact result than the sequential chain.
template <int VEC>
__global__ __launch_bounds__(1024) void spmm_long_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ long_rows, const float* __restrict__ X, int64_t ldx, float* Y, int64_t ldy,
    const float* T, int64_t ldt, float alpha, float beta, int F, int lpr) {
  __shared__ float s_part[1024 * VEC];
  const int tid = threadIdx.x;
  const int groups = 1024 / lpr, g = tid / lpr, l = tid - g * lpr;
  const int f = l * VEC;
  const bool active = f < F;
  const int row = long_rows[blockIdx.x];
  const int a = rowptr[row], b = rowptr[row + 1];
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const float* Xf = X + (active ? f : 0);
  constexpr int U = 8;
  for (int q0 = a + g; q0 < b; q0 += groups * U) {
    int c[U];
    float v[U];
    float x[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + groups * u;
      const bool live = q < b;
      c[u] = live ? col[q] : col[a];
      v[u] = live ? val[q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) ldv<VEC>(Xf + (int64_t)c[u] * ldx, x[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = q0 + groups * u < b;              // select, not multiply-by-zero: a dead slot must not inject NaN
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = live ? fmaf(v[u], x[u][i], acc[i]) : acc[i];
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) s_part[tid * VEC + i] = acc[i];
  __syncthreads();
  if (g == 0 && active) {
    float out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = 0.f;
    for (int gg = 0; gg < groups; ++gg)
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] += s_part[(gg * lpr + l) * VEC + i];
    if (T != nullptr) {
      float t[VEC];
      ldv<VEC>(T + (int64_t)row * ldt + f, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] = alpha * out[i] + beta * t[i];
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] = alpha * out[i];
    }
    stv<VEC>(Y + (int64_t)row * ldy + f, out);
  }
}

// slots of a CSR operator whose source lies within +-32 / +-96 rows of the destination, the longest row (selects the ELLW
// layout) and the list of rows longer than long_len (handled by spmm_long_rows_kernel)
__global__ __launch_bounds__(256) void csr_locality_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int n_rows,
                                                           int32_t* out, int32_t* long_rows, int long_cap, int long_len) {
  const int row = (int)(blockIdx.x * 256 + threadIdx.x);
  int n32 = 0, n96 = 0;
  if (row < n_rows) {
    const int len = rowptr[row + 1] - rowptr[row];
    atomicMax(&out[2], len);
    if (long_rows != nullptr && len > long_len) {
      const int slot = atomicAdd(&out[3], 1);            // out[3]: number of long rows (may exceed long_cap: list truncated)
      if (slot < long_cap) long_rows[slot] = row;
    }
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
// edges instead of the reference's [B,E] temporary); rows node-major [N][B][C].  transpose_s: the operator is the
// transposed CSR (feature gradient), so the attention entry of slot (row <- col) is S[b, col, row].
__global__ __launch_bounds__(256) void spmm_att_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ S, int n_rows, int B, int C, const float* __restrict__ X,
    float* __restrict__ Y, int transpose_s) {
  // one thread per (row, b, c) element
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n_rows * B * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int b = (int)((idx / C) % B);
  const int row = (int)(idx / ((int64_t)B * C));
  const int a = rowptr[row], e = rowptr[row + 1];
  const float* Sb = S + (int64_t)b * n_rows * n_rows;
  float acc = 0.f;
  for (int q = a; q < e; ++q) {
    const int j = col[q];
    const float sv = transpose_s ? Sb[(int64_t)j * n_rows + row] : Sb[(int64_t)row * n_rows + j];
    const float w = val[q] * sv;
    acc = fmaf(w, X[((int64_t)j * B + b) * C + c], acc);
  }
  Y[idx] = acc;
}

// Gradient of the attention-weighted aggregation w.r.t. the attention (sampled dense-dense product):
//   dS[b, row, col[q]] += val[q] * < G[row, b, :], X[col[q], b, :] >     for every slot q and batch entry b.
// One 16-lane group per (slot, b): lanes stride the C channels, shuffle-reduce, one atomic per (slot, b)
// (duplicate (row, col) slots — the Laplacian diagonal and the extra "-1" diagonal of variant 1 — land on one entry).
__global__ __launch_bounds__(256) void sddmm_att_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val, int n_rows,
    int B, int C, const float* __restrict__ G, const float* __restrict__ X, float* dS) {
  const int row = (int)blockIdx.x;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int a = rowptr[row], e = rowptr[row + 1];
  const int64_t work = (int64_t)(e - a) * B;
  const int64_t iters = (work + 15) / 16;           // block-uniform trip count: the shuffles need every lane
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t w = it * 16 + grp;
    const bool live = w < work;
    const int q = a + (int)(live ? w / B : 0);
    const int b = (int)(live ? w % B : 0);
    const int j = col[q < e ? q : a];
    float acc = 0.f;
    if (live) {
      const float* gp = G + ((int64_t)row * B + b) * C;
      const float* xp = X + ((int64_t)j * B + b) * C;
      for (int c = l16; c < C; c += 16) acc = fmaf(gp[c], xp[c], acc);
    }
    acc += __shfl_xor(acc, 8, 16);
    acc += __shfl_xor(acc, 4, 16);
    acc += __shfl_xor(acc, 2, 16);
    acc += __shfl_xor(acc, 1, 16);
    if (live && l16 == 0) atomicAdd(dS + ((int64_t)b * n_rows + row) * n_rows + j, val[q] * acc);
  }
}

template <int VEC>
int launch_spmm(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const float* X,
                int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt, float alpha, float beta,
                int64_t F, pgt_stream_t stream, int skip_len = 0) {
  const int64_t Fv = F / VEC;
  const int n = (int)n_rows, Fi = (int)F;
  dim3 block(256);
  if (Fv <= 64) {
#define PGT_SPMM_CASE(L, TR_, U_)                                                                             \
  PGT_LAUNCH((spmm_tile_kernel<VEC, L, TR_, U_>), dim3((unsigned)pgt_cdiv(n_rows, TR_)), block, stream, rowptr, \
             col, val, n, X, ldx, Y, ldy, T, ldt, alpha, beta, Fi, (g_tile_xcd ? 1 : 0) | ((g_tile_nt == 2 || (g_tile_nt == 1 && (int64_t)n * Fi * 4 >= ((int64_t)32 << 20))) ? 2 : 0), skip_len)
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
  if (strcmp(key, "spmm_tile_nt") == 0) { g_tile_nt = value; return 1; }
  if (strcmp(key, "spmm_tile_rows") == 0) { g_tile_rows = value; return 1; }
  if (strcmp(key, "spmm_unroll") == 0) { g_unroll = value; return 1; }
  if (strcmp(key, "spmm_wide_xcd") == 0) { g_wide_xcd = value; return 1; }
  if (strcmp(key, "spmm_ellw") == 0) { g_ellw = value; return 1; }
  if (strcmp(key, "spmm_ellw_rows") == 0) { g_ellw_rows = value > 0 ? value : 0; return 1; }
  if (strcmp(key, "spmm_ellw_cus") == 0) { g_ellw_cus = value > 0 ? value : 0; return 1; }
  if (strcmp(key, "spmm_ellw_cfg") == 0) { g_ellw_cfg = value; return 1; }
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

static int ellw_device_cus() {
  if (g_ellw_cus > 0) return g_ellw_cus;
#ifdef PGT_EMU
  return 4;
#else
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else cus = 256;
  }
  return cus;
#endif
}

static int ellw_far_rows(int config, bool source_scaled) {
  if (config == 3) return source_scaled ? EllwCfgC::FAR0 : EllwCfgC::FAR1;
  return config == 1 ? (source_scaled ? EllwCfgA::FAR0 : EllwCfgA::FAR1) : (source_scaled ? EllwCfgB::FAR0 : EllwCfgB::FAR1);
}
static int ellw_wrmax(int config) { return config == 1 ? EllwCfgA::WRMAX : (config == 2 ? EllwCfgB::WRMAX : EllwCfgC::WRMAX); }
static int ellw_slots_cap(int config) { return config == 1 ? EllwCfgA::SLOTS : (config == 2 ? EllwCfgB::SLOTS : EllwCfgC::SLOTS); }

extern "C" int pgt_ellw_plan(int64_t n_rows, int32_t halo, int32_t max_row_len, int32_t source_scaled, int32_t* tile_rows,
                             int32_t* width, int32_t* config, int64_t* n_tiles, int32_t* far_rows) {
  PGT_REQUIRE(tile_rows && width && config && n_tiles && far_rows, "pgt_ellw_plan: null pointer");
  *tile_rows = 0; *width = 0; *config = 0; *n_tiles = 0; *far_rows = 0;
  PGT_REQUIRE(n_rows >= 1 && n_rows < ((int64_t)1 << 31) - 1024, "pgt_ellw_plan: n_rows out of range");
  PGT_REQUIRE(halo >= 0 && 2 * halo <= EllwCfgB::WRMAX - 8, "pgt_ellw_plan: halo %d outside [0, %d]", (int)halo,
              (EllwCfgB::WRMAX - 8) / 2);
  PGT_REQUIRE(max_row_len >= 0 && max_row_len <= ELLW_WMAX, "pgt_ellw_plan: rows of up to %d slots exceed the layout's %d",
              (int)max_row_len, ELLW_WMAX);
  const int W = max_row_len <= 8 ? 8 : (int)pgt_cdiv(max_row_len, 8) * 8;
  // halo 0 = the layout of a renumbered operator (pgt_tile_order_host): launch shape C
  // shape A for rows of up to 8 slots, and up to 16 in source-scale mode (no coefficient block: N = 200 000, in-degree 10 - 16, clocks
  // settled: 26.3 - 26.6 us against 27.2 - 27.4 on shape B; per-slot mode at W = 24: B 31.8 - 32.5 against A 32.9 - 33.7)
  const int cfg = halo == 0 ? 3 : (g_ellw_cfg == 1 || g_ellw_cfg == 2 ? g_ellw_cfg
                                   : ((W <= 8 || halo > 40 || (W <= 16 && source_scaled != 0)) ? 1 : 2));
  *config = cfg;
  *far_rows = ellw_far_rows(cfg, source_scaled != 0);
  const int wrmax = ellw_wrmax(cfg), slots_cap = ellw_slots_cap(cfg);
  const int per_cu = cfg == 2 ? 2 : 1;
  int cap = wrmax - 2 * halo;
  if (cap > slots_cap / W) cap = slots_cap / W;
  if (g_ellw_rows > 0 && cap > g_ellw_rows) cap = g_ellw_rows;
  cap &= ~3;
  PGT_REQUIRE(cap >= 4, "pgt_ellw_plan: no room for a tile");
  // whole rounds of the resident workgroups (one or two per CU): the smallest number of rounds whose tiles fit, then
  // the tile height that spreads the rows evenly over rounds * slots tiles (a last, nearly empty round would cost a
  // full tile time)
  const int64_t cus = (int64_t)ellw_device_cus() * per_cu;
  const int64_t rounds = pgt_cdiv(n_rows, cus * cap);
  int64_t tr = pgt_cdiv(pgt_cdiv(n_rows, cus * rounds), 4) * 4;
  if (tr > cap) tr = cap;
  if (tr < 4) tr = 4;
  *tile_rows = (int32_t)tr;
  *width = W;
  *n_tiles = pgt_cdiv(n_rows, tr);
  return PGT_OK;
}

static int ellw_check(const char* who, const pgt_ellw* op, int64_t n_rows) {
  PGT_REQUIRE(op != nullptr && op->slots != nullptr, "%s: null operator", who);
  PGT_REQUIRE(op->width >= 8 && op->width % 8 == 0 && op->width <= ELLW_WMAX, "%s: width %d", who, (int)op->width);
  PGT_REQUIRE(op->config >= 1 && op->config <= 3, "%s: config %d (1 = one workgroup per CU, 2 = two, 3 = renumbered)", who,
              (int)op->config);
  const int wrmax = ellw_wrmax(op->config);
  const int slots_cap = ellw_slots_cap(op->config);
  PGT_REQUIRE(op->halo >= (op->config == 3 ? 0 : 1) && op->tile_rows >= 1 && op->tile_rows + 2 * op->halo <= wrmax &&
                  (int64_t)op->tile_rows * op->width <= slots_cap,
              "%s: tile of %d rows x %d slots with halo %d does not fit the kernel (see pgt_ellw_plan)", who,
              (int)op->tile_rows, (int)op->width, (int)op->halo);
  PGT_REQUIRE(op->n_tiles == pgt_cdiv(n_rows, op->tile_rows), "%s: n_tiles %lld does not cover %lld rows", who,
              (long long)op->n_tiles, (long long)n_rows);
  PGT_REQUIRE(op->n_tiles < ((int64_t)1 << 31) && op->n_tiles * op->tile_rows * op->width < ((int64_t)1 << 40),
              "%s: operator too large", who);
  PGT_REQUIRE(op->hub_col == nullptr ||
                  (op->hub_val != nullptr && op->hub_rows != nullptr && op->hub_partial != nullptr && op->n_hub >= 1 &&
                   op->hub_split >= 1 && (int64_t)op->n_hub * op->hub_split <= op->n_tiles),
              "%s: hub tables need hub_val, hub_rows, hub_partial, n_hub >= 1 and n_hub * hub_split <= n_tiles", who);
  PGT_REQUIRE(op->far_src == nullptr || (op->far_col != nullptr && op->order != nullptr),
              "%s: far_src restates far_col through order: both must be set", who);
  PGT_REQUIRE(op->far_col == nullptr || op->far_rows == ellw_far_rows(op->config, op->scale != nullptr),
              "%s: far_rows %d does not match the kernel's table for this config / mode (see pgt_ellw_plan)", who,
              (int)op->far_rows);
  return PGT_OK;
}

extern "C" int pgt_ellw_build(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int64_t nnz,
                              const pgt_ellw* op, uint16_t* slots, float* vals, float* scale, int32_t* far_col,
                              int32_t* far_cnt, int32_t* info, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 1 && nnz >= 0, "pgt_ellw_build: bad size");
  PGT_REQUIRE(rowptr && col && val && slots && info, "pgt_ellw_build: null pointer");
  PGT_REQUIRE((far_col == nullptr) == (far_cnt == nullptr), "pgt_ellw_build: far_col and far_cnt go together");
  pgt_ellw tmp = *op;
  tmp.slots = slots;
  tmp.scale = scale;
  tmp.far_col = far_col;
  tmp.order = nullptr;
  if (int rc = ellw_check("pgt_ellw_build", &tmp, n_rows)) return rc;
  if (far_col != nullptr) {
    // -1 = unused entry of the far table; far_cnt counts a tile's DISTINCT out-of-window sources that got an LDS row
    if (hipMemsetAsync(far_col, 0xff, (size_t)op->n_tiles * op->far_rows * sizeof(int32_t), (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(far_cnt, 0, (size_t)op->n_tiles * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
      pgt_set_error("pgt_ellw_build: memset failed");
      return PGT_ERR_LAUNCH;
    }
  }
  PGT_REQUIRE(nnz < ((int64_t)1 << 31), "pgt_ellw_build: nnz exceeds int32 indexing");
  if (hipMemsetAsync(info, 0, 4 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
    pgt_set_error("pgt_ellw_build: memset failed");
    return PGT_ERR_LAUNCH;
  }
  if (scale != nullptr) {
    if (hipMemsetAsync(scale, 0, (size_t)n_rows * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      pgt_set_error("pgt_ellw_build: memset failed");
      return PGT_ERR_LAUNCH;
    }
    if (nnz > 0)
      PGT_LAUNCH(ellw_scale_scatter_kernel, dim3((unsigned)pgt_cdiv(nnz, 256)), dim3(256), stream, col, val, (int)nnz, scale);
  }
  const int64_t total = op->n_tiles * op->tile_rows * op->width;
  PGT_REQUIRE(pgt_cdiv(total, 256) < ((int64_t)1 << 31), "pgt_ellw_build: grid too large");
  PGT_LAUNCH(ellw_build_kernel, dim3((unsigned)pgt_cdiv(total, 256)), dim3(256), stream, rowptr, col, val, (int)n_rows,
             (int)op->tile_rows, (int)op->halo, (int)op->width, (int)op->n_tiles, (const float*)scale, slots, vals, info,
             far_col, far_cnt, (int)op->far_rows);
  return pgt_check_launch("pgt_ellw_build");
}

extern "C" int pgt_spmm_ellw_f32(const pgt_ellw* op, const int32_t* rowptr, const int32_t* col, const float* val,
                                 int64_t n_rows, const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T,
                                 int64_t ldt, float alpha, float beta, int64_t F, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && F >= 0, "pgt_spmm_ellw_f32: negative size");
  if (n_rows == 0 || F == 0) return PGT_OK;
  if (int rc = spmm_validate("pgt_spmm_ellw_f32", rowptr, n_rows, X, ldx, Y, ldy, T, ldt, F)) return rc;
  PGT_REQUIRE(col && val, "pgt_spmm_ellw_f32: the CSR operator the layout was built from is required");
  if (int rc = ellw_check("pgt_spmm_ellw_f32", op, n_rows)) return rc;
  PGT_REQUIRE((op->vals != nullptr) != (op->scale != nullptr), "pgt_spmm_ellw_f32: exactly one of vals / scale must be set");
  PGT_REQUIRE((op->order != nullptr) == (op->config == 3),
              "pgt_spmm_ellw_f32: config 3 is the launch shape of a renumbered layout (order set), and only that");
  PgtVecPick vp;
  vp.width(F); vp.operand(X, ldx); vp.operand(Y, ldy); vp.operand(T, ldt);
  const int64_t max_ld = ldx > ldy ? (ldx > ldt ? ldx : ldt) : (ldy > ldt ? ldy : ldt);
  // shapes the window kernel does not cover run the CSR row tiles (same sums, fmaf chain)
  const bool window_ok = F % 64 == 0 && F / 64 <= 65535 && vp.v == 4 && (n_rows + EllwCfgA::WRMAX) * max_ld < ((int64_t)1 << 31);
  // a renumbered layout's CSR is in layout numbering: the CSR kernels on it would answer in the wrong row space
  PGT_REQUIRE(op->order == nullptr || window_ok,
              "pgt_spmm_ellw_f32: a renumbered layout covers F = 64 k on 16-byte aligned operands only; use pgt_spmm_csr_f32 "
              "on the caller's CSR for this shape");
  if (op->order == nullptr && (!g_ellw || !window_ok))
    return pgt_spmm_csr_f32(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  const int flags = (g_tile_xcd ? 1 : 0) | (g_tile_nt ? 2 : 0);
  dim3 grid((unsigned)op->n_tiles, (unsigned)(F / 64));   // y: 64-float column chunks (1 for the F = 64 north-star shape)
  // the hubs' pieces ride with the tiles at F = 64 (one column chunk); otherwise the caller produces the hub rows (pgt_spmm_csr_rows_f32)
  const bool fold = op->hub_col != nullptr && F == 64;
  EllwHub hub = {op->hub_col, op->hub_val, op->hub_partial, fold ? (int)(op->n_hub * op->hub_split) : 0};
#define PGT_ELLW_GO_(MODE_, CFG_, W8C_, HUB_)                                                                         \
  PGT_LAUNCH((spmm_ellw64_kernel<MODE_, CFG_, W8C_, std::is_same<CFG_, EllwCfgC>::value, HUB_>), grid, dim3(CFG_::THREADS), stream, \
             op->slots, op->vals, op->scale, rowptr, col, val, (int)n_rows, (int)op->tile_rows, (int)op->halo,            \
             (int)op->width, X, (int)ldx, Y, (int)ldy, T, (int)ldt, alpha, beta, flags, op->far_col, op->order, hub, op->far_src)
#define PGT_ELLW_GO(MODE_, CFG_, W8C_)                                                   \
  do {                                                                                   \
    if (fold) PGT_ELLW_GO_(MODE_, CFG_, W8C_, true);                                     \
    else PGT_ELLW_GO_(MODE_, CFG_, W8C_, false);                                         \
  } while (0)
#define PGT_ELLW_W(MODE_, CFG_)                                  \
  do {                                                           \
    if (op->width == 8) PGT_ELLW_GO(MODE_, CFG_, 1);             \
    else if (op->width == 16) PGT_ELLW_GO(MODE_, CFG_, 2);       \
    else PGT_ELLW_GO(MODE_, CFG_, 0);                            \
  } while (0)
  if (op->scale != nullptr) {
    if (op->config == 1) PGT_ELLW_W(0, EllwCfgA); else if (op->config == 2) PGT_ELLW_W(0, EllwCfgB); else PGT_ELLW_W(0, EllwCfgC);
  } else {
    if (op->config == 1) PGT_ELLW_W(1, EllwCfgA); else if (op->config == 2) PGT_ELLW_W(1, EllwCfgB); else PGT_ELLW_W(1, EllwCfgC);
  }
#undef PGT_ELLW_W
#undef PGT_ELLW_GO
#undef PGT_ELLW_GO_
  if (fold) {
    if (int rc = pgt_check_launch("pgt_spmm_ellw_f32")) return rc;
    const int waves = (op->config == 2 ? EllwCfgB::THREADS : EllwCfgA::THREADS) / 64;
    PGT_LAUNCH(ellw_hub_combine_kernel, dim3((unsigned)op->n_hub), dim3(1024), stream, op->hub_rows, (int)op->hub_split * waves,
               (const float*)op->hub_partial, Y, (int)ldy, T, (int)ldt, alpha, beta);
  }
  return pgt_check_launch("pgt_spmm_ellw_f32");
}

extern "C" int pgt_csr_locality(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t* out4,
                                int32_t* long_rows, int64_t long_cap, int32_t long_len, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && long_cap >= 0, "pgt_csr_locality: negative size");
  PGT_REQUIRE(out4 != nullptr, "pgt_csr_locality: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) - 256 && long_cap < ((int64_t)1 << 31), "pgt_csr_locality: size exceeds int32 indexing");
  if (hipMemsetAsync(out4, 0, 4 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
    pgt_set_error("pgt_csr_locality: memset failed");
    return PGT_ERR_LAUNCH;
  }
  if (n_rows == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && col, "pgt_csr_locality: null pointer");
  PGT_LAUNCH(csr_locality_kernel, dim3((unsigned)pgt_cdiv(n_rows, 256)), dim3(256), stream, rowptr, col, (int)n_rows,
             out4, long_cap > 0 ? long_rows : (int32_t*)nullptr, (int)long_cap, (int)long_len);
  return pgt_check_launch("pgt_csr_locality");
}

extern "C" int pgt_spmm_csr_long_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                     const int32_t* long_rows, int64_t n_long, int32_t long_len, const float* X,
                                     int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt, float alpha,
                                     float beta, int64_t F, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && F >= 0 && n_long >= 0, "pgt_spmm_csr_long_f32: negative size");
  if (n_rows == 0 || F == 0) return PGT_OK;
  if (int rc = spmm_validate("pgt_spmm_csr_long_f32", rowptr, n_rows, X, ldx, Y, ldy, T, ldt, F)) return rc;
  PGT_REQUIRE(n_long == 0 || (long_rows != nullptr && long_len > 0), "pgt_spmm_csr_long_f32: long-row list missing");
  PGT_REQUIRE(n_long < ((int64_t)1 << 31), "pgt_spmm_csr_long_f32: too many long rows");
  PgtVecPick vp;
  vp.width(F); vp.operand(X, ldx); vp.operand(Y, ldy); vp.operand(T, ldt);
  const int64_t Fv = F / vp.v;
  // rows wider than the tile kernels cover (node-major batches) already spread a row over many wavefronts
  if (n_long == 0 || Fv > 64) return pgt_spmm_csr_f32(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  int rc;
  if (vp.v == 4) rc = launch_spmm<4>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream, long_len);
  else if (vp.v == 2) rc = launch_spmm<2>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream, long_len);
  else rc = launch_spmm<1>(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream, long_len);
  if (rc) return rc;
  int lpr = 4;
  while (lpr < Fv) lpr <<= 1;
  dim3 grid((unsigned)n_long), block(1024);
  if (vp.v == 4) PGT_LAUNCH((spmm_long_rows_kernel<4>), grid, block, stream, rowptr, col, val, long_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  else if (vp.v == 2) PGT_LAUNCH((spmm_long_rows_kernel<2>), grid, block, stream, rowptr, col, val, long_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  else PGT_LAUNCH((spmm_long_rows_kernel<1>), grid, block, stream, rowptr, col, val, long_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  return pgt_check_launch("pgt_spmm_csr_long_f32");
}

extern "C" int pgt_spmm_csr_rows_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                     const int32_t* rows, int64_t n_listed, const float* X, int64_t ldx, float* Y,
                                     int64_t ldy, const float* T, int64_t ldt, float alpha, float beta, int64_t F,
                                     pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && F >= 0 && n_listed >= 0, "pgt_spmm_csr_rows_f32: negative size");
  if (n_rows == 0 || F == 0 || n_listed == 0) return PGT_OK;
  if (int rc = spmm_validate("pgt_spmm_csr_rows_f32", rowptr, n_rows, X, ldx, Y, ldy, T, ldt, F)) return rc;
  PGT_REQUIRE(rows != nullptr && col != nullptr && val != nullptr, "pgt_spmm_csr_rows_f32: null pointer");
  PGT_REQUIRE(n_listed < ((int64_t)1 << 31), "pgt_spmm_csr_rows_f32: too many rows");
  PgtVecPick vp;
  vp.width(F); vp.operand(X, ldx); vp.operand(Y, ldy); vp.operand(T, ldt);
  PGT_REQUIRE(F / vp.v <= 64, "pgt_spmm_csr_rows_f32: rows of %lld floats exceed the kernel's 64 lanes x %d floats (node-major batches "
              "spread a row over many wavefronts already: pgt_spmm_csr_f32)", (long long)F, vp.v);
  int lpr = 4;
  while (lpr < F / vp.v) lpr <<= 1;
  dim3 grid((unsigned)n_listed), block(1024);
  if (vp.v == 4) PGT_LAUNCH((spmm_long_rows_kernel<4>), grid, block, stream, rowptr, col, val, rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  else if (vp.v == 2) PGT_LAUNCH((spmm_long_rows_kernel<2>), grid, block, stream, rowptr, col, val, rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  else PGT_LAUNCH((spmm_long_rows_kernel<1>), grid, block, stream, rowptr, col, val, rows, X, ldx, Y, ldy, T, ldt, alpha, beta, (int)F, lpr);
  return pgt_check_launch("pgt_spmm_csr_rows_f32");
}

extern "C" int pgt_spmm_csr_att_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                    const float* S, int64_t n_rows, int64_t B, int64_t C, const float* X,
                                    float* Y, int transpose_s, pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && B >= 0 && C >= 0, "pgt_spmm_csr_att_f32: negative size");
  if (n_rows == 0 || B == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && S && X && Y, "pgt_spmm_csr_att_f32: null pointer");
  PGT_REQUIRE(Y != X, "pgt_spmm_csr_att_f32: Y must not alias X");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) && B < ((int64_t)1 << 31) && C < ((int64_t)1 << 31),
              "pgt_spmm_csr_att_f32: size exceeds int32 indexing");
  const int64_t total = n_rows * B * C;
  PGT_REQUIRE(pgt_cdiv(total, 256) < ((int64_t)1 << 31), "pgt_spmm_csr_att_f32: grid too large");
  dim3 grid((unsigned)pgt_cdiv(total, 256)), block(256);
  PGT_LAUNCH(spmm_att_kernel, grid, block, stream, rowptr, col, val, S, (int)n_rows, (int)B, (int)C, X, Y,
             transpose_s ? 1 : 0);
  return pgt_check_launch("pgt_spmm_csr_att_f32");
}

extern "C" int pgt_sddmm_att_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                 int64_t B, int64_t C, const float* G, const float* X, float* dS,
                                 pgt_stream_t stream) {
  PGT_REQUIRE(n_rows >= 0 && B >= 0 && C >= 0, "pgt_sddmm_att_f32: negative size");
  if (n_rows == 0 || B == 0 || C == 0) return PGT_OK;
  PGT_REQUIRE(rowptr && G && X && dS, "pgt_sddmm_att_f32: null pointer");
  PGT_REQUIRE(n_rows < ((int64_t)1 << 31) && B < ((int64_t)1 << 31) && C < ((int64_t)1 << 31),
              "pgt_sddmm_att_f32: size exceeds int32 indexing");
  PGT_LAUNCH(sddmm_att_kernel, dim3((unsigned)n_rows), dim3(256), stream, rowptr, col, val, (int)n_rows, (int)B, (int)C,
             G, X, dS);
  return pgt_check_launch("pgt_sddmm_att_f32");
}
