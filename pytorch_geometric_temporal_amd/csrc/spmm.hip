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
int g_tile_nt = 1;     // pgt_tune("spmm_tile_nt"): non-temporal stores of the aggregated rows: 1 = when Y exceeds the L2s
                       // (>= 32 MiB: 33.5 vs 34.4 us at N = 200 k, F = 64), 2 = always, 0 = never
int g_tile_rows = 32;  // rows per tile for the F = 64 fast path (32 | 64 | 128); 32: finer tail, measured best
int g_unroll = 8;      // neighbour loads in flight per lane group (4 | 8)
int g_wide_xcd = 1;    // XCD-slab block mapping of the wide kernel
int g_quad = 0;        // F = 64: 1 = barrier-free persistent quad kernel instead of the row-tile kernel (measured tie)
int g_quad_blocks = 7; // quad kernel: resident workgroups per CU (70 VGPRs -> 7 wavefronts per SIMD)
int g_band_blocks = 3;  // band kernel: resident workgroups per CU the chunking aims at (<= 3: 160-VGPR kernel)
int g_band_xcd = 1;     // band kernel: contiguous chunk ranges per XCD
int g_band_cu = 4;      // locality-ordered F = 64 schedule: 4 = 32-row tiles with the X window in LDS (spmm_wtile64_kernel,
                        // measured best), 3 = 64-row tiles, 1 / 2 = one / two 1024-thread workgroups per CU, 0 = small ring workgroups
int g_wtile_wgs = 0;     // window-tile kernel: persistent workgroups per CU (0 = 5)
int g_wtile_tpw = 1;     // window-tile kernel: tiles per workgroup (1 = independent workgroups, measured best: 31.5 us;
                         // 0 = persistent, g_wtile_wgs per CU with the next tile prefetched: 36.4 us)
int g_band_nblk = 0;    // test hook: number of workgroups of the per-CU band kernel (0 = one or two per CU)

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
    int64_t ldt, float alpha, float beta, int F, int xcd_remap) {
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
// spmm_wtile64_kernel<TR, H> — F = 64, locality-ordered operator: the row-tile schedule with the tile's X WINDOW in
// LDS.  The degree sweep shows the row-tile kernel paying ~1.5 us per neighbour at the vector L1's
// 64 B/clk/CU (the gather re-reads every neighbour row through L1) on top of a 24 us streaming base, while the
// LDS-window kernels gather for free but carry a 34 us base of steps and barriers.  Here a 256-thread workgroup owns
// TR rows and first loads rows [r0 - H, r0 + TR + H) of X into LDS with fully coalesced float4 reads, all in flight
// at once (a window is read (TR + 2H) / TR times through L1 instead of `degree` times), together with the rowptr
// slice; the (col, val) slots follow (they need rowptr), and the whole gather is then LDS reads (ds_read_b128, one
// 256-byte row per 16-lane group).  Neighbours outside the window are read from global memory, so any operator is
// handled.  No steps, no ring, two barriers: independent workgroups overlap each other's round trips.
template <int TR, int H>
__global__ __launch_bounds__(256) void spmm_wtile64_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int ldx, float* Y, int ldy, const float* T, int ldt,
    float alpha, float beta, int tiles_per_wg, int n_tiles, int xcd_remap) {
  constexpr int WR = TR + 2 * H;             // window rows
  constexpr int CAP = 16 * TR;               // staged slots per tile; fuller tiles read their slots from global
  constexpr int XPT = WR * 16 / 256;         // float4 window loads per thread
  constexpr int SPT = CAP / 256;             // staged slots per thread
  __shared__ pgt_f4 s_x[WR * 16];
  __shared__ int s_rp[TR + 1];
  __shared__ int s_col[CAP];
  __shared__ float s_val[CAP];

  const int tid = threadIdx.x;
  const int wg = xcd_remap ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int t_first = wg * tiles_per_wg;
  const int t_last = (t_first + tiles_per_wg < n_tiles) ? t_first + tiles_per_wg : n_tiles;
  if (t_first >= t_last) return;
  const int l16 = tid & 15, rg = tid >> 4;   // 16 row-groups of 16 lanes
  const float* Xl = X + l16 * 4;

  // One memory phase per tile, requested while the PREVIOUS tile is gathered out of LDS: the tile's slot range comes
  // from two wave-uniform (scalar) reads of rowptr, so the (col, val) requests go out right behind the X window and
  // the rowptr slice instead of after a barrier and a second round trip.
  int p_e0 = 0, p_nnz = 0, p_rp = 0;
  pgt_f4 xw[XPT];
  int cq[SPT];
  float vq[SPT];
  auto fetch = [&](int tile) {
    const int r0 = tile * TR;
    const int nr = (n_rows - r0 < TR) ? (n_rows - r0) : TR;
    p_e0 = rowptr[r0];
    p_nnz = rowptr[r0 + nr] - p_e0;
    p_rp = rowptr[r0 + (tid <= nr ? tid : nr)];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int r = r0 - H + rg + 16 * i;
      r = r < 0 ? 0 : (r < n_rows ? r : n_rows - 1);
      xw[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(r * ldx));
    }
    const bool st = p_nnz <= CAP;
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      const int q = tid + 256 * i;
      const int qc = p_nnz > 0 ? ((st && q < p_nnz) ? p_e0 + q : p_e0) : 0;   // clamped: always a valid slot
      cq[i] = col[qc];
      vq[i] = val[qc];
    }
  };

  fetch(t_first);
  for (int tile = t_first; tile < t_last; ++tile) {
    const int r0 = tile * TR;
    const int nr = (n_rows - r0 < TR) ? (n_rows - r0) : TR;
    const int w0 = r0 - H;                   // first window row (may be negative: clamped loads, never matched)
    const int e0 = p_e0, nnz = p_nnz;
    const bool staged = nnz <= CAP;
    __syncthreads();                         // the previous tile's gather no longer reads LDS
    if (tid <= nr) s_rp[tid] = p_rp;
#pragma unroll
    for (int i = 0; i < XPT; ++i) s_x[(rg + 16 * i) * 16 + l16] = xw[i];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
      const int q = tid + 256 * i;
      if (staged && q < nnz) { s_col[q] = cq[i]; s_val[q] = vq[i]; }
    }
    __syncthreads();
    if (tile + 1 < t_last) fetch(tile + 1);  // in flight while this tile is gathered and stored
    // gather out of the window, sequential fma chain in slot order (bit-identical to the row-tile kernel)
    const int w_lo = w0 < 0 ? 0 : w0;
    const int w_hi = (w0 + WR < n_rows) ? w0 + WR : n_rows;
    for (int r = rg; r < nr; r += 16) {
      const int a = s_rp[r] - e0, b = s_rp[r + 1] - e0;
      pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
      int q = a;
      for (; q + 4 <= b; q += 4) {
        int c[4];
        float v[4];
        pgt_f4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          c[u] = staged ? s_col[q + u] : col[e0 + q + u];
          v[u] = staged ? s_val[q + u] : val[e0 + q + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (c[u] >= w_lo && c[u] < w_hi) x[u] = s_x[(c[u] - w0) * 16 + l16];
          else x[u] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(c[u] * ldx));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x = fmaf(v[u], x[u].x, acc.x); acc.y = fmaf(v[u], x[u].y, acc.y);
          acc.z = fmaf(v[u], x[u].z, acc.z); acc.w = fmaf(v[u], x[u].w, acc.w);
        }
      }
      for (; q < b; ++q) {
        const int c0 = staged ? s_col[q] : col[e0 + q];
        const float v0 = staged ? s_val[q] : val[e0 + q];
        pgt_f4 x0;
        if (c0 >= w_lo && c0 < w_hi) x0 = s_x[(c0 - w0) * 16 + l16];
        else x0 = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(c0 * ldx));
        acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y);
        acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
      }
      pgt_f4 out;
      if (T != nullptr) {
        const pgt_f4 t = *reinterpret_cast<const pgt_f4*>(T + (unsigned)((r0 + r) * ldt) + l16 * 4);
        out = pgt_mk4(alpha * acc.x + beta * t.x, alpha * acc.y + beta * t.y, alpha * acc.z + beta * t.z,
                      alpha * acc.w + beta * t.w);
      } else {
        out = pgt_mk4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
      }
      *reinterpret_cast<pgt_f4*>(Y + (unsigned)((r0 + r) * ldy) + l16 * 4) = out;
    }
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
// Work distribution: static.  The rows are cut into gridDim.x contiguous chunks (rows_per_chunk each, a multiple of 4)
// and chunk ranges are contiguous per XCD; every chunk is swept in `steps` equal steps of `srows` <= 64 rows.
// (A dynamic per-XCD ticket counter was measured and rejected: at N = 200 000 a workgroup only owns ~3 steps, so the
// per-chunk prologue it adds costs far more than the finish-time spread it removes: 57 us vs 33 us.)
template <int RING>
__global__ __launch_bounds__(256, (RING == 128 ? 3 : 2)) void spmm_band64_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int ldx, float* Y, int ldy, const float* T, int ldt,
    float alpha, float beta, int rows_per_chunk, int srows, int xcd_remap) {
  constexpr int S = 64, H = (RING - S) / 2;  // S: ring capacity per step; the step actually advances `srows` rows
  __shared__ pgt_f4 s_x[RING * 16];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, l16 = lane & 15;
  const int chunk = xcd_remap ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int c0 = chunk * rows_per_chunk;
  if (c0 >= n_rows) return;  // whole workgroup
  const int c1 = (c0 + rows_per_chunk < n_rows) ? c0 + rows_per_chunk : n_rows;
  PGT_TRACE_MARK(0);
  {
  const int rg = tid >> 4;  // 16 row-groups of 16 lanes: one 256-B row each per staging pass
  const float* Xl = X + l16 * 4;

  // Software pipeline, one step (64 rows) deep: while step s is computed out of LDS, the X rows of step s+1
  // (registers t), the (col, val) slots of step s+1 (registers mc/mv) and rowptr of step s+2 are already in
  // flight, so no step waits on a global-memory round trip it issued itself.
  auto clampr = [&](int r) { return r < 0 ? 0 : (r < n_rows ? r : n_rows - 1); };
  auto load_rp = [&](int s0, int& r0, int& r1) {  // rowptr[s0 + lane], rowptr[s0 + lane + 1] (clamped at n_rows)
    const int i0 = s0 + lane < n_rows ? s0 + lane : n_rows;
    const int i1 = s0 + lane + 1 < n_rows ? s0 + lane + 1 : n_rows;
    r0 = rowptr[i0];
    r1 = rowptr[i1];
  };
  // first 16 slots of the wave's four row-quads of the step starting at s0
  auto load_cv = [&](int s0, int r0, int r1, int (&qa)[4], int (&qn)[4], int (&qc)[4], float (&qv)[4]) {
    const int send = s0 + srows < c1 ? s0 + srows : c1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = (wave + 4 * j) * 4 + g;
      const int a = __shfl(r0, rl), b = __shfl(r1, rl);
      qa[j] = a;
      qn[j] = (s0 + rl < send) ? b - a : 0;
      // unconditional loads from a clamped slot (keeps the number of loads in flight static, so the compiler can
      // count them instead of draining the queue); dead lanes are masked by `qn` at use
      const int q = (l16 < qn[j]) ? a + l16 : (b > 0 ? b - 1 : 0);
      qc[j] = col[q];
      qv[j] = val[q];
    }
  };

  pgt_f4 t[4], pre[4];
  int rpA0, rpA1, rpB0 = 0, rpB1 = 0;
  int qa[4], qn[4], qc[4];
  float qv[4];
  // (rowptr loads are always issued FIRST in a phase: vmcnt retires in order, so whatever is issued behind a load
  //  — including the Y stores — has to drain before that load's result can be used)
  load_rp(c0, rpA0, rpA1);
  if (c0 + srows < c1) load_rp(c0 + srows, rpB0, rpB1);
  {
    // leading half-window rows [c0 - H, c0 + H) and the first step's rows [c0 + H, c0 + S + H)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = c0 - H + rg + 16 * i;
      pre[i] = (i * 16 < 2 * H) ? *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(clampr(r) * ldx)) : pgt_mk4(0, 0, 0, 0);
    }
    pgt_f4 pre2[(2 * H > 64) ? (2 * H - 64) / 16 : 1];
    if constexpr (2 * H > 64) {
#pragma unroll
      for (int i = 4; i < 2 * H / 16; ++i)
        pre2[i - 4] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(clampr(c0 - H + rg + 16 * i) * ldx));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      t[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(clampr(c0 + H + rg + 16 * i) * ldx));
    load_cv(c0, rpA0, rpA1, qa, qn, qc, qv);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = c0 - H + rg + 16 * i;
      if (i * 16 < 2 * H && r >= 0 && r < n_rows) s_x[(r & (RING - 1)) * 16 + l16] = pre[i];
    }
    if constexpr (2 * H > 64) {
#pragma unroll
      for (int i = 4; i < 2 * H / 16; ++i) {
        const int r = c0 - H + rg + 16 * i;
        if (r >= 0 && r < n_rows) s_x[(r & (RING - 1)) * 16 + l16] = pre2[i - 4];
      }
    }
  }

  PGT_TRACE_MARK(1);
  // One step.  (ca, cn, cc, cv) hold this step's slots (loaded a step ago); the next step's go to (na, nn, nc, nv).
  // The two register sets ping-pong between calls: copying a set would make the compiler wait for its loads.
  auto step = [&](const int s0, int (&ca)[4], int (&cn)[4], int (&cc)[4], float (&cv)[4], int (&na)[4], int (&nn)[4],
                  int (&nc)[4], float (&nv)[4], int& rc0, int& rc1, int& rn0, int& rn1) {
    // rows [s0 + H, s0 + srows + H) of this step, fetched one step ago
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = s0 + H + rg + 16 * i;
      if (r < n_rows && rg + 16 * i < srows) s_x[(r & (RING - 1)) * 16 + l16] = t[i];
    }
    __syncthreads();
    PGT_TRACE_MARK(2 + 2 * ((s0 - c0) / srows));
    const bool more = s0 + srows < c1;
    // the next step's loads are issued before any of this step's LDS work
    if (more) {
      if (s0 + 2 * srows < c1) load_rp(s0 + 2 * srows, rn0, rn1);   // rowptr of step s+2
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(clampr(s0 + srows + H + rg + 16 * i) * ldx));
      load_cv(s0 + srows, rc0, rc1, na, nn, nc, nv);                 // slots of step s+1 (rowptr came a step ago)
    }
    const int send = more ? s0 + srows : c1;
    const int w_lo = (s0 - H > 0) ? s0 - H : 0;
    const int w_hi = (s0 + srows + H < n_rows) ? s0 + srows + H : n_rows;

    // (a lambda invoked with literal quad indices: the slot registers must never be indexed dynamically)
    auto quad = [&](const int j, const int a, const int n, int mc, float mv) {
      const int row = s0 + (wave + 4 * j) * 4 + g;
      pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
      for (int q0 = 0; __ballot(q0 < n) != 0ull; q0 += 16) {
        const bool mine = q0 + l16 < n;
        if (q0 > 0) {  // rows longer than the 16 prefetched slots
          mc = mine ? col[a + q0 + l16] : 0;
          mv = mine ? val[a + q0 + l16] : 0.f;
        }
        // wave-uniform: does any live slot of this 16-slot chunk point outside the resident window?
        const bool any_far = __ballot(mine && (mc < w_lo || mc >= w_hi)) != 0ull;
        for (int u0 = 0; u0 < 16 && __ballot(q0 + u0 < n) != 0ull; u0 += 4) {
          int c[4];
          float v[4];
          pgt_f4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { c[u] = __shfl(mc, u0 + u, 16); v[u] = __shfl(mv, u0 + u, 16); }
          // every ring slot is mapped LDS, so the read is unconditional (a dead slot's value is discarded below)
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = s_x[(c[u] & (RING - 1)) * 16 + l16];
          if (any_far) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (q0 + u0 + u < n && (c[u] < w_lo || c[u] >= w_hi))
                x[u] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(c[u] * ldx));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool live = q0 + u0 + u < n;  // select, not multiply-by-zero: a dead slot must not inject NaN
            acc.x = live ? fmaf(v[u], x[u].x, acc.x) : acc.x;
            acc.y = live ? fmaf(v[u], x[u].y, acc.y) : acc.y;
            acc.z = live ? fmaf(v[u], x[u].z, acc.z) : acc.z;
            acc.w = live ? fmaf(v[u], x[u].w, acc.w) : acc.w;
          }
        }
      }
      if (row < send) {
        pgt_f4 o;
        if (T != nullptr) {
          const pgt_f4 tt = *reinterpret_cast<const pgt_f4*>(T + (unsigned)(row * ldt + l16 * 4));
          o = pgt_mk4(alpha * acc.x + beta * tt.x, alpha * acc.y + beta * tt.y, alpha * acc.z + beta * tt.z,
                          alpha * acc.w + beta * tt.w);
        } else {
          o = pgt_mk4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
        }
        *reinterpret_cast<pgt_f4*>(Y + (unsigned)(row * ldy + l16 * 4)) = o;
      }
    };
    quad(0, ca[0], cn[0], cc[0], cv[0]);
    quad(1, ca[1], cn[1], cc[1], cv[1]);
    quad(2, ca[2], cn[2], cc[2], cv[2]);
    quad(3, ca[3], cn[3], cc[3], cv[3]);
    __syncthreads();
    PGT_TRACE_MARK(3 + 2 * ((s0 - c0) / srows));
  };
  int ra[4], rn[4], rc[4];
  float rv[4];
#pragma unroll 1
  for (int s0 = c0; s0 < c1; s0 += 2 * srows) {
    step(s0, qa, qn, qc, qv, ra, rn, rc, rv, rpB0, rpB1, rpA0, rpA1);
    if (s0 + srows < c1) step(s0 + srows, ra, rn, rc, rv, qa, qn, qc, qv, rpA0, rpA1, rpB0, rpB1);
  }
  }
  PGT_TRACE_MARK(15);
}

// ---------------------------------------------------------------------------------------------------------------
// spmm_band64_cu_kernel<RING> — the LDS-window schedule with ONE 1024-thread workgroup per CU.
// PMC analysis of the other schedules (profiles/r01e_pmc_ns.csv): a CU's vector L1 keeps ~64 read misses in flight;
// the neighbour gather of the tile / quad schedules produces 3x more L1 misses than the rows a CU needs (L1 hit rate
// 50 %), and although they hit in L2 (~500 cycles) they occupy the same miss slots as the HBM fetches (~1 500 cycles),
// which caps the HBM stream at ~60 % of what a copy achieves.  Serving the gather from LDS leaves the miss slots to the
// one-time HBM fetch of X, Y and the CSR arrays.  Compared with spmm_band64_kernel (3-4 small workgroups per CU, 3-5
// steps each) one workgroup owns all ~n_rows/256 rows of its CU: one prologue and 2H halo rows per CU instead of per
// chunk, 12+ pipelined steps, every wavefront gathers exactly one row-quad per step, and all CUs finish together.
template <int RING>
__global__ __launch_bounds__(1024) void spmm_band64_cu_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int n_rows, const float* __restrict__ X, int ldx, float* Y, int ldy, const float* T, int ldt,
    float alpha, float beta, int rows_per_chunk, int srows, int xcd_remap) {
  constexpr int S = 64, H = (RING - S) / 2;
  constexpr int PRE = 2 * H / 64;            // 64-row passes of the leading half-window
  constexpr int MAXCHUNK = 2048;             // rows per workgroup (rowptr slice staged in LDS)
  __shared__ pgt_f4 s_x[RING * 16];
  __shared__ int s_rp[MAXCHUNK + 1];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, l16 = lane & 15;
  const int rg = tid >> 4;                   // 64 row-groups of 16 lanes: one 256-B row each per staging pass
  const int chunk = xcd_remap ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int c0 = chunk * rows_per_chunk;
  if (c0 >= n_rows) return;  // whole workgroup
  const int c1 = (c0 + rows_per_chunk < n_rows) ? c0 + rows_per_chunk : n_rows;
  const float* Xl = X + l16 * 4;
  PGT_TRACE_MARK(0);

  auto clampr = [&](int r) { return r < 0 ? 0 : (r < n_rows ? r : n_rows - 1); };
  // this thread's float4 of row s0 + H + rg (the rows step s0 adds to the window).  Always issued — past the end of
  // the chunk every lane re-reads the chunk's last window row (an L1 hit) — so that the number of requests in flight
  // is the same on every path and the compiler can count them (s_waitcnt vmcnt(N)) instead of draining the queue.
  const int last_row = clampr(c1 + H - 1);
  auto load_x = [&](int s0) {
    const int r = s0 + H + rg;
    return *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)((r < c1 + H ? clampr(r) : last_row) * ldx));
  };
  // first 16 slots of this wavefront's row-quad of the step starting at s0 (wave w owns rows 4w .. 4w+3 of the step);
  // rowptr comes out of LDS, so the request depends on nothing that is still in flight
  auto load_cv = [&](int s0, int& qa, int& qn, int& qc, float& qv) {
    const int send = s0 + srows < c1 ? s0 + srows : c1;
    const int row = s0 + wave * 4 + g;
    const int rl = row < c1 ? row - c0 : c1 - c0;
    const int a = s_rp[rl], b = s_rp[rl < c1 - c0 ? rl + 1 : rl];
    qa = a;
    qn = (row < send) ? b - a : 0;
    const int q = (l16 < qn) ? a + l16 : (b > 0 ? b - 1 : 0);
    qc = col[q];
    qv = val[q];
  };

  // ---- prologue: rowptr slice -> LDS, leading half-window -> LDS, X rows of steps 0..2 -> registers
  for (int i = tid; i <= c1 - c0; i += 1024) s_rp[i] = rowptr[c0 + i];
  pgt_f4 pre[PRE];
#pragma unroll
  for (int i = 0; i < PRE; ++i) pre[i] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(clampr(c0 - H + rg + 64 * i) * ldx));
  pgt_f4 tA = load_x(c0), tB = load_x(c0 + srows), tC = load_x(c0 + 2 * srows);
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    const int r = c0 - H + rg + 64 * i;
    if (r >= 0 && r < n_rows) s_x[(r & (RING - 1)) * 16 + l16] = pre[i];
  }
  __syncthreads();   // s_rp visible
  int aA, nA, cA, aB = 0, nB = 0, cB = 0, aC = 0, nC = 0, cC = 0;
  float vA, vB = 0.f, vC = 0.f;
  load_cv(c0, aA, nA, cA, vA);
  load_cv(c0 + srows, aB, nB, cB, vB);
  PGT_TRACE_MARK(1);

  // One step.  tt: this step's X rows (requested three steps ago), refilled with the rows of step s0 + 3*srows.
  // (ca, cn, cc, cv): this step's slots; the slots of step s0 + 2*srows go to (na, nn, nc, nv).  The three register sets
  // rotate through the three calls of the unrolled loop: nothing that is in flight is ever copied.
  auto step = [&](const int s0, pgt_f4& tt, int ca, int cn, int cc, float cv, int& na, int& nn, int& nc, float& nv) {
    {
      const int r = s0 + H + rg;
      if (r < n_rows && rg < srows && s0 < c1) s_x[(r & (RING - 1)) * 16 + l16] = tt;
    }
    __syncthreads();
    tt = load_x(s0 + 3 * srows);
    load_cv(s0 + 2 * srows, na, nn, nc, nv);
    const int send = s0 + srows < c1 ? s0 + srows : c1;
    const int w_lo = (s0 - H > 0) ? s0 - H : 0;
    const int w_hi = (send + H < n_rows) ? send + H : n_rows;   // rows past send + H were not fetched (load_x clamps)
    const int row = s0 + wave * 4 + g;
    const int a = ca, n = cn;
    int mc = cc;
    float mv = cv;
    pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    for (int q0 = 0; __ballot(q0 < n) != 0ull; q0 += 16) {
      const bool mine = q0 + l16 < n;
      if (q0 > 0) {  // rows longer than the 16 prefetched slots
        mc = mine ? col[a + q0 + l16] : 0;
        mv = mine ? val[a + q0 + l16] : 0.f;
      }
      const bool any_far = __ballot(mine && (mc < w_lo || mc >= w_hi)) != 0ull;
      for (int u0 = 0; u0 < 16 && __ballot(q0 + u0 < n) != 0ull; u0 += 8) {
        int c[8];
        float v[8];
        pgt_f4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c[u] = __shfl(mc, u0 + u, 16); v[u] = __shfl(mv, u0 + u, 16); }
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = s_x[(c[u] & (RING - 1)) * 16 + l16];
        if (any_far) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (q0 + u0 + u < n && (c[u] < w_lo || c[u] >= w_hi))
              x[u] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(c[u] * ldx));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool live = q0 + u0 + u < n;
          acc.x = live ? fmaf(v[u], x[u].x, acc.x) : acc.x;
          acc.y = live ? fmaf(v[u], x[u].y, acc.y) : acc.y;
          acc.z = live ? fmaf(v[u], x[u].z, acc.z) : acc.z;
          acc.w = live ? fmaf(v[u], x[u].w, acc.w) : acc.w;
        }
      }
    }
    if (row < send) {
      pgt_f4 o;
      if (T != nullptr) {
        const pgt_f4 t4 = *reinterpret_cast<const pgt_f4*>(T + (unsigned)(row * ldt + l16 * 4));
        o = pgt_mk4(alpha * acc.x + beta * t4.x, alpha * acc.y + beta * t4.y, alpha * acc.z + beta * t4.z,
                    alpha * acc.w + beta * t4.w);
      } else {
        o = pgt_mk4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
      }
      *reinterpret_cast<pgt_f4*>(Y + (unsigned)(row * ldy + l16 * 4)) = o;
    }
    __syncthreads();
  };
#pragma unroll 1
  for (int s0 = c0; s0 < c1; s0 += 3 * srows) {
    // (steps past the end of the chunk are empty: no LDS write, no slots, no store — but the same barriers and requests)
    step(s0, tA, aA, nA, cA, vA, aC, nC, cC, vC);
    step(s0 + srows, tB, aB, nB, cB, vB, aA, nA, cA, vA);
    step(s0 + 2 * srows, tC, aC, nC, cC, vC, aB, nB, cB, vB);
  }
  PGT_TRACE_MARK(15);
}

// ---------------------------------------------------------------------------------------------------------------
// spmm_quad64_kernel — F = 64, barrier-free and persistent: every wavefront walks its own sequence of row-quads
// (4 rows x 16 lanes x float4).  No LDS, no workgroup barrier: a quad's (col, val) slots are fetched with one
// coalesced 16-slot read per row and broadcast with ds_bpermute; rowptr of quad i+2 and the slots of quad i+1 are in
// flight while the neighbour rows of quad i are gathered (two register sets ping-pong, nothing in flight is copied).
// Work split: XCD x (blockIdx % 8) owns a contiguous eighth of the quads, its wavefronts take them round-robin, so at
// any time one XCD works on a narrow band of rows (L2 locality) and the tail is one quad per wavefront, not one tile
// per workgroup (the tile kernel's second, ragged wave of workgroups cost 9 - 33 us of a 33 us launch).
__global__ __launch_bounds__(256) void spmm_quad64_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val, int n_rows,
    const float* __restrict__ X, int ldx, float* Y, int ldy, const float* T, int ldt, float alpha, float beta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l16 = lane & 15;
  const int x = (int)(blockIdx.x & 7u);
  const int nwx = (int)(gridDim.x >> 3) * 4;              // wavefronts per XCD
  const int wi = (int)(blockIdx.x >> 3) * 4 + wave;
  const int nq = (n_rows + 3) >> 2;
  const int qx = (nq + 7) >> 3;
  const int qend = (x + 1) * qx < nq ? (x + 1) * qx : nq;
  int q = x * qx + wi;
  if (q >= qend) return;
  const float* Xl = X + l16 * 4;

  auto load_rp = [&](int qq, int& a, int& b) {
    const int row = 4 * qq + g;
    a = rowptr[row < n_rows ? row : n_rows];
    b = rowptr[row + 1 < n_rows ? row + 1 : n_rows];
  };
  auto load_cv = [&](int a, int b, int& n, int& mc, float& mv) {   // first 16 slots of the row, one per lane
    n = b - a;
    const int idx = l16 < n ? a + l16 : (b > 0 ? b - 1 : 0);
    mc = col[idx];
    mv = val[idx];
  };
  auto gather_store = [&](int qq, int a, int n, int mc, float mv) {
    const int row = 4 * qq + g;
    pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
    for (int q0 = 0; __ballot(q0 < n) != 0ull; q0 += 16) {
      if (q0 > 0) {                                          // rows longer than the 16 prefetched slots
        const int idx = q0 + l16 < n ? a + q0 + l16 : a;
        mc = col[idx];
        mv = val[idx];
      }
      for (int u0 = 0; u0 < 16 && __ballot(q0 + u0 < n) != 0ull; u0 += 8) {
        int c[8];
        float v[8];
        pgt_f4 xx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c[u] = __shfl(mc, u0 + u, 16); v[u] = __shfl(mv, u0 + u, 16); }
#pragma unroll
        for (int u = 0; u < 8; ++u)   // dead slots re-read the last live source row (harmless, discarded below)
          xx[u] = *reinterpret_cast<const pgt_f4*>(Xl + (unsigned)(c[u] * ldx));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool live = q0 + u0 + u < n;
          acc.x = live ? fmaf(v[u], xx[u].x, acc.x) : acc.x;
          acc.y = live ? fmaf(v[u], xx[u].y, acc.y) : acc.y;
          acc.z = live ? fmaf(v[u], xx[u].z, acc.z) : acc.z;
          acc.w = live ? fmaf(v[u], xx[u].w, acc.w) : acc.w;
        }
      }
    }
    if (row < n_rows) {
      pgt_f4 o;
      if (T != nullptr) {
        const pgt_f4 tt = *reinterpret_cast<const pgt_f4*>(T + (unsigned)(row * ldt + l16 * 4));
        o = pgt_mk4(alpha * acc.x + beta * tt.x, alpha * acc.y + beta * tt.y, alpha * acc.z + beta * tt.z,
                    alpha * acc.w + beta * tt.w);
      } else {
        o = pgt_mk4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
      }
      *reinterpret_cast<pgt_f4*>(Y + (unsigned)(row * ldy + l16 * 4)) = o;
    }
  };

  int a0, b0, a1 = 0, b1 = 0;
  load_rp(q, a0, b0);
  if (q + nwx < qend) load_rp(q + nwx, a1, b1);
  int n0, c0;
  float v0;
  load_cv(a0, b0, n0, c0, v0);
  int s0 = a0;
#pragma unroll 1
  for (;;) {
    // --- even half: gather quad q from set 0; slots of q + nwx -> set 1; rowptr of q + 2 nwx -> (a0, b0)
    int n1 = 0, c1 = 0, s1 = 0;
    float v1 = 0.f;
    const bool has1 = q + nwx < qend;
    if (q + 2 * nwx < qend) load_rp(q + 2 * nwx, a0, b0);
    if (has1) { load_cv(a1, b1, n1, c1, v1); s1 = a1; }
    gather_store(q, s0, n0, c0, v0);
    if (!has1) break;
    q += nwx;
    // --- odd half: gather quad q from set 1; slots of q + nwx -> set 0; rowptr of q + 2 nwx -> (a1, b1)
    const bool has0 = q + nwx < qend;
    if (q + 2 * nwx < qend) load_rp(q + 2 * nwx, a1, b1);
    if (has0) { load_cv(a0, b0, n0, c0, v0); s0 = a0; }
    gather_store(q, s1, n1, c1, v1);
    if (!has0) break;
    q += nwx;
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
                int64_t F, pgt_stream_t stream) {
  const int64_t Fv = F / VEC;
  const int n = (int)n_rows, Fi = (int)F;
  dim3 block(256);
  if (Fv <= 64) {
#define PGT_SPMM_CASE(L, TR_, U_)                                                                             \
  PGT_LAUNCH((spmm_tile_kernel<VEC, L, TR_, U_>), dim3((unsigned)pgt_cdiv(n_rows, TR_)), block, stream, rowptr, \
             col, val, n, X, ldx, Y, ldy, T, ldt, alpha, beta, Fi, (g_tile_xcd ? 1 : 0) | ((g_tile_nt == 2 || (g_tile_nt == 1 && (int64_t)n * Fi * 4 >= ((int64_t)32 << 20))) ? 2 : 0))
    if (Fv <= 4) { PGT_SPMM_CASE(4, 64, 4); }
    else if (Fv <= 8) { PGT_SPMM_CASE(8, 64, 4); }
    else if (Fv <= 16) {
      if (VEC == 4 && Fi == 64 && g_quad && n_rows >= 1024) {
        const int64_t max_ld = ldx > ldy ? (ldx > ldt ? ldx : ldt) : (ldy > ldt ? ldy : ldt);
        if (n_rows * max_ld < ((int64_t)1 << 31)) {
          int64_t nblk = 256 * (int64_t)g_quad_blocks;              // persistent: g_quad_blocks workgroups per CU
          const int64_t need = pgt_cdiv(pgt_cdiv(n_rows, 4), 4);    // one quad per wavefront at least
          if (nblk > need) nblk = need;
          nblk = pgt_cdiv(nblk, 8) * 8;                             // every XCD gets the same number of workgroups
          PGT_LAUNCH(spmm_quad64_kernel, dim3((unsigned)nblk), block, stream, rowptr, col, val, n, X, (int)ldx, Y,
                     (int)ldy, T, (int)ldt, alpha, beta);
          return pgt_check_launch("pgt_spmm_csr_f32");
        }
      }
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
  if (strcmp(key, "spmm_quad") == 0) { g_quad = value; return 1; }
  if (strcmp(key, "spmm_quad_blocks") == 0) { g_quad_blocks = value > 0 ? value : 1; return 1; }
  if (strcmp(key, "spmm_band_blocks") == 0) { g_band_blocks = value > 0 ? value : 1; return 1; }
  if (strcmp(key, "spmm_band_xcd") == 0) { g_band_xcd = value; return 1; }
  if (strcmp(key, "spmm_band_cu") == 0) { g_band_cu = value; return 1; }
  if (strcmp(key, "spmm_wtile_wgs") == 0) { g_wtile_wgs = value; return 1; }
  if (strcmp(key, "spmm_wtile_tpw") == 0) { g_wtile_tpw = value; return 1; }
  if (strcmp(key, "spmm_band_nblk") == 0) { g_band_nblk = value > 0 ? value : 0; return 1; }
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
  const int64_t max_ld = ldx > ldy ? (ldx > ldt ? ldx : ldt) : (ldy > ldt ? ldy : ldt);
  if (halo <= 0 || halo > 96 || F != 64 || vp.v != 4 || n_rows * max_ld >= ((int64_t)1 << 31))
    return pgt_spmm_csr_f32(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  const int ring = halo <= 32 ? 128 : 256;
  if (g_band_cu == 4 && halo > 32)           // a +-96 window is 224 rows of LDS per 32-row tile: the plain tiles win
    return pgt_spmm_csr_f32(rowptr, col, val, n_rows, X, ldx, Y, ldy, T, ldt, alpha, beta, F, stream);
  if (g_band_cu == 3 || g_band_cu == 4) {    // row tiles with the X window in LDS (3: 64-row tiles, 4: 32-row tiles)
    const int tr = g_band_cu == 3 ? 64 : 32;
    const int64_t n_tiles = pgt_cdiv(n_rows, tr);
    // persistent workgroups, a contiguous run of tiles each (the next tile's memory phase overlaps this tile's gather)
    const int64_t resident = 256 * (int64_t)(g_wtile_wgs > 0 ? g_wtile_wgs : 5);
    const int64_t tpw = g_wtile_tpw > 0 ? g_wtile_tpw : pgt_cdiv(n_tiles, resident);
    dim3 grid((unsigned)pgt_cdiv(n_tiles, tpw)), block(256);
#define PGT_WTILE_GO(TR_, H_)                                                                                      \
  PGT_LAUNCH((spmm_wtile64_kernel<TR_, H_>), grid, block, stream, rowptr, col, val, (int)n_rows, X, (int)ldx, Y,   \
             (int)ldy, T, (int)ldt, alpha, beta, (int)tpw, (int)n_tiles, g_band_xcd)
    if (halo <= 32) { if (tr == 64) PGT_WTILE_GO(64, 32); else PGT_WTILE_GO(32, 32); }
    else { if (tr == 64) PGT_WTILE_GO(64, 96); else PGT_WTILE_GO(32, 96); }
#undef PGT_WTILE_GO
    return pgt_check_launch("pgt_spmm_csr_band_f32");
  }
  if (g_band_cu && pgt_cdiv(n_rows, g_band_nblk > 0 ? g_band_nblk : 256 * (int64_t)(g_band_cu >= 2 ? 2 : 1)) + 4 <= 2048) {
    // one 1024-thread workgroup per CU (g_band_cu == 2: two): each owns a contiguous 1/256 (1/512) of the rows
    int64_t nblk = g_band_nblk > 0 ? g_band_nblk : 256 * (int64_t)(g_band_cu >= 2 ? 2 : 1);
    int64_t rpc = pgt_cdiv(pgt_cdiv(n_rows, nblk), 4) * 4;
    if (rpc < 64) rpc = 64;
    const int64_t steps = pgt_cdiv(rpc, 64);
    const int64_t srows = pgt_cdiv(pgt_cdiv(rpc, steps), 4) * 4;
    nblk = pgt_cdiv(n_rows, rpc);
    dim3 grid((unsigned)nblk), block(1024);
    if (ring == 128) {
      PGT_LAUNCH((spmm_band64_cu_kernel<128>), grid, block, stream, rowptr, col, val, (int)n_rows, X, (int)ldx, Y,
                 (int)ldy, T, (int)ldt, alpha, beta, (int)rpc, (int)srows, g_band_xcd);
    } else {
      PGT_LAUNCH((spmm_band64_cu_kernel<256>), grid, block, stream, rowptr, col, val, (int)n_rows, X, (int)ldx, Y,
                 (int)ldy, T, (int)ldt, alpha, beta, (int)rpc, (int)srows, g_band_xcd);
    }
    return pgt_check_launch("pgt_spmm_csr_band_f32");
  }
  // chunking: g_band_blocks resident workgroups per CU (registers admit 3 at RING = 128, LDS 2 at RING = 256); each chunk is
  // swept in equal steps of at most 64 rows
  const int per_cu = ring == 128 ? (g_band_blocks < 3 ? g_band_blocks : 3) : (g_band_blocks < 2 ? g_band_blocks : 2);
  int64_t nblk = 256 * (int64_t)per_cu;
  int64_t rpc = pgt_cdiv(n_rows, nblk);
  if (rpc < 64) rpc = 64;
  const int64_t steps = pgt_cdiv(rpc, 64);
  const int64_t srows = pgt_cdiv(pgt_cdiv(rpc, steps), 4) * 4;
  rpc = srows * steps;
  nblk = pgt_cdiv(n_rows, rpc);
  dim3 grid((unsigned)nblk), block(256);
  if (ring == 128) {
    PGT_LAUNCH((spmm_band64_kernel<128>), grid, block, stream, rowptr, col, val, (int)n_rows, X, (int)ldx, Y, (int)ldy, T,
               (int)ldt, alpha, beta, (int)rpc, (int)srows, g_band_xcd);
  } else {
    PGT_LAUNCH((spmm_band64_kernel<256>), grid, block, stream, rowptr, col, val, (int)n_rows, X, (int)ldx, Y, (int)ldy, T,
               (int)ldt, alpha, beta, (int)rpc, (int)srows, g_band_xcd);
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
