// Small-graph diffusion stack: ONE launch produces T1o, T1i, T2o, T2i (dcrnn.py:85-106) — or, backward, folds the
// gradients of those four terms into d/dT0 — for a batch of samples that share one graph of N <= ~300 nodes
// (METR-LA 207, PeMS-BAY 325, Chickenpox 20, EnglandCovid 129).
//
// Layout: BATCH-major rows m = b*N + n, so one sample's [N, C] block is contiguous in every stack segment.
// A 1024-thread workgroup owns one sample at a time (persistent loop over samples): the sample's T0 block is read from
// HBM once (coalesced, contiguous ~55 KB), parked in LDS, and both hops of both directions are gathered out of LDS
// (ds_read_b64, one row = C floats); the CSR operators (rowptr/col/val of P_o and P_i, ~26 KB at METR-LA size) also
// live in LDS.  HBM traffic per stack: read 1 block, write 4 (forward) / read 5, write 1 (backward) — the per-hop
// launches of pgt_spmm_csr_f32 read 6 and write 4 (forward) / read 12, write 4 (backward).
// Accumulation per output row is sequential in slot order, exactly like pgt_spmm_csr_f32 (deterministic).
#include "pgt_common.h"

// lab/slab_lab.hip defines these to take the kernels apart (phase timeline, stores / gathers / loads removed); in the
// library they are compile-time constants
#ifndef PGT_LAB_SKIP
#define PGT_LAB_SKIP(bit) false
#endif
#ifndef PGT_TRACE_MARK2
#define PGT_TRACE_MARK2(iter, slot) do { } while (0)
#endif

namespace {

struct SlabArgs {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;
  int N, C, K, nnz_o, nnz_i;
  int n_samples;
  float* TS;            // segment s of sample b starts at TS + s*seg_stride + b*N*C
  int64_t seg_stride;
  int folded;           // backward only: the "- Tx_0" adjoint was folded into the weights (ops.fold_backward_weight)
  int sorted;           // quad kernels: rows handed out in descending order of their slot counts (pgt_tune("slab_sort"))
};

#ifdef PGT_EMU
constexpr int SLAB_CUS = 4;
#else
constexpr int SLAB_CUS = 256;
#endif
constexpr int SLAB_THREADS = 1024;
constexpr int MAXT_CAP = 8;  // tasks (row, V-float column group) per thread: N * C / V <= MAXT_CAP * 1024
                             // (the kernels are instantiated for 2 / 4 / 7 / 8 tasks per thread: register arrays)

// V = 4: TWO column pairs per lane (8-byte aligned each; a row of C = 66 floats is 16.5 of them, the last lane of a
// row carries one live pair).  The (col, val) slot read of an edge is then shared by four floats instead of two:
// the slot reads are half of the LDS traffic of the V = 2 form (profiles/r01h_pmc_slab.csv).
struct P2 { float2 a, b; };
template <int V>
struct VecT;
template <>
struct VecT<4> { typedef P2 type; };
template <>
struct VecT<2> { typedef float2 type; };
template <>
struct VecT<1> { typedef float type; };

__device__ __forceinline__ float2 fma2(float a, float2 x, float2 acc) {
  return make_float2(fmaf(a, x.x, acc.x), fmaf(a, x.y, acc.y));
}
__device__ __forceinline__ float fma2(float a, float x, float acc) { return fmaf(a, x, acc); }
__device__ __forceinline__ float2 zero2(float2) { return make_float2(0.f, 0.f); }
__device__ __forceinline__ float zero2(float) { return 0.f; }
// alpha * a + beta * b
__device__ __forceinline__ float2 axpby(float al, float2 a, float be, float2 b) {
  return make_float2(al * a.x + be * b.x, al * a.y + be * b.y);
}
__device__ __forceinline__ float axpby(float al, float a, float be, float b) { return al * a + be * b; }
__device__ __forceinline__ float2 add3(float2 a, float2 b, float2 c) { return make_float2(a.x + b.x + c.x, a.y + b.y + c.y); }
__device__ __forceinline__ float add3(float a, float b, float c) { return a + b + c; }
__device__ __forceinline__ P2 fma2(float w, P2 x, P2 acc) { P2 r; r.a = fma2(w, x.a, acc.a); r.b = fma2(w, x.b, acc.b); return r; }
__device__ __forceinline__ P2 zero2(P2) { P2 r; r.a = make_float2(0.f, 0.f); r.b = r.a; return r; }
__device__ __forceinline__ P2 axpby(float al, P2 x, float be, P2 y) { P2 r; r.a = axpby(al, x.a, be, y.a); r.b = axpby(al, x.b, be, y.b); return r; }
__device__ __forceinline__ P2 add3(P2 x, P2 y, P2 z) { P2 r; r.a = add3(x.a, y.a, z.a); r.b = add3(x.b, y.b, z.b); return r; }

// Element access at a FLOAT offset.  `full`: the second pair of a P2 exists (else it mirrors the first on loads and is
// not stored).  The same helpers serve global memory and LDS.
__device__ __forceinline__ void ldT(const float* p, bool, float& v) { v = *p; }
__device__ __forceinline__ void ldT(const float* p, bool, float2& v) { v = *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void ldT(const float* p, bool full, P2& v) {
  v.a = *reinterpret_cast<const float2*>(p);
  v.b = *reinterpret_cast<const float2*>(p + (full ? 2 : 0));
}
__device__ __forceinline__ void stT(float* p, bool, float v) { *p = v; }
__device__ __forceinline__ void stT(float* p, bool, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void stT(float* p, bool full, P2 v) {
  *reinterpret_cast<float2*>(p) = v.a;
  if (full) *reinterpret_cast<float2*>(p + 2) = v.b;
}
// task idx -> float offset of its element inside an [N, C] block: row idx / CV, column V * (idx % CV); CV = ceil(C / V)
template <int V>
__device__ __forceinline__ int task_offset(int idx, int CV, int C, int& r, int& c, bool& full) {
  r = idx / CV;
  c = (idx - r * CV) * V;
  full = (V < 4) || (c + 2 < C);
  return r * C + c;
}

// LDS carve-up (bytes): two [N*C] float blocks, then the two operators
struct SlabLds {
  float* bufA; float* bufB;
  int* rp_o; int* rp_i;
  int2* cv_o; int2* cv_i;   // slots packed as (col, bits of val): one ds_read_b64 per slot
};
__device__ __forceinline__ SlabLds carve(char* base, const SlabArgs& a) {
  SlabLds s;
  const size_t blk = (((size_t)a.N * a.C * 4) + 15) & ~(size_t)15;
  s.bufA = reinterpret_cast<float*>(base);
  s.bufB = reinterpret_cast<float*>(base + blk);
  char* p = base + 2 * blk;
  s.cv_o = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_o * 8;
  s.cv_i = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_i * 8;
  s.rp_o = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.rp_i = reinterpret_cast<int*>(p);
  return s;
}

static size_t slab_lds_bytes(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i) {
  const size_t blk = (((size_t)N * C * 4) + 15) & ~(size_t)15;
  return 2 * blk + 2 * (size_t)(N + 1) * 4 + 2 * (size_t)(nnz_o + nnz_i) * 4;
}

__device__ __forceinline__ float as_float(int v) {
  union { int i; float f; } u;
  u.i = v;
  return u.f;
}
__device__ __forceinline__ int as_int(float v) {
  union { int i; float f; } u;
  u.f = v;
  return u.i;
}

// sum over the slots of row r, sequential fma chain in slot order; four slots' LDS reads are in flight at a time
template <typename T>
__device__ __forceinline__ T gather_row(const int* __restrict__ rp, const int2* __restrict__ cv,
                                        const float* __restrict__ buf, int r, int c, int C) {
  T acc = zero2(T());
  int q = rp[r];
  const int e = rp[r + 1];
  for (; q + 4 <= e; q += 4) {
    int2 s4[4];
    T x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = cv[q + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const T*>(buf + s4[u].x * C + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma2(as_float(s4[u].y), x[u], acc);
  }
  for (; q < e; ++q) {
    const int2 s1 = cv[q];
    acc = fma2(as_float(s1.y), *reinterpret_cast<const T*>(buf + s1.x * C + c), acc);
  }
  return acc;
}

// the same row sum for the two-pair element (`full`: the lane's second pair exists); U slots' LDS reads in flight
template <typename T, int U = 4>
__device__ __forceinline__ T gather_row_p2(const int* __restrict__ rp, const int2* __restrict__ cv,
                                        const float* __restrict__ buf, int r, int c, int C, bool full) {
  T acc = zero2(T());
  int q = rp[r];
  const int e = rp[r + 1];
  for (; q + U <= e; q += U) {
    int2 s4[U];
    T x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s4[u] = cv[q + u];
#pragma unroll
    for (int u = 0; u < U; ++u) ldT(buf + s4[u].x * C + c, full, x[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fma2(as_float(s4[u].y), x[u], acc);
  }
  for (; q < e; ++q) {
    const int2 s1 = cv[q];
    T x1;
    ldT(buf + s1.x * C + c, full, x1);
    acc = fma2(as_float(s1.y), x1, acc);
  }
  return acc;
}

__device__ __forceinline__ void stage_csr(const SlabArgs& a, const SlabLds& s, int tid) {
  for (int i = tid; i <= a.N; i += SLAB_THREADS) { s.rp_o[i] = a.rp_o[i]; s.rp_i[i] = a.rp_i[i]; }
  for (int i = tid; i < a.nnz_o; i += SLAB_THREADS) { int2 t; t.x = a.col_o[i]; t.y = as_int(a.val_o[i]); s.cv_o[i] = t; }
  for (int i = tid; i < a.nnz_i; i += SLAB_THREADS) { int2 t; t.x = a.col_i[i]; t.y = as_int(a.val_i[i]); s.cv_i[i] = t; }
}

// forward: segments [T0 | T1o T1i | T2o T2i]; K = 2 or 3
template <int V, int LDS_BYTES, int MAXT>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_fwd_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = a.C / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);

  // Software pipeline over samples: the T0 block of the NEXT sample is fetched into registers while this sample's
  // hops run out of LDS, so no sample waits on its own HBM read.
  T t0n[MAXT];
  if ((int)blockIdx.x < a.n_samples) {
    const float* nb = a.TS + (int64_t)blockIdx.x * a.N * a.C;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      t0n[j] = reinterpret_cast<const T*>(nb)[idx < ntask ? idx : ntask - 1];
    }
  }
  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    T t0[MAXT], i1[MAXT];
    PGT_LDS_BARRIER();  // CSR staged (first pass) / every lane done with the LDS blocks of the previous sample
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      t0[j] = t0n[j];
      if (idx < ntask) reinterpret_cast<T*>(s.bufA)[idx] = t0[j];
    }
    PGT_LDS_BARRIER();
    if (b + (int)gridDim.x < a.n_samples) {
      const float* nb = a.TS + (int64_t)(b + (int)gridDim.x) * a.N * a.C;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        t0n[j] = reinterpret_cast<const T*>(nb)[idx < ntask ? idx : ntask - 1];
      }
    }
    // hop 1: T1o = P_o T0 (straight into bufB, which nobody reads during this hop), T1i = P_i T0 (registers: bufA
    // is still being read)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T o1 = gather_row<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C);
        i1[j] = gather_row<T>(s.rp_i, s.cv_i, s.bufA, r, c, a.C);
        reinterpret_cast<T*>(base + 1 * a.seg_stride)[idx] = o1;
        reinterpret_cast<T*>(base + 2 * a.seg_stride)[idx] = i1[j];
        if (a.K >= 3) reinterpret_cast<T*>(s.bufB)[idx] = o1;
      }
    }
    if (a.K < 3) continue;  // (uniform) K == 2: no second hop
    PGT_LDS_BARRIER();        // everyone has finished reading T0 out of bufA
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) reinterpret_cast<T*>(s.bufA)[idx] = i1[j];
    }
    PGT_LDS_BARRIER();
    // hop 2: T2 = 2 P T1 - T0   (Tx_0 is never advanced in the reference, dcrnn.py:106)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T o2 = gather_row<T>(s.rp_o, s.cv_o, s.bufB, r, c, a.C);
        const T i2 = gather_row<T>(s.rp_i, s.cv_i, s.bufA, r, c, a.C);
        reinterpret_cast<T*>(base + 3 * a.seg_stride)[idx] = axpby(2.0f, o2, -1.0f, t0[j]);
        reinterpret_cast<T*>(base + 4 * a.seg_stride)[idx] = axpby(2.0f, i2, -1.0f, t0[j]);
      }
    }
  }
}

// backward on the TRANSPOSED operators (a.rp_o = bwd_o ...): segments [G0 | G1o G1i | G2o G2i] -> G0 (in place)
//   K == 3:  G1d += 2 P_d^T G2d ;  G0 += P_o^T G1o + P_i^T G1i  [ - G2o - G2i unless folded ]
//   K == 2:  G0 += P_o^T G1o + P_i^T G1i
template <int V, int LDS_BYTES, int MAXT>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_bwd_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = a.C / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);
  // the pair of blocks that opens a sample's first gather phase: (G2o, G2i) for K == 3, (G1o, G1i) for K == 2;
  // the next sample's pair is prefetched into registers while this sample is processed
  const int64_t lead = (a.K >= 3 ? 3 : 1) * a.seg_stride;
  T pa[MAXT], pb[MAXT];
  auto prefetch = [&](int bb) {
    const float* nb = a.TS + (int64_t)bb * a.N * a.C + lead;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      const int ic = idx < ntask ? idx : ntask - 1;
      pa[j] = reinterpret_cast<const T*>(nb)[ic];
      pb[j] = reinterpret_cast<const T*>(nb + a.seg_stride)[ic];
    }
  };
  if ((int)blockIdx.x < a.n_samples) prefetch((int)blockIdx.x);

  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    PGT_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        reinterpret_cast<T*>(s.bufA)[idx] = pa[j];
        reinterpret_cast<T*>(s.bufB)[idx] = pb[j];
      }
    }
    PGT_LDS_BARRIER();
    if (b + (int)gridDim.x < a.n_samples) prefetch(b + (int)gridDim.x);
    if (a.K >= 3) {
      T g1o[MAXT], g1i[MAXT];
      // G1d' = G1d + 2 P_d^T G2d   (the elementwise operands are requested before the LDS gathers)
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        const int ic = idx < ntask ? idx : ntask - 1;
        g1o[j] = reinterpret_cast<const T*>(base + 1 * a.seg_stride)[ic];
        g1i[j] = reinterpret_cast<const T*>(base + 2 * a.seg_stride)[ic];
      }
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        if (idx < ntask) {
          const int r = idx / CV, c = (idx - r * CV) * V;
          g1o[j] = axpby(2.0f, gather_row<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C), 1.0f, g1o[j]);
          g1i[j] = axpby(2.0f, gather_row<T>(s.rp_i, s.cv_i, s.bufB, r, c, a.C), 1.0f, g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        if (idx < ntask) {
          reinterpret_cast<T*>(s.bufA)[idx] = g1o[j];
          reinterpret_cast<T*>(s.bufB)[idx] = g1i[j];
        }
      }
      PGT_LDS_BARRIER();
    }
    // G0 += P_o^T G1o' + P_i^T G1i'   [ - G2o - G2i unless folded ]
    T g0[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      g0[j] = reinterpret_cast<const T*>(base)[idx < ntask ? idx : ntask - 1];
    }
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T po = gather_row<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C);
        const T pi = gather_row<T>(s.rp_i, s.cv_i, s.bufB, r, c, a.C);
        T g = g0[j];
        if (a.K >= 3 && !a.folded) {  // G0 -= G2o + G2i (re-read: the unfolded form is the rare one)
          const T g2o = reinterpret_cast<const T*>(base + 3 * a.seg_stride)[idx];
          const T g2i = reinterpret_cast<const T*>(base + 4 * a.seg_stride)[idx];
          g = add3(g, axpby(-1.0f, g2o, 0.0f, g2o), axpby(-1.0f, g2i, 0.0f, g2i));
        }
        reinterpret_cast<T*>(base)[idx] = add3(g, po, pi);
      }
    }
  }
}

// forward, two column pairs per lane (V = 4): same schedule as dconv_slab_fwd_kernel, element access by float offset
template <int V, int LDS_BYTES, int MAXT>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_fwd_p2_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = (a.C + V - 1) / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);

  // this thread's tasks: float offset inside a block and whether the second pair exists.  Kept in registers only for
  // the two-pair element (a division per task); the narrower elements recompute them (offset = V * idx).
  int off_[V == 4 ? MAXT : 1];
  bool full_[V == 4 ? MAXT : 1];
  if constexpr (V == 4) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      int r, c;
      off_[j] = task_offset<V>(idx < ntask ? idx : ntask - 1, CV, a.C, r, c, full_[j]);
    }
  }
  auto LIVE = [&](int j) { return tid + j * SLAB_THREADS < ntask; };
  auto OFF = [&](int j) {
    if constexpr (V == 4) return off_[j];
    else { const int idx = tid + j * SLAB_THREADS; return (idx < ntask ? idx : ntask - 1) * V; }
  };
  auto FULL = [&](int j) { if constexpr (V == 4) return full_[j]; else return true; };

  // Software pipeline over samples: the T0 block of the NEXT sample is fetched into registers while this sample's
  // hops run out of LDS, so no sample waits on its own HBM read.
  T t0n[MAXT];
  if ((int)blockIdx.x < a.n_samples) {
    const float* nb = a.TS + (int64_t)blockIdx.x * a.N * a.C;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldT(nb + OFF(j), FULL(j), t0n[j]);
  }
  int it_ = 0;
  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x, ++it_) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    T t0[MAXT], i1[MAXT];
    PGT_TRACE_MARK2(it_, 0);
    PGT_LDS_BARRIER();  // CSR staged (first pass) / every lane done with the LDS blocks of the previous sample
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      t0[j] = t0n[j];
      if (LIVE(j)) stT(s.bufA + OFF(j), FULL(j), t0[j]);
    }
    PGT_LDS_BARRIER();
    PGT_TRACE_MARK2(it_, 1);
    if (b + (int)gridDim.x < a.n_samples && !PGT_LAB_SKIP(4)) {
      const float* nb = a.TS + (int64_t)(b + (int)gridDim.x) * a.N * a.C;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) ldT(nb + OFF(j), FULL(j), t0n[j]);
    }
    // hop 1: T1o = P_o T0 (straight into bufB, which nobody reads during this hop), T1i = P_i T0 (registers: bufA
    // is still being read)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (LIVE(j)) {
        const int idx = tid + j * SLAB_THREADS;
        const int r = idx / CV, c = (idx - r * CV) * V;
        T o1 = t0[j];
        i1[j] = t0[j];
        if (!PGT_LAB_SKIP(2)) {
          o1 = gather_row_p2<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C, FULL(j));
          i1[j] = gather_row_p2<T>(s.rp_i, s.cv_i, s.bufA, r, c, a.C, FULL(j));
        }
        if (!PGT_LAB_SKIP(1)) {
          stT(base + 1 * a.seg_stride + OFF(j), FULL(j), o1);
          stT(base + 2 * a.seg_stride + OFF(j), FULL(j), i1[j]);
        }
        if (a.K >= 3) stT(s.bufB + OFF(j), FULL(j), o1);
      }
    }
    PGT_TRACE_MARK2(it_, 2);
    if (a.K < 3) continue;  // (uniform) K == 2: no second hop
    PGT_LDS_BARRIER();        // everyone has finished reading T0 out of bufA
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (LIVE(j)) stT(s.bufA + OFF(j), FULL(j), i1[j]);
    PGT_LDS_BARRIER();
    PGT_TRACE_MARK2(it_, 3);
    // hop 2: T2 = 2 P T1 - T0   (Tx_0 is never advanced in the reference, dcrnn.py:106)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (LIVE(j)) {
        const int idx = tid + j * SLAB_THREADS;
        const int r = idx / CV, c = (idx - r * CV) * V;
        T o2 = t0[j], i2 = t0[j];
        if (!PGT_LAB_SKIP(2)) {
          o2 = gather_row_p2<T>(s.rp_o, s.cv_o, s.bufB, r, c, a.C, FULL(j));
          i2 = gather_row_p2<T>(s.rp_i, s.cv_i, s.bufA, r, c, a.C, FULL(j));
        }
        if (!PGT_LAB_SKIP(1)) {
          stT(base + 3 * a.seg_stride + OFF(j), FULL(j), axpby(2.0f, o2, -1.0f, t0[j]));
          stT(base + 4 * a.seg_stride + OFF(j), FULL(j), axpby(2.0f, i2, -1.0f, t0[j]));
        }
      }
    }
    PGT_TRACE_MARK2(it_, 4);
  }
}

// backward, two column pairs per lane (V = 4); see dconv_slab_bwd_kernel.  On the TRANSPOSED operators (a.rp_o = bwd_o ...): segments [G0 | G1o G1i | G2o G2i] -> G0 (in place)
//   K == 3:  G1d += 2 P_d^T G2d ;  G0 += P_o^T G1o + P_i^T G1i  [ - G2o - G2i unless folded ]
//   K == 2:  G0 += P_o^T G1o + P_i^T G1i
template <int V, int LDS_BYTES, int MAXT>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_bwd_p2_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = (a.C + V - 1) / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);
  // this thread's tasks: float offset inside a block and whether the second pair exists.  Kept in registers only for
  // the two-pair element (a division per task); the narrower elements recompute them (offset = V * idx).
  int off_[V == 4 ? MAXT : 1];
  bool full_[V == 4 ? MAXT : 1];
  if constexpr (V == 4) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      int r, c;
      off_[j] = task_offset<V>(idx < ntask ? idx : ntask - 1, CV, a.C, r, c, full_[j]);
    }
  }
  auto LIVE = [&](int j) { return tid + j * SLAB_THREADS < ntask; };
  auto OFF = [&](int j) {
    if constexpr (V == 4) return off_[j];
    else { const int idx = tid + j * SLAB_THREADS; return (idx < ntask ? idx : ntask - 1) * V; }
  };
  auto FULL = [&](int j) { if constexpr (V == 4) return full_[j]; else return true; };
  // the pair of blocks that opens a sample's first gather phase: (G2o, G2i) for K == 3, (G1o, G1i) for K == 2;
  // the next sample's pair is prefetched into registers while this sample is processed
  const int64_t lead = (a.K >= 3 ? 3 : 1) * a.seg_stride;
  T pa[MAXT], pb[MAXT];
  auto prefetch = [&](int bb) {
    const float* nb = a.TS + (int64_t)bb * a.N * a.C + lead;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      ldT(nb + OFF(j), FULL(j), pa[j]);
      ldT(nb + a.seg_stride + OFF(j), FULL(j), pb[j]);
    }
  };
  if ((int)blockIdx.x < a.n_samples) prefetch((int)blockIdx.x);

  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    PGT_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (LIVE(j)) {
        stT(s.bufA + OFF(j), FULL(j), pa[j]);
        stT(s.bufB + OFF(j), FULL(j), pb[j]);
      }
    }
    PGT_LDS_BARRIER();
    if (b + (int)gridDim.x < a.n_samples) prefetch(b + (int)gridDim.x);
    if (a.K >= 3) {
      T g1o[MAXT], g1i[MAXT];
      // G1d' = G1d + 2 P_d^T G2d   (the elementwise operands are requested before the LDS gathers)
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        ldT(base + 1 * a.seg_stride + OFF(j), FULL(j), g1o[j]);
        ldT(base + 2 * a.seg_stride + OFF(j), FULL(j), g1i[j]);
      }
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (LIVE(j)) {
          const int idx = tid + j * SLAB_THREADS;
          const int r = idx / CV, c = (idx - r * CV) * V;
          g1o[j] = axpby(2.0f, gather_row_p2<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C, FULL(j)), 1.0f, g1o[j]);
          g1i[j] = axpby(2.0f, gather_row_p2<T>(s.rp_i, s.cv_i, s.bufB, r, c, a.C, FULL(j)), 1.0f, g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (LIVE(j)) {
          stT(s.bufA + OFF(j), FULL(j), g1o[j]);
          stT(s.bufB + OFF(j), FULL(j), g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
    }
    // G0 += P_o^T G1o' + P_i^T G1i'   [ - G2o - G2i unless folded ]
    T g0[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldT(base + OFF(j), FULL(j), g0[j]);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (LIVE(j)) {
        const int idx = tid + j * SLAB_THREADS;
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T po = gather_row_p2<T>(s.rp_o, s.cv_o, s.bufA, r, c, a.C, FULL(j));
        const T pi = gather_row_p2<T>(s.rp_i, s.cv_i, s.bufB, r, c, a.C, FULL(j));
        T g = g0[j];
        if (a.K >= 3 && !a.folded) {  // G0 -= G2o + G2i (re-read: the unfolded form is the rare one)
          T g2o, g2i;
          ldT(base + 3 * a.seg_stride + OFF(j), FULL(j), g2o);
          ldT(base + 4 * a.seg_stride + OFF(j), FULL(j), g2i);
          g = add3(g, axpby(-1.0f, g2o, 0.0f, g2o), axpby(-1.0f, g2i, 0.0f, g2i));
        }
        stT(base + OFF(j), FULL(j), add3(g, po, pi));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Quad layout (round 3): whole-sample kernels for C = 64 + E columns (E = 0: the hidden-columns-only backward stack;
// E = 2: [X_t, H] of the benchmarked model).  lab/slab_lab.hip on the kernels above at B = 1024: 88 us whole, 52 us with the
// gathers replaced by copies, 41 us stores only — the LDS gathers (~36 us) ADD to the memory time, and they run at a
// quarter of the LDS rate: a 66-float row is 16.5 float2 pairs, so a 32-lane `ds_read_b64` group straddles two source
// rows whose 264-byte pitch puts them on the same banks (2-way conflict on every feature read), and a slot costs three
// LDS instructions per lane (slot, two pairs).  Here the 64 hidden columns of the block live in LDS with a 256-byte
// pitch and a lane owns one 16-byte quad of them: ONE `ds_read_b128` per slot, and the hardware's b128 lane groups
// ({0-3, 12-15, 20-27}, ...) then cover all 64 banks once whatever rows the lanes point at — conflict-free by
// construction.  The E leading columns (the input features) sit in a side block of one padded quad per row and are the
// 17th task of a row, so that every lane executes the same instruction stream.  Same fmaf chain per element as the
// kernels above: bit-identical results.
// global access of a task's element.  G4: 16-byte aligned quads.  Otherwise (C = 66: a row starts every 264 bytes) a hidden
// quad is 8-byte aligned: still ONE dwordx4 access — the hardware takes any dword-aligned address — and the leading pair of
// a row is a float2 (its upper half mirrors the lower on loads and is never stored).
template <bool G4>
__device__ __forceinline__ void ldQ(const float* p, bool main_, pgt_f4& v) {
  if constexpr (G4) {
    v = *reinterpret_cast<const pgt_f4*>(p);
  } else {
    if (main_) {
      __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), 16);
    } else {
      const float2 a = *reinterpret_cast<const float2*>(p);
      v = pgt_mk4(a.x, a.y, a.x, a.y);
    }
  }
}
template <bool G4>
__device__ __forceinline__ void stQ(float* p, bool main_, pgt_f4 v) {
  if constexpr (G4) {
    *reinterpret_cast<pgt_f4*>(p) = v;
  } else {
    if (main_) __builtin_memcpy(__builtin_assume_aligned(p, 8), &v, 16);
    else *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
  }
}
__device__ __forceinline__ pgt_f4 fma4(float w, pgt_f4 x, pgt_f4 acc) {
  return pgt_mk4(fmaf(w, x.x, acc.x), fmaf(w, x.y, acc.y), fmaf(w, x.z, acc.z), fmaf(w, x.w, acc.w));
}
__device__ __forceinline__ pgt_f4 axpby4(float al, pgt_f4 a, float be, pgt_f4 b) {
  return pgt_mk4(axpby(al, a.x, be, b.x), axpby(al, a.y, be, b.y), axpby(al, a.z, be, b.z), axpby(al, a.w, be, b.w));
}
__device__ __forceinline__ pgt_f4 add34(pgt_f4 a, pgt_f4 b, pgt_f4 c) {
  return pgt_mk4(add3(a.x, b.x, c.x), add3(a.y, b.y, c.y), add3(a.z, b.z, c.z), add3(a.w, b.w, c.w));
}
// row sum over the slots of row r: x = the quad at float offset `qoff + source * pitch` of the block; U slots in flight
template <int U = 4>
__device__ __forceinline__ pgt_f4 gather_q(const int* __restrict__ rp, const int2* __restrict__ cv,
                                           const float* __restrict__ blk, int r, int qoff, int pitch) {
  pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
  int q = rp[r];
  const int e = rp[r + 1];
  for (; q + U <= e; q += U) {
    int2 s4[U];
    pgt_f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s4[u] = cv[q + u];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const pgt_f4*>(blk + qoff + s4[u].x * pitch);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fma4(as_float(s4[u].y), x[u], acc);
  }
  for (; q < e; ++q) {
    const int2 s1 = cv[q];
    acc = fma4(as_float(s1.y), *reinterpret_cast<const pgt_f4*>(blk + qoff + s1.x * pitch), acc);
  }
  return acc;
}
// (Measured and dropped: both directions of a row in ONE loop over the longer slot list, 2 x 2 reads in flight — 108 - 127 us
// against 78 us: the predicated tail and the extra live registers cost more than the shorter chain saves.)
// floats of one LDS block: [N][64] hidden quads + (E > 0) one padded quad per row for the leading columns
__host__ __device__ inline size_t slab_q_block_floats(int64_t N, int E) { return (size_t)N * 64 + (E > 0 ? (size_t)N * 4 : 0); }
static size_t slab_q_lds_bytes(int64_t N, int E, int64_t nnz_o, int64_t nnz_i) {
  return 2 * slab_q_block_floats(N, E) * 4 + 2 * (size_t)(N + 1) * 4 + 2 * (size_t)(nnz_o + nnz_i) * 4 + 2 * (size_t)N * 4 + 16;
}
__device__ __forceinline__ SlabLds carve_q(char* base, const SlabArgs& a, int E) {
  SlabLds s;
  const size_t blk = slab_q_block_floats(a.N, E) * 4;
  s.bufA = reinterpret_cast<float*>(base);
  s.bufB = reinterpret_cast<float*>(base + blk);
  char* p = base + 2 * blk;
  s.cv_o = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_o * 8;
  s.cv_i = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_i * 8;
  s.rp_o = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.rp_i = reinterpret_cast<int*>(p);
  return s;
}
// Task tid + j * 1024 of a sample.  Tasks 0 .. 16 N - 1 are the hidden quads, sixteen per row, so that a wavefront's lanes
// 16 i .. 16 i + 15 own one row (the b128 lane groups then never meet on a bank); the N leading-pair tasks follow.
template <int E, int MAXT, int THREADS = SLAB_THREADS>
struct SlabQTasks {
  int rj[MAXT];     // main: (row << 5) | quad (0 .. 15);  leading pair of a row: (row << 5) | 16
  int N, C, ntask, tid;
  // `order` (LDS, may be null): position p -> row; the sixteen lanes of position p own row order[p]
  __device__ __forceinline__ void init(const SlabArgs& a, int tid_, const int* order = nullptr) {
    N = a.N; C = a.C; tid = tid_; ntask = a.N * (16 + (E > 0 ? 1 : 0));
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * THREADS, ic = idx < ntask ? idx : ntask - 1;
      const int pos = ic < 16 * a.N ? (ic >> 4) : ic - 16 * a.N;
      const int row = order ? order[pos] : pos;
      rj[j] = ic < 16 * a.N ? ((row << 5) | (ic & 15)) : ((row << 5) | 16);
    }
  }
  __device__ __forceinline__ bool live(int j) const { return tid + j * THREADS < ntask; }
  __device__ __forceinline__ int row(int j) const { return rj[j] >> 5; }
  __device__ __forceinline__ bool main_(int j) const { return E == 0 || (rj[j] & 16) == 0; }
  __device__ __forceinline__ int quad(int j) const { return rj[j] & 15; }
  __device__ __forceinline__ int goff(int j) const { return row(j) * C + (main_(j) ? E + 4 * quad(j) : 0); }   // in the sample's [N, C] block
  __device__ __forceinline__ int qoff(int j) const { return main_(j) ? 4 * quad(j) : N * 64; }               // quad column inside an LDS block
  __device__ __forceinline__ int pitch(int j) const { return main_(j) ? 64 : 4; }
  __device__ __forceinline__ int loff(int j) const { return qoff(j) + row(j) * pitch(j); }
};

// Rows by slot count (pgt_tune("slab_sort", 1); off by default: it measured no gain).  A wavefront's four rows are gathered in lock step, so it runs as long as its LONGEST row (METR-LA: 7.3 slots
// per row on average, up to 3x that): rows are handed to the lane groups in descending order of their two operators' slot counts —
// position p of the order is the row with p longer-or-earlier rows — so a wavefront's four rows end together and every wavefront
// gets rows from each quartile (its j-th task is 64 positions further down).  Which lane computes a row changes nothing in the
// row's sum: bit-identical.  Computed once per workgroup from the staged rowptr arrays (N <= ~330: a rank by counting).
__device__ __forceinline__ const int* slab_row_order(const SlabArgs& a, const SlabLds& s, int tid, bool enabled) {
  if (!enabled) return nullptr;
  int* keys = s.rp_i + (a.N + 1);
  int* order = keys + a.N;
  __syncthreads();                                            // rowptr arrays staged
  for (int r = tid; r < a.N; r += SLAB_THREADS) keys[r] = (s.rp_o[r + 1] - s.rp_o[r]) + (s.rp_i[r + 1] - s.rp_i[r]);
  __syncthreads();
  for (int r = tid; r < a.N; r += SLAB_THREADS) {
    const int kr = keys[r];
    int rank = 0;
    for (int q = 0; q < a.N; ++q) { const int kq = keys[q]; rank += (kq > kr) || (kq == kr && q < r); }
    order[rank] = r;
  }
  __syncthreads();
  return order;
}

// forward: segments [T0 | T1o T1i | T2o T2i]; K = 2 or 3 (see dconv_slab_fwd_kernel)
template <int E, bool G4, int MAXT>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_fwd_q_kernel(SlabArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
  const SlabLds s = carve_q(smem, a, E);
  const int tid = threadIdx.x;
  stage_csr(a, s, tid);
  SlabQTasks<E, MAXT> k;
  k.init(a, tid, slab_row_order(a, s, tid, a.sorted != 0));
  pgt_f4 t0n[MAXT];
  if ((int)blockIdx.x < a.n_samples) {
    const float* nb = a.TS + (int64_t)blockIdx.x * a.N * a.C;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldQ<G4>(nb + k.goff(j), k.main_(j), t0n[j]);
  }
  int it_ = 0;
  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x, ++it_) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    pgt_f4 t0[MAXT], i1[MAXT];
    PGT_TRACE_MARK2(it_, 0);
    PGT_LDS_BARRIER();  // CSR staged (first pass) / every lane done with the LDS blocks of the previous sample
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      t0[j] = t0n[j];
      if (k.live(j)) *reinterpret_cast<pgt_f4*>(s.bufA + k.loff(j)) = t0[j];
    }
    PGT_LDS_BARRIER();
    PGT_TRACE_MARK2(it_, 1);
    if (b + (int)gridDim.x < a.n_samples && !PGT_LAB_SKIP(4)) {
      const float* nb = a.TS + (int64_t)(b + (int)gridDim.x) * a.N * a.C;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) ldQ<G4>(nb + k.goff(j), k.main_(j), t0n[j]);
    }
    // hop 1: T1o = P_o T0 (into bufB, which nobody reads during this hop), T1i = P_i T0 (registers: bufA is being read)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        pgt_f4 o1 = t0[j];
        i1[j] = t0[j];
        if (!PGT_LAB_SKIP(2)) {
          o1 = gather_q(s.rp_o, s.cv_o, s.bufA, k.row(j), k.qoff(j), k.pitch(j));
          i1[j] = gather_q(s.rp_i, s.cv_i, s.bufA, k.row(j), k.qoff(j), k.pitch(j));
        }
        if (!PGT_LAB_SKIP(1)) {
          stQ<G4>(base + 1 * a.seg_stride + k.goff(j), k.main_(j), o1);
          stQ<G4>(base + 2 * a.seg_stride + k.goff(j), k.main_(j), i1[j]);
        }
        if (a.K >= 3) *reinterpret_cast<pgt_f4*>(s.bufB + k.loff(j)) = o1;
      }
    }
    PGT_TRACE_MARK2(it_, 2);
    if (a.K < 3) continue;  // (uniform) K == 2: no second hop
    PGT_LDS_BARRIER();        // everyone has finished reading T0 out of bufA
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (k.live(j)) *reinterpret_cast<pgt_f4*>(s.bufA + k.loff(j)) = i1[j];
    PGT_LDS_BARRIER();
    PGT_TRACE_MARK2(it_, 3);
    // hop 2: T2 = 2 P T1 - T0   (Tx_0 is never advanced in the reference, dcrnn.py:106)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        pgt_f4 o2 = t0[j], i2 = t0[j];
        if (!PGT_LAB_SKIP(2)) {
          o2 = gather_q(s.rp_o, s.cv_o, s.bufB, k.row(j), k.qoff(j), k.pitch(j));
          i2 = gather_q(s.rp_i, s.cv_i, s.bufA, k.row(j), k.qoff(j), k.pitch(j));
        }
        if (!PGT_LAB_SKIP(1)) {
          stQ<G4>(base + 3 * a.seg_stride + k.goff(j), k.main_(j), axpby4(2.0f, o2, -1.0f, t0[j]));
          stQ<G4>(base + 4 * a.seg_stride + k.goff(j), k.main_(j), axpby4(2.0f, i2, -1.0f, t0[j]));
        }
      }
    }
    PGT_TRACE_MARK2(it_, 4);
  }
}

// backward on the TRANSPOSED operators; see dconv_slab_bwd_kernel
template <int E, bool G4, int MAXT, int GU>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_bwd_q_kernel(SlabArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
  const SlabLds s = carve_q(smem, a, E);
  const int tid = threadIdx.x;
  stage_csr(a, s, tid);
  SlabQTasks<E, MAXT> k;
  k.init(a, tid, slab_row_order(a, s, tid, a.sorted != 0));
  const int64_t lead = (a.K >= 3 ? 3 : 1) * a.seg_stride;
  pgt_f4 pa[MAXT], pb[MAXT];
  auto prefetch = [&](int bb) {
    const float* nb = a.TS + (int64_t)bb * a.N * a.C + lead;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      ldQ<G4>(nb + k.goff(j), k.main_(j), pa[j]);
      ldQ<G4>(nb + a.seg_stride + k.goff(j), k.main_(j), pb[j]);
    }
  };
  if ((int)blockIdx.x < a.n_samples) prefetch((int)blockIdx.x);
  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    PGT_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        *reinterpret_cast<pgt_f4*>(s.bufA + k.loff(j)) = pa[j];
        *reinterpret_cast<pgt_f4*>(s.bufB + k.loff(j)) = pb[j];
      }
    }
    PGT_LDS_BARRIER();
    if (b + (int)gridDim.x < a.n_samples) prefetch(b + (int)gridDim.x);
    if (a.K >= 3) {
      pgt_f4 g1o[MAXT], g1i[MAXT];
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        ldQ<G4>(base + 1 * a.seg_stride + k.goff(j), k.main_(j), g1o[j]);
        ldQ<G4>(base + 2 * a.seg_stride + k.goff(j), k.main_(j), g1i[j]);
      }
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (k.live(j)) {
          g1o[j] = axpby4(2.0f, gather_q<GU>(s.rp_o, s.cv_o, s.bufA, k.row(j), k.qoff(j), k.pitch(j)), 1.0f, g1o[j]);
          g1i[j] = axpby4(2.0f, gather_q<GU>(s.rp_i, s.cv_i, s.bufB, k.row(j), k.qoff(j), k.pitch(j)), 1.0f, g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (k.live(j)) {
          *reinterpret_cast<pgt_f4*>(s.bufA + k.loff(j)) = g1o[j];
          *reinterpret_cast<pgt_f4*>(s.bufB + k.loff(j)) = g1i[j];
        }
      }
      PGT_LDS_BARRIER();
    }
    pgt_f4 g0[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldQ<G4>(base + k.goff(j), k.main_(j), g0[j]);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        const pgt_f4 po = gather_q<GU>(s.rp_o, s.cv_o, s.bufA, k.row(j), k.qoff(j), k.pitch(j));
        const pgt_f4 pi = gather_q<GU>(s.rp_i, s.cv_i, s.bufB, k.row(j), k.qoff(j), k.pitch(j));
        pgt_f4 g = g0[j];
        if (a.K >= 3 && !a.folded) {  // G0 -= G2o + G2i (re-read: the unfolded form is the rare one)
          pgt_f4 g2o, g2i;
          ldQ<G4>(base + 3 * a.seg_stride + k.goff(j), k.main_(j), g2o);
          ldQ<G4>(base + 4 * a.seg_stride + k.goff(j), k.main_(j), g2i);
          g = add34(g, axpby4(-1.0f, g2o, 0.0f, g2o), axpby4(-1.0f, g2i, 0.0f, g2i));
        }
        stQ<G4>(base + k.goff(j), k.main_(j), add34(g, po, pi));
      }
    }
  }
}

int g_slab_pairs = 2;   // pgt_tune("slab_pairs"): column pairs per lane of the LDS-resident stack kernels (1 | 2)
// (Measured and dropped in round 3: a single-block form of these kernels — ONE [N, 64] block + the packed slots in LDS, 80.5 KB,
// so that TWO 512-thread workgroups per CU work on whole samples and one's gathers run under the other's memory phase.  Seven
// tasks per thread, the hop's two directions taking turns in the block: 112 us forward / 99 us backward against 77 - 78 us for
// the kernels above.  With the column windows (103 - 123 us) that is the second design whose point was a second resident
// workgroup, and the second that lost: on this part the phases of co-resident workgroups do not interleave usefully for this
// access pattern.)
int g_slab_sort = 0;      // pgt_tune("slab_sort"): 1 = the quad kernels hand rows out by slot count (a wavefront's rows end together), 0 = in row
                          // order.  Measured in round 4 (same box, B = 1024): 80.3 -> 80.1 us per launch, 14.20 -> 14.22 ms per step - the divergence of
                          // a wavefront's four rows is NOT what the gather phases cost; off by default, kept as a switch
int g_slab_gu = 2;        // pgt_tune("slab_gu"): LDS reads in flight per gather of the four-task backward quad kernels (2 | 4)
int g_slab_quad = 1;      // pgt_tune("slab_quad"): 0 = C = 64 / 66 blocks on the pair-layout kernels (A/B)
// the quad-layout kernels take C = 64 (16-byte aligned segments) or C = 66 (8-byte aligned), at most 4 tasks per thread
static int slab_quad_kind(const SlabArgs& a) {
  if (!g_slab_quad || g_slab_pairs < 2) return 0;
  const int E = a.C - 64;
  if (E != 0 && E != 2) return 0;
  if ((int64_t)a.N * (16 + (E > 0)) > 4 * SLAB_THREADS) return 0;
  if (slab_q_lds_bytes(a.N, E, a.nnz_o, a.nnz_i) > 160 * 1024) return 0;
  if (E == 0) return (pgt_aligned(a.TS, 16) && a.seg_stride % 4 == 0) ? 1 : 0;
  return (pgt_aligned(a.TS, 8) && a.seg_stride % 2 == 0) ? 2 : 0;
}
template <bool BWD>
int launch_slab_q(const SlabArgs& a, int kind, pgt_stream_t stream) {
  const int tpr = kind == 1 ? 16 : 17;
  const int tpt = (int)pgt_cdiv((int64_t)a.N * tpr, SLAB_THREADS);
  const int nblk = a.n_samples < 256 ? a.n_samples : 256;
  dim3 grid((unsigned)nblk), block(SLAB_THREADS);
  // backward, four tasks per thread: five register blocks are live, so the gathers keep two reads in flight instead of
  // four (no spills: 77.5 us against 89 us at B = 1024, C = 64; pgt_tune("slab_gu", 4) for the A/B)
#define PGT_SLABQ(E_, G4_, T_)                                                                        \
  do {                                                                                                \
    if (BWD && T_ >= 4 && g_slab_gu != 4) PGT_LAUNCH((dconv_slab_bwd_q_kernel<E_, G4_, T_, 2>), grid, block, stream, a); \
    else if (BWD) PGT_LAUNCH((dconv_slab_bwd_q_kernel<E_, G4_, T_, 4>), grid, block, stream, a);      \
    else PGT_LAUNCH((dconv_slab_fwd_q_kernel<E_, G4_, T_>), grid, block, stream, a);                  \
  } while (0)
  if (kind == 1) { if (tpt <= 2) PGT_SLABQ(0, true, 2); else PGT_SLABQ(0, true, 4); }
  else { if (tpt <= 2) PGT_SLABQ(2, false, 2); else PGT_SLABQ(2, false, 4); }
#undef PGT_SLABQ
  return pgt_check_launch(BWD ? "pgt_dconv_stack_slab_bwd_f32" : "pgt_dconv_stack_slab_f32");
}

// ---------------------------------------------------------------------------------------------------------------------
// Column-split form (round 3).  The recursion acts on every column independently, so a sample's C columns are cut into
// `nsplit` windows of whole column pairs and a work item is (sample, window): the LDS blocks shrink to [N][cw], two or
// three workgroups fit a CU — one gathers out of LDS while another one's HBM loads and stores are in flight, which the
// single resident 1024-thread workgroup of the unsplit kernels cannot overlap (0.36 - 0.40 of 8 TB/s) — and a batch of
// 64 samples fills 128 - 192 CUs instead of 64.  Arithmetic per element is unchanged (same fmaf chain in slot order):
// bit-identical to the unsplit kernels and to pgt_spmm_csr_f32.
// Item order: w = ((sample / 8) * nsplit + part) * 8 + sample % 8, grid a multiple of 8 * nsplit: the windows of one
// sample run at the same time on ONE XCD (block b -> XCD b % 8), so the 128-byte lines they share meet in that XCD's L2,
// and a workgroup keeps its window for the whole launch (task tables are loop-invariant).
struct SlabWin { int c0, cw; };
__host__ __device__ inline SlabWin slab_window(int C, int nsplit, int part) {
  const int cp = C / 2, base = cp / nsplit, rem = cp % nsplit;
  SlabWin w;
  w.c0 = 2 * (part * base + (part < rem ? part : rem));
  w.cw = 2 * (base + (part < rem ? 1 : 0));
  return w;
}
static size_t slab_w_lds_bytes(int64_t N, int64_t C, int nsplit, int64_t nnz_o, int64_t nnz_i) {
  const int cwmax = slab_window((int)C, nsplit, 0).cw;
  const size_t blk = (((size_t)N * cwmax * 4) + 15) & ~(size_t)15;
  return 2 * blk + 2 * (size_t)(N + 1) * 4 + 2 * (size_t)(nnz_o + nnz_i) * 4;
}

template <int THREADS>
__device__ __forceinline__ void stage_csr_t(const SlabArgs& a, const SlabLds& s, int tid) {
  for (int i = tid; i <= a.N; i += THREADS) { s.rp_o[i] = a.rp_o[i]; s.rp_i[i] = a.rp_i[i]; }
  for (int i = tid; i < a.nnz_o; i += THREADS) { int2 t; t.x = a.col_o[i]; t.y = as_int(a.val_o[i]); s.cv_o[i] = t; }
  for (int i = tid; i < a.nnz_i; i += THREADS) { int2 t; t.x = a.col_i[i]; t.y = as_int(a.val_i[i]); s.cv_i[i] = t; }
}

// what every windowed kernel sets up: LDS carve-up with the widest window's pitch, this workgroup's window, its task table
template <int THREADS, int MAXT>
struct SlabTasks {
  // one register per task: (row << 8) | (window-relative column / 2); offsets, the half-task flag and liveness are
  // recomputed where they are used (a few integer operations against four registers per task held across the item loop)
  int rc[MAXT];
  int cw, c0, C, ntask, tid;
  __device__ __forceinline__ void init(const SlabArgs& a, int nsplit, int tid_) {
    const int part = ((int)blockIdx.x >> 3) % nsplit;
    const SlabWin w = slab_window(a.C, nsplit, part);
    cw = w.cw; c0 = w.c0; C = a.C; tid = tid_;
    const int CV = (w.cw + 3) / 4;
    ntask = a.N * CV;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * THREADS;
      const int ic = idx < ntask ? idx : ntask - 1;
      const int r = ic / CV, c = (ic - r * CV) * 4;
      rc[j] = (r << 8) | (c >> 1);
    }
  }
  __device__ __forceinline__ bool live(int j) const { return tid + j * THREADS < ntask; }
  __device__ __forceinline__ int row(int j) const { return rc[j] >> 8; }
  __device__ __forceinline__ int col(int j) const { return (rc[j] & 255) << 1; }
  __device__ __forceinline__ bool full(int j) const { return col(j) + 2 < cw; }
  __device__ __forceinline__ int goff(int j) const { return row(j) * C + c0 + col(j); }   // inside a sample's [N, C] block
  __device__ __forceinline__ int loff(int j) const { return row(j) * cw + col(j); }       // inside the [N, cw] LDS block
};
__device__ __forceinline__ SlabLds carve_w(char* base, const SlabArgs& a, int nsplit) {
  SlabLds s;
  const int cwmax = slab_window(a.C, nsplit, 0).cw;
  const size_t blk = (((size_t)a.N * cwmax * 4) + 15) & ~(size_t)15;
  s.bufA = reinterpret_cast<float*>(base);
  s.bufB = reinterpret_cast<float*>(base + blk);
  char* p = base + 2 * blk;
  s.cv_o = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_o * 8;
  s.cv_i = reinterpret_cast<int2*>(p); p += (size_t)a.nnz_i * 8;
  s.rp_o = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.rp_i = reinterpret_cast<int*>(p);
  return s;
}
// sample of item w (item order above); false past the batch
__device__ __forceinline__ bool slab_item_sample(const SlabArgs& a, int nsplit, int w, int& sample) {
  sample = (w / (8 * nsplit)) * 8 + (w & 7);
  return sample < a.n_samples;
}

template <int THREADS, int MAXT, int WPC>
__global__ __launch_bounds__(THREADS, (WPC * THREADS + 255) / 256) void dconv_slab_fwd_w_kernel(SlabArgs a, int nsplit, int n_items) {
  typedef P2 T;
  constexpr int GU = WPC >= 3 ? 2 : 4;     // LDS reads in flight per gather: more resident wavefronts, fewer registers each
  constexpr int LDS_BYTES = WPC == 1 ? 160 * 1024 : WPC == 2 ? 80 * 1024 : 54528;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve_w(smem, a, nsplit);
  const int tid = threadIdx.x;
  stage_csr_t<THREADS>(a, s, tid);
  SlabTasks<THREADS, MAXT> k;
  k.init(a, nsplit, tid);
  const int G = (int)gridDim.x;
  T t0n[MAXT];
  int sample;
  if (slab_item_sample(a, nsplit, (int)blockIdx.x, sample)) {
    const float* nb = a.TS + (int64_t)sample * a.N * a.C;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldT(nb + k.goff(j), k.full(j), t0n[j]);
  }
  for (int w = (int)blockIdx.x; w < n_items; w += G) {
    if (!slab_item_sample(a, nsplit, w, sample)) break;     // (the last group of eight may be ragged; nothing follows it)
    float* base = a.TS + (int64_t)sample * a.N * a.C;
    T t0[MAXT], i1[MAXT];
    PGT_LDS_BARRIER();  // CSR staged (first pass) / every lane done with the LDS blocks of the previous item
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      t0[j] = t0n[j];
      if (k.live(j)) stT(s.bufA + k.loff(j), k.full(j), t0[j]);
    }
    PGT_LDS_BARRIER();
    int sn;
    if (w + G < n_items && slab_item_sample(a, nsplit, w + G, sn)) {
      const float* nb = a.TS + (int64_t)sn * a.N * a.C;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) ldT(nb + k.goff(j), k.full(j), t0n[j]);
    }
    // hop 1: T1o = P_o T0 (into bufB, which nobody reads during this hop), T1i = P_i T0 (registers: bufA is being read)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        const T o1 = gather_row_p2<T, GU>(s.rp_o, s.cv_o, s.bufA, k.row(j), k.col(j), k.cw, k.full(j));
        i1[j] = gather_row_p2<T, GU>(s.rp_i, s.cv_i, s.bufA, k.row(j), k.col(j), k.cw, k.full(j));
        stT(base + 1 * a.seg_stride + k.goff(j), k.full(j), o1);
        stT(base + 2 * a.seg_stride + k.goff(j), k.full(j), i1[j]);
        if (a.K >= 3) stT(s.bufB + k.loff(j), k.full(j), o1);
      }
    }
    if (a.K < 3) continue;  // (uniform) K == 2: no second hop
    PGT_LDS_BARRIER();        // everyone has finished reading T0 out of bufA
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (k.live(j)) stT(s.bufA + k.loff(j), k.full(j), i1[j]);
    PGT_LDS_BARRIER();
    // hop 2: T2 = 2 P T1 - T0   (Tx_0 is never advanced in the reference, dcrnn.py:106)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        const T o2 = gather_row_p2<T, GU>(s.rp_o, s.cv_o, s.bufB, k.row(j), k.col(j), k.cw, k.full(j));
        const T i2 = gather_row_p2<T, GU>(s.rp_i, s.cv_i, s.bufA, k.row(j), k.col(j), k.cw, k.full(j));
        stT(base + 3 * a.seg_stride + k.goff(j), k.full(j), axpby(2.0f, o2, -1.0f, t0[j]));
        stT(base + 4 * a.seg_stride + k.goff(j), k.full(j), axpby(2.0f, i2, -1.0f, t0[j]));
      }
    }
  }
}

// backward, column-split; see dconv_slab_bwd_kernel for the recursion
template <int THREADS, int MAXT, int WPC>
__global__ __launch_bounds__(THREADS, (WPC * THREADS + 255) / 256) void dconv_slab_bwd_w_kernel(SlabArgs a, int nsplit, int n_items) {
  typedef P2 T;
  constexpr int GU = WPC >= 3 ? 2 : 4;     // LDS reads in flight per gather: more resident wavefronts, fewer registers each
  constexpr int LDS_BYTES = WPC == 1 ? 160 * 1024 : WPC == 2 ? 80 * 1024 : 54528;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve_w(smem, a, nsplit);
  const int tid = threadIdx.x;
  stage_csr_t<THREADS>(a, s, tid);
  SlabTasks<THREADS, MAXT> k;
  k.init(a, nsplit, tid);
  const int G = (int)gridDim.x;
  const int64_t lead = (a.K >= 3 ? 3 : 1) * a.seg_stride;
  T pa[MAXT], pb[MAXT];
  auto prefetch = [&](int bb) {
    const float* nb = a.TS + (int64_t)bb * a.N * a.C + lead;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      ldT(nb + k.goff(j), k.full(j), pa[j]);
      ldT(nb + a.seg_stride + k.goff(j), k.full(j), pb[j]);
    }
  };
  int sample;
  if (slab_item_sample(a, nsplit, (int)blockIdx.x, sample)) prefetch(sample);
  for (int w = (int)blockIdx.x; w < n_items; w += G) {
    if (!slab_item_sample(a, nsplit, w, sample)) break;
    float* base = a.TS + (int64_t)sample * a.N * a.C;
    PGT_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        stT(s.bufA + k.loff(j), k.full(j), pa[j]);
        stT(s.bufB + k.loff(j), k.full(j), pb[j]);
      }
    }
    PGT_LDS_BARRIER();
    int sn;
    if (w + G < n_items && slab_item_sample(a, nsplit, w + G, sn)) prefetch(sn);
    if (a.K >= 3) {
      T g1o[MAXT], g1i[MAXT];
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        ldT(base + 1 * a.seg_stride + k.goff(j), k.full(j), g1o[j]);
        ldT(base + 2 * a.seg_stride + k.goff(j), k.full(j), g1i[j]);
      }
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (k.live(j)) {
          g1o[j] = axpby(2.0f, gather_row_p2<T, GU>(s.rp_o, s.cv_o, s.bufA, k.row(j), k.col(j), k.cw, k.full(j)), 1.0f, g1o[j]);
          g1i[j] = axpby(2.0f, gather_row_p2<T, GU>(s.rp_i, s.cv_i, s.bufB, k.row(j), k.col(j), k.cw, k.full(j)), 1.0f, g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (k.live(j)) {
          stT(s.bufA + k.loff(j), k.full(j), g1o[j]);
          stT(s.bufB + k.loff(j), k.full(j), g1i[j]);
        }
      }
      PGT_LDS_BARRIER();
    }
    T g0[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) ldT(base + k.goff(j), k.full(j), g0[j]);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (k.live(j)) {
        const T po = gather_row_p2<T, GU>(s.rp_o, s.cv_o, s.bufA, k.row(j), k.col(j), k.cw, k.full(j));
        const T pi = gather_row_p2<T, GU>(s.rp_i, s.cv_i, s.bufB, k.row(j), k.col(j), k.cw, k.full(j));
        T g = g0[j];
        if (a.K >= 3 && !a.folded) {  // G0 -= G2o + G2i (re-read: the unfolded form is the rare one)
          T g2o, g2i;
          ldT(base + 3 * a.seg_stride + k.goff(j), k.full(j), g2o);
          ldT(base + 4 * a.seg_stride + k.goff(j), k.full(j), g2i);
          g = add3(g, axpby(-1.0f, g2o, 0.0f, g2o), axpby(-1.0f, g2i, 0.0f, g2i));
        }
        stT(base + k.goff(j), k.full(j), add3(g, po, pi));
      }
    }
  }
}

int g_slab_split = 1;     // pgt_tune("slab_split"): 1 = column-split kernels where they apply (auto), 0 = never, n >= 2 = n windows
int g_slab_threads = 0;   // pgt_tune("slab_threads"): 0 = planned, else the workgroup size of the column-split kernels (A/B)
int g_slab_wpc = 0;       // pgt_tune("slab_wpc"): 0 = as many workgroups per CU as the LDS need allows (<= 3), else at most this

struct SlabPlan { int nsplit, threads, maxt, wpc; };
// Workgroup shapes the windowed kernels are instantiated for, per workgroups-per-CU class: the ones whose register need
// (82 - 92 with two tasks per thread, 106 - 122 with three: scripts/kernel_resources.py) fits the class's wavefronts per
// SIMD without spilling.
static const int kSlabShapes[][3] = {   // {wpc, threads, maxt}
    {1, 1024, 2}, {1, 1024, 3}, {2, 512, 2}, {2, 640, 2}, {2, 512, 3}, {3, 512, 2}};
constexpr int kSlabNumShapes = (int)(sizeof(kSlabShapes) / sizeof(kSlabShapes[0]));
static size_t slab_class_bytes(int wpc) { return wpc == 1 ? 160 * 1024 : wpc == 2 ? 80 * 1024 : 54528; }
constexpr int SLAB_MAX_SPLIT = 8;
static bool slab_w_applies(int64_t C) { return C % 2 == 0 && C >= 8; }

// the smallest shape of class `wpc` (optionally of `threads` threads) with a slot for every task; -1 = none
static int slab_pick_shape(int wpc, int64_t ntask, int threads) {
  int best = -1;
  for (int i = 0; i < kSlabNumShapes; ++i) {
    if (kSlabShapes[i][0] != wpc || (threads > 0 && kSlabShapes[i][1] != threads)) continue;
    const int64_t slots = (int64_t)kSlabShapes[i][1] * kSlabShapes[i][2];
    if (slots < ntask) continue;
    if (best < 0 || slots < (int64_t)kSlabShapes[best][1] * kSlabShapes[best][2]) best = i;
  }
  return best;
}
// plan for `nsplit` windows: the densest class (most workgroups per CU, at most wpc_cap) that holds the window's blocks
// and has a shape for its tasks
static bool slab_plan_for(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i, int nsplit, int wpc_cap, int threads, SlabPlan* p) {
  if (nsplit < 1 || nsplit > C / 4) return false;                   // at least two column pairs (one full task) per window
  const size_t need = slab_w_lds_bytes(N, C, nsplit, nnz_o, nnz_i);
  const int64_t ntask = N * ((slab_window((int)C, nsplit, 0).cw + 3) / 4);
  for (int wpc = wpc_cap < 3 ? wpc_cap : 3; wpc >= 1; --wpc) {
    if (need > slab_class_bytes(wpc)) continue;
    int i = slab_pick_shape(wpc, ntask, threads);
    if (i < 0 && threads > 0) i = slab_pick_shape(wpc, ntask, 0);   // a forced size nothing fits: as planned
    if (i < 0) continue;
    if (p) { p->nsplit = nsplit; p->wpc = wpc; p->threads = kSlabShapes[i][1]; p->maxt = kSlabShapes[i][2]; }
    return true;
  }
  return false;
}
// the fewest windows that put at least two workgroups on a CU; failing that, the fewest that run at all
static bool slab_plan_auto(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i, int wpc_cap, int threads, SlabPlan* p) {
  SlabPlan q;
  for (int n = 1; n <= SLAB_MAX_SPLIT; ++n)
    if (slab_plan_for(N, C, nnz_o, nnz_i, n, wpc_cap, threads, &q) && (q.wpc >= 2 || wpc_cap < 2)) { if (p) *p = q; return true; }
  for (int n = 1; n <= SLAB_MAX_SPLIT; ++n)
    if (slab_plan_for(N, C, nnz_o, nnz_i, n, wpc_cap, threads, p)) return true;
  return false;
}

// whole-sample kernels: both [N, C] blocks and the operators in one workgroup's LDS
static bool slab_whole_ok(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i) {
  const int V = (C % 2 == 0) ? 2 : 1;
  return N * (C / V) <= (int64_t)MAXT_CAP * SLAB_THREADS &&        // (the narrowest element the launch may pick)
         slab_lds_bytes(N, C, nnz_o, nnz_i) <= 160 * 1024;
}

// Measured on MI355X (scripts/slab_probe.py, METR-LA shape): with at least a sample per CU the whole-sample kernels win
// (B = 1024, C = 66 forward: 90 us against 103 - 123 us for every windowed shape: a window is an 88-byte piece of each
// 264-byte row, its 128-byte lines are fetched by two or three workgroups, and two resident workgroups did NOT overlap
// their phases usefully); with fewer samples than CUs the windows are what fills the chip (B = 64: 20.7 -> 12.1 us with
// four windows on 1024-thread workgroups, 13.2 with three, 14.6 with eight).  So: as many windows as idle CUs per sample.
static bool slab_plan(const SlabArgs& a, SlabPlan* p) {
  if (g_slab_split == 0 || g_slab_pairs < 2 || !slab_w_applies(a.C) || !pgt_aligned(a.TS, 8) || a.seg_stride % 2) return false;
  const int cap = g_slab_wpc > 0 ? g_slab_wpc : 3;
  if (g_slab_split >= 2) {
    if (slab_plan_for(a.N, a.C, a.nnz_o, a.nnz_i, g_slab_split, cap, g_slab_threads, p)) return true;
    return slab_plan_auto(a.N, a.C, a.nnz_o, a.nnz_i, cap, g_slab_threads, p);
  }
  if (!slab_whole_ok(a.N, a.C, a.nnz_o, a.nnz_i))                    // only fits column by column
    return slab_plan_auto(a.N, a.C, a.nnz_o, a.nnz_i, cap, g_slab_threads, p);
  if (2 * (int64_t)a.n_samples > SLAB_CUS) return false;
  int want = SLAB_CUS / (a.n_samples > 0 ? a.n_samples : 1);
  if (want > SLAB_MAX_SPLIT) want = SLAB_MAX_SPLIT;
  for (int n = want; n >= 2; --n)                                     // one workgroup per CU: the 1024-thread shapes first
    if (slab_plan_for(a.N, a.C, a.nnz_o, a.nnz_i, n, g_slab_wpc > 0 ? cap : 1, g_slab_threads, p) ||
        slab_plan_for(a.N, a.C, a.nnz_o, a.nnz_i, n, cap, g_slab_threads, p)) return true;
  return false;
}

template <bool BWD>
int launch_slab_w(const SlabArgs& a, const SlabPlan& p, pgt_stream_t stream) {
  const int groups = (int)pgt_cdiv(a.n_samples, 8);
  const int unit = 8 * p.nsplit;
  const int n_items = groups * unit;
  int grid_n = (p.wpc * SLAB_CUS / unit) * unit;             // whole item groups: a workgroup keeps its window
  if (grid_n < unit) grid_n = unit;
  if (grid_n > n_items) grid_n = n_items;
  dim3 grid((unsigned)grid_n), block((unsigned)p.threads);
#define PGT_SLABW(W_, T_, M_)                                                                                   \
  if (p.wpc == W_ && p.threads == T_ && p.maxt == M_) {                                                         \
    if (BWD) PGT_LAUNCH((dconv_slab_bwd_w_kernel<T_, M_, W_>), grid, block, stream, a, p.nsplit, n_items);      \
    else PGT_LAUNCH((dconv_slab_fwd_w_kernel<T_, M_, W_>), grid, block, stream, a, p.nsplit, n_items);          \
    return pgt_check_launch(BWD ? "pgt_dconv_stack_slab_bwd_f32" : "pgt_dconv_stack_slab_f32");                 \
  }
  PGT_SLABW(1, 1024, 2) PGT_SLABW(1, 1024, 3) PGT_SLABW(2, 512, 2) PGT_SLABW(2, 640, 2) PGT_SLABW(2, 512, 3) PGT_SLABW(3, 512, 2)
#undef PGT_SLABW
  pgt_set_error("pgt_dconv_stack_slab: no kernel for the planned shape");
  return PGT_ERR_INVALID;
}


int slab_supported(int64_t N, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i, size_t* bytes) {
  if (N <= 0 || C <= 0 || K < 2 || K > 3) return 0;
  if (nnz_o < 0 || nnz_i < 0 || nnz_o > (1 << 24) || nnz_i > (1 << 24)) return 0;
  if (bytes) *bytes = slab_lds_bytes(N, C, nnz_o, nnz_i);
  if (slab_whole_ok(N, C, nnz_o, nnz_i)) return 1;
  // too large for one workgroup's LDS as a whole: the column-split kernels may still take it window by window
  // (PeMS-BAY, 325 nodes, at hidden 64: two windows of 34 / 32 columns)
  return g_slab_split != 0 && g_slab_pairs >= 2 && slab_w_applies(C) && slab_plan_auto(N, C, nnz_o, nnz_i, 3, 0, nullptr);
}

template <bool BWD>
int launch_slab(const SlabArgs& a, size_t need, pgt_stream_t stream) {
  SlabPlan plan;
  if (slab_plan(a, &plan)) return launch_slab_w<BWD>(a, plan, stream);
  if (!slab_whole_ok(a.N, a.C, a.nnz_o, a.nnz_i)) {
    pgt_set_error("pgt_dconv_stack_slab: the block only fits column by column, which needs 8-byte aligned even-width segments");
    return PGT_ERR_INVALID;
  }
  if (const int kind = slab_quad_kind(a)) return launch_slab_q<BWD>(a, kind, stream);
  int V = (a.C % 2 == 0 && pgt_aligned(a.TS, 8) && a.seg_stride % 2 == 0) ? 2 : 1;
  // two column pairs per lane (half the slot reads) while its register arrays fit: at most 4 tasks per thread
  if (V == 2 && g_slab_pairs >= 2 && a.C >= 8 && (int64_t)a.N * pgt_cdiv(a.C, 4) <= 4 * SLAB_THREADS) V = 4;
  const int64_t ntask = (int64_t)a.N * pgt_cdiv(a.C, V);
  if (ntask > (int64_t)MAXT_CAP * SLAB_THREADS) {
    pgt_set_error("pgt_dconv_stack_slab: block too large for the scalar path");
    return PGT_ERR_INVALID;
  }
  const int tpt = (int)pgt_cdiv(ntask, SLAB_THREADS);
  const int nblk = a.n_samples < 256 ? a.n_samples : 256;
  dim3 grid((unsigned)nblk), block(SLAB_THREADS);
#define PGT_SLAB_K(V_, L_, T_)                                                                        \
  do {                                                                                                \
    if (BWD) PGT_LAUNCH((dconv_slab_bwd_kernel<V_, L_, T_>), grid, block, stream, a);                 \
    else PGT_LAUNCH((dconv_slab_fwd_kernel<V_, L_, T_>), grid, block, stream, a);                     \
  } while (0)
#define PGT_SLAB_P2(L_, T_)                                                                           \
  do {                                                                                                \
    if (BWD) PGT_LAUNCH((dconv_slab_bwd_p2_kernel<4, L_, T_>), grid, block, stream, a);               \
    else PGT_LAUNCH((dconv_slab_fwd_p2_kernel<4, L_, T_>), grid, block, stream, a);                   \
  } while (0)
#define PGT_SLAB_T(V_, L_)                                                                            \
  do {                                                                                                \
    if (tpt <= 2) PGT_SLAB_K(V_, L_, 2);                                                              \
    else if (tpt <= 4) PGT_SLAB_K(V_, L_, 4);                                                         \
    else if (tpt <= 7) PGT_SLAB_K(V_, L_, 7);                                                         \
    else PGT_SLAB_K(V_, L_, 8);                                                                       \
  } while (0)
  if (V == 4) {                                        // at most 4 tasks per thread (checked above)
    if (need <= 64 * 1024) { if (tpt <= 2) PGT_SLAB_P2(64 * 1024, 2); else PGT_SLAB_P2(64 * 1024, 4); }
    else { if (tpt <= 2) PGT_SLAB_P2(160 * 1024, 2); else PGT_SLAB_P2(160 * 1024, 4); }
  } else if (need <= 64 * 1024) {
    if (V == 2) PGT_SLAB_T(2, 64 * 1024); else PGT_SLAB_T(1, 64 * 1024);
  } else {
    if (V == 2) PGT_SLAB_T(2, 160 * 1024); else PGT_SLAB_T(1, 160 * 1024);
  }
#undef PGT_SLAB_T
#undef PGT_SLAB_P2
#undef PGT_SLAB_K
  return pgt_check_launch(BWD ? "pgt_dconv_stack_slab_bwd_f32" : "pgt_dconv_stack_slab_f32");
}

int slab_entry(bool bwd, const pgt_csr* o, const pgt_csr* i, int64_t nnz_o, int64_t nnz_i, int64_t N,
               int64_t n_samples, int64_t C, int64_t K, float* TS, int64_t seg_stride, int folded,
               pgt_stream_t stream) {
  const char* who = bwd ? "pgt_dconv_stack_slab_bwd_f32" : "pgt_dconv_stack_slab_f32";
  PGT_REQUIRE(N >= 0 && n_samples >= 0 && C >= 0, "%s: negative size", who);
  if (N == 0 || n_samples == 0 || C == 0 || K < 2) return PGT_OK;
  PGT_REQUIRE(o && i && TS, "%s: null pointer", who);
  PGT_REQUIRE(o->rowptr && i->rowptr && (nnz_o == 0 || (o->col && o->val)) && (nnz_i == 0 || (i->col && i->val)),
              "%s: null operator", who);
  size_t need = 0;
  PGT_REQUIRE(slab_supported(N, C, K, nnz_o, nnz_i, &need),
              "%s: shape not supported by the LDS-resident schedule (N=%lld C=%lld K=%lld); use pgt_spmm_csr_f32", who,
              (long long)N, (long long)C, (long long)K);
  PGT_REQUIRE(n_samples < ((int64_t)1 << 31) && n_samples * N * C < ((int64_t)1 << 40), "%s: batch too large", who);
  SlabArgs a{o->rowptr, o->col, o->val, i->rowptr, i->col, i->val, (int)N, (int)C, (int)K, (int)nnz_o, (int)nnz_i,
             (int)n_samples, TS, seg_stride, folded, g_slab_sort};
  return bwd ? launch_slab<true>(a, need, stream) : launch_slab<false>(a, need, stream);
}

}  // namespace

void pgt_slab_set_pairs(int v) { g_slab_pairs = v; }
void pgt_slab_set_split(int v) { g_slab_split = v; }
void pgt_slab_set_threads(int v) { g_slab_threads = v; }
void pgt_slab_set_wpc(int v) { g_slab_wpc = v; }
void pgt_slab_set_quad(int v) { g_slab_quad = v; }
void pgt_slab_set_gu(int v) { g_slab_gu = v; }
void pgt_slab_set_sort(int v) { g_slab_sort = v; }

extern "C" int pgt_dconv_stack_slab_fits(int64_t N, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i) {
  return slab_supported(N, C, K, nnz_o, nnz_i, nullptr);
}

extern "C" int pgt_dconv_stack_slab_plan(int64_t N, int64_t n_samples, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i,
                                         int32_t* plan) {
  PGT_REQUIRE(plan != nullptr, "pgt_dconv_stack_slab_plan: null pointer");
  plan[0] = plan[1] = plan[2] = plan[3] = 0;
  if (!slab_supported(N, C, K, nnz_o, nnz_i, nullptr)) return PGT_OK;
  plan[0] = 1;                                            // whole-sample kernels: one 1024-thread workgroup per sample
  plan[1] = 1; plan[2] = SLAB_THREADS; plan[3] = 0;
  SlabArgs a{};
  a.N = (int)N; a.C = (int)C; a.K = (int)K; a.nnz_o = (int)nnz_o; a.nnz_i = (int)nnz_i;
  a.n_samples = (int)(n_samples < ((int64_t)1 << 30) ? n_samples : ((int64_t)1 << 30));
  a.TS = nullptr; a.seg_stride = 0;                       // (alignment of the caller's buffers is checked at launch)
  SlabPlan p;
  if (slab_plan(a, &p)) { plan[0] = p.nsplit; plan[1] = p.wpc; plan[2] = p.threads; plan[3] = p.maxt; }
  return PGT_OK;
}

extern "C" int pgt_dconv_stack_slab_f32(const pgt_csr* fwd_o, const pgt_csr* fwd_i, int64_t nnz_o, int64_t nnz_i,
                                        int64_t N, int64_t n_samples, int64_t C, int64_t K, float* TS,
                                        int64_t seg_stride, pgt_stream_t stream) {
  return slab_entry(false, fwd_o, fwd_i, nnz_o, nnz_i, N, n_samples, C, K, TS, seg_stride, 0, stream);
}

extern "C" int pgt_dconv_stack_slab_bwd_f32(const pgt_csr* bwd_o, const pgt_csr* bwd_i, int64_t nnz_o,
                                            int64_t nnz_i, int64_t N, int64_t n_samples, int64_t C, int64_t K,
                                            float* G, int64_t seg_stride, int folded, pgt_stream_t stream) {
  return slab_entry(true, bwd_o, bwd_i, nnz_o, nnz_i, N, n_samples, C, K, G, seg_stride, folded, stream);
}
