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

namespace {

struct SlabArgs {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;
  int N, C, K, nnz_o, nnz_i;
  int n_samples;
  float* TS;            // segment s of sample b starts at TS + s*seg_stride + b*N*C
  int64_t seg_stride;
  int folded;           // backward only: the "- Tx_0" adjoint was folded into the weights (ops.fold_backward_weight)
};

constexpr int SLAB_THREADS = 1024;
constexpr int MAXT = 8;  // tasks (row, V-float column group) per thread: N * C / V <= MAXT * 1024

template <int V>
struct VecT;
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

// LDS carve-up (bytes): two [N*C] float blocks, then the two operators
struct SlabLds {
  float* bufA; float* bufB;
  int* rp_o; int* rp_i; int* col_o; int* col_i; float* val_o; float* val_i;
};
__device__ __forceinline__ SlabLds carve(char* base, const SlabArgs& a) {
  SlabLds s;
  const size_t blk = (((size_t)a.N * a.C * 4) + 15) & ~(size_t)15;
  s.bufA = reinterpret_cast<float*>(base);
  s.bufB = reinterpret_cast<float*>(base + blk);
  char* p = base + 2 * blk;
  s.rp_o = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.rp_i = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.col_o = reinterpret_cast<int*>(p); p += (size_t)a.nnz_o * 4;
  s.col_i = reinterpret_cast<int*>(p); p += (size_t)a.nnz_i * 4;
  s.val_o = reinterpret_cast<float*>(p); p += (size_t)a.nnz_o * 4;
  s.val_i = reinterpret_cast<float*>(p);
  return s;
}

static size_t slab_lds_bytes(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i) {
  const size_t blk = (((size_t)N * C * 4) + 15) & ~(size_t)15;
  return 2 * blk + 2 * (size_t)(N + 1) * 4 + 2 * (size_t)(nnz_o + nnz_i) * 4;
}

template <typename T>
__device__ __forceinline__ T gather_row(const int* __restrict__ rp, const int* __restrict__ col,
                                        const float* __restrict__ val, const float* __restrict__ buf, int r, int c,
                                        int C) {
  T acc = zero2(T());
  const int e = rp[r + 1];
  for (int q = rp[r]; q < e; ++q) acc = fma2(val[q], *reinterpret_cast<const T*>(buf + col[q] * C + c), acc);
  return acc;
}

__device__ __forceinline__ void stage_csr(const SlabArgs& a, const SlabLds& s, int tid) {
  for (int i = tid; i <= a.N; i += SLAB_THREADS) { s.rp_o[i] = a.rp_o[i]; s.rp_i[i] = a.rp_i[i]; }
  for (int i = tid; i < a.nnz_o; i += SLAB_THREADS) { s.col_o[i] = a.col_o[i]; s.val_o[i] = a.val_o[i]; }
  for (int i = tid; i < a.nnz_i; i += SLAB_THREADS) { s.col_i[i] = a.col_i[i]; s.val_i[i] = a.val_i[i]; }
}

// forward: segments [T0 | T1o T1i | T2o T2i]; K = 2 or 3
template <int V, int LDS_BYTES>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_fwd_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = a.C / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);

  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    T t0[MAXT], o1[MAXT], i1[MAXT];
    __syncthreads();  // CSR staged (first pass) / every lane done with the LDS blocks of the previous sample
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        t0[j] = reinterpret_cast<const T*>(base)[idx];
        reinterpret_cast<T*>(s.bufA)[idx] = t0[j];
      }
    }
    __syncthreads();
    // hop 1: T1o = P_o T0, T1i = P_i T0
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        o1[j] = gather_row<T>(s.rp_o, s.col_o, s.val_o, s.bufA, r, c, a.C);
        i1[j] = gather_row<T>(s.rp_i, s.col_i, s.val_i, s.bufA, r, c, a.C);
        reinterpret_cast<T*>(base + 1 * a.seg_stride)[idx] = o1[j];
        reinterpret_cast<T*>(base + 2 * a.seg_stride)[idx] = i1[j];
      }
    }
    if (a.K < 3) continue;  // (uniform) K == 2: no second hop
    __syncthreads();        // everyone has finished reading T0 out of bufA
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        reinterpret_cast<T*>(s.bufA)[idx] = o1[j];
        reinterpret_cast<T*>(s.bufB)[idx] = i1[j];
      }
    }
    __syncthreads();
    // hop 2: T2 = 2 P T1 - T0   (Tx_0 is never advanced in the reference, dcrnn.py:106)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T o2 = gather_row<T>(s.rp_o, s.col_o, s.val_o, s.bufA, r, c, a.C);
        const T i2 = gather_row<T>(s.rp_i, s.col_i, s.val_i, s.bufB, r, c, a.C);
        reinterpret_cast<T*>(base + 3 * a.seg_stride)[idx] = axpby(2.0f, o2, -1.0f, t0[j]);
        reinterpret_cast<T*>(base + 4 * a.seg_stride)[idx] = axpby(2.0f, i2, -1.0f, t0[j]);
      }
    }
  }
}

// backward on the TRANSPOSED operators (a.rp_o = bwd_o ...): segments [G0 | G1o G1i | G2o G2i] -> G0 (in place)
//   K == 3:  G1d += 2 P_d^T G2d ;  G0 += P_o^T G1o + P_i^T G1i  [ - G2o - G2i unless folded ]
//   K == 2:  G0 += P_o^T G1o + P_i^T G1i
template <int V, int LDS_BYTES>
__global__ __launch_bounds__(SLAB_THREADS) void dconv_slab_bwd_kernel(SlabArgs a) {
  typedef typename VecT<V>::type T;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const SlabLds s = carve(smem, a);
  const int tid = threadIdx.x;
  const int CV = a.C / V;
  const int ntask = a.N * CV;
  stage_csr(a, s, tid);

  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* base = a.TS + (int64_t)b * a.N * a.C;
    T g1o[MAXT], g1i[MAXT];
    __syncthreads();
    if (a.K >= 3) {
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        if (idx < ntask) {
          reinterpret_cast<T*>(s.bufA)[idx] = reinterpret_cast<const T*>(base + 3 * a.seg_stride)[idx];
          reinterpret_cast<T*>(s.bufB)[idx] = reinterpret_cast<const T*>(base + 4 * a.seg_stride)[idx];
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        if (idx < ntask) {
          const int r = idx / CV, c = (idx - r * CV) * V;
          const T po = gather_row<T>(s.rp_o, s.col_o, s.val_o, s.bufA, r, c, a.C);
          const T pi = gather_row<T>(s.rp_i, s.col_i, s.val_i, s.bufB, r, c, a.C);
          g1o[j] = axpby(2.0f, po, 1.0f, reinterpret_cast<const T*>(base + 1 * a.seg_stride)[idx]);
          g1i[j] = axpby(2.0f, pi, 1.0f, reinterpret_cast<const T*>(base + 2 * a.seg_stride)[idx]);
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        const int idx = tid + j * SLAB_THREADS;
        if (idx < ntask) {
          g1o[j] = reinterpret_cast<const T*>(base + 1 * a.seg_stride)[idx];
          g1i[j] = reinterpret_cast<const T*>(base + 2 * a.seg_stride)[idx];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        reinterpret_cast<T*>(s.bufA)[idx] = g1o[j];
        reinterpret_cast<T*>(s.bufB)[idx] = g1i[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int idx = tid + j * SLAB_THREADS;
      if (idx < ntask) {
        const int r = idx / CV, c = (idx - r * CV) * V;
        const T po = gather_row<T>(s.rp_o, s.col_o, s.val_o, s.bufA, r, c, a.C);
        const T pi = gather_row<T>(s.rp_i, s.col_i, s.val_i, s.bufB, r, c, a.C);
        T g0 = reinterpret_cast<const T*>(base)[idx];
        if (a.K >= 3 && !a.folded) {  // G0 -= G2o + G2i (re-read: the unfolded form is the rare one)
          const T g2o = reinterpret_cast<const T*>(base + 3 * a.seg_stride)[idx];
          const T g2i = reinterpret_cast<const T*>(base + 4 * a.seg_stride)[idx];
          g0 = add3(g0, axpby(-1.0f, g2o, 0.0f, g2o), axpby(-1.0f, g2i, 0.0f, g2i));
        }
        reinterpret_cast<T*>(base)[idx] = add3(g0, po, pi);
      }
    }
  }
}

int slab_supported(int64_t N, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i, size_t* bytes) {
  if (N <= 0 || C <= 0 || K < 2 || K > 3) return 0;
  const int V = (C % 2 == 0) ? 2 : 1;
  if (N * (C / V) > (int64_t)MAXT * SLAB_THREADS) return 0;
  if (nnz_o < 0 || nnz_i < 0 || nnz_o > (1 << 24) || nnz_i > (1 << 24)) return 0;
  const size_t need = slab_lds_bytes(N, C, nnz_o, nnz_i);
  if (bytes) *bytes = need;
  return need <= 160 * 1024;
}

template <bool BWD>
int launch_slab(const SlabArgs& a, size_t need, pgt_stream_t stream) {
  const int V = (a.C % 2 == 0 && pgt_aligned(a.TS, 8) && a.seg_stride % 2 == 0) ? 2 : 1;
  if (V == 1 && (int64_t)a.N * a.C > (int64_t)MAXT * SLAB_THREADS) {
    pgt_set_error("pgt_dconv_stack_slab: block too large for the scalar path");
    return PGT_ERR_INVALID;
  }
  const int nblk = a.n_samples < 256 ? a.n_samples : 256;
  dim3 grid((unsigned)nblk), block(SLAB_THREADS);
#define PGT_SLAB_GO(V_, L_)                                                                          \
  do {                                                                                               \
    if (BWD) PGT_LAUNCH((dconv_slab_bwd_kernel<V_, L_>), grid, block, stream, a);                    \
    else PGT_LAUNCH((dconv_slab_fwd_kernel<V_, L_>), grid, block, stream, a);                        \
  } while (0)
  if (need <= 32 * 1024) { if (V == 2) PGT_SLAB_GO(2, 32 * 1024); else PGT_SLAB_GO(1, 32 * 1024); }
  else if (need <= 80 * 1024) { if (V == 2) PGT_SLAB_GO(2, 80 * 1024); else PGT_SLAB_GO(1, 80 * 1024); }
  else { if (V == 2) PGT_SLAB_GO(2, 160 * 1024); else PGT_SLAB_GO(1, 160 * 1024); }
#undef PGT_SLAB_GO
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
             (int)n_samples, TS, seg_stride, folded};
  return bwd ? launch_slab<true>(a, need, stream) : launch_slab<false>(a, need, stream);
}

}  // namespace

extern "C" int pgt_dconv_stack_slab_fits(int64_t N, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i) {
  return slab_supported(N, C, K, nnz_o, nnz_i, nullptr);
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
