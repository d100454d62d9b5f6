// LAB RECORD (not built into the library): the per-sample fused diffusion convolution + gate kernel of DESIGN.md section 9
// ("stack in LDS -> MFMA"), built, parity-green on the CPU test double and on the GPU (commit "Fused diffusion
// convolution + gate kernel", tests/test_dcrnn.py at that commit) and MEASURED at the benchmark shape (B = 1024,
// N = 207, C = 66, hidden 64; scripts at that commit: scripts/fused_probe.py):
//
//     two-launch path (stack kernel + gate-fused GEMM)     zr: 82 + 253 = 335 us      h: 82 + 154 = 236 us
//     this kernel                                          zr: 492 us                 h: 410 us
//     ... with the MFMAs removed 306 us, with the gathers removed 375 us, with the stack stores removed 435 us,
//     with all three removed 160 us (T_0 loads, nine barriers, epilogue)
//
// The parts ADD: one resident workgroup per CU (two [N, C + 4] blocks + both operators = 142 KB of LDS leave no room for
// a second one) runs load -> MFMA -> gather -> MFMA -> ... strictly in sequence, so MFMA (46 us per sample, 31 us ideal),
// LDS gathers (29 us), stack stores (14 us) and the skeleton (40 us) never overlap, while the two-launch path keeps four
// GEMM workgroups per CU in different phases.  Even with ideal MFMA issue and the stack kernel's gather speed the sum
// is ~80 us per sample = 320 us per launch: parity at best.  The step stays on the two-launch path.
//
// Fused diffusion convolution + gate for small graphs (K = 3): the five diffusion terms of one DConv never leave the CU
// between the aggregation and the feature transform.
//
//   out = sum_s T_s W_s + b,   T_0 = [X_t, H],  T_1^{o,i} = P_{o,i} T_0,  T_2^{o,i} = 2 P_{o,i} T_1^{o,i} - T_0   (dcrnn.py:79-111)
//   zr:  ZR = sigmoid(out) [M, 2 O];  XHR[:, Fin:] = H * R                                          (dcrnn.py:172-186)
//   h :  HT = tanh(out);  H' = Z * H + (1 - Z) * HT                                                 (dcrnn.py:188-192)
//
// Today's path is two launches per convolution: the LDS-resident stack kernel writes the four diffused terms to HBM
// (needed again by the weight gradient) and the GEMM re-reads all five.  Here one 512-thread workgroup
// owns a sample: T_0 (overwritten in place by each direction's second hop, then fetched again) and the first-hop term
// live in LDS next to both CSR operators (packed (col, val) slots), and
// every term is multiplied by its weight block straight out of LDS with v_mfma_f32_32x32x2_f32 while it is still
// there.  The terms are still stored once (the weight-gradient GEMM of the backward pass reads them); what disappears
// is the GEMM's re-read of 5 x [M, C] and the second launch.
//
// Layout: LDS rows are CP = roundup(C, 4) + 2 floats apart (C = 66 -> 70): rows start on different bank pairs, so the
// ds_read_b64 of 32 lanes reading 32 consecutive rows at one column pair is conflict-free, and a column pair of a row stays
// 8-byte aligned for the gather.  Columns C .. CP-1 are zero: the K extent is padded to a multiple of 4.
// MFMA operands: lanes 0-31 read the float2 (A[i][4g], A[i][4g+1]) of row i, lanes 32-63 read (A[i][4g+2], A[i][4g+3]);
// the first MFMA of a group multiplies the .x halves (k = 4g and 4g+2), the second the .y halves (k = 4g+1, 4g+3); the
// weight rows are fetched to match, coalesced along the output column (L2-resident: every workgroup reads the same 169 KB).
// Output tiles: 32 x 32; 512 threads = 8 wavefronts, two per SIMD; wave w owns column tile w % NCT and row tiles
// w / NCT + j * (8 / NCT): at N = 207, 2 O = 128 the seven row tiles x four column tiles give every SIMD seven tiles.
#include <string.h>

#include "pgt_common.h"

namespace {

struct FusedArgs {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;
  int N, C, Fin, O, nnz_o, nnz_i, n_samples;
  float* TS; int64_t seg_stride;     // segment s of sample b: TS + s * seg_stride + b * N * C; segment 0 holds T_0
  const float* W;                    // [5 C, NOUT] row-major, NOUT = 2 O (zr) or O (h)
  const float* bias;                 // [NOUT] or null
  float* ZR;                         // zr: out [M, 2 O];  h: in (Z = ZR[:, :O])
  float* XHR; int64_t ldxhr;         // zr: segment 0 of the candidate's stack; columns Fin.. receive H * R
  float* HT;                         // h: tanh output [M, O]
  const float* Hp; int64_t ldhp;     // h: previous hidden state
  float* Hout; int64_t ldo;          // h: new hidden state
  float* Hnext; int64_t ldn;         // h: second copy (next step's stack slot) or null
  int dbg;                           // lab switches (pgt_tune("dconv_fused_dbg")): 1 = no MFMA, 2 = no gathers, 4 = no stack stores
};

constexpr int FUSED_THREADS = 512;   // 8 wavefronts, two per SIMD: 256 registers per lane (four accumulator tiles + the held second hop)
constexpr int FUSED_LDS = 163840;
       // column pairs per thread held across the barrier of the second hop

__device__ __forceinline__ float f_as(int v) { union { int i; float f; } u; u.i = v; return u.f; }
__device__ __forceinline__ int i_as(float v) { union { int i; float f; } u; u.f = v; return u.i; }
__device__ __forceinline__ int2 mk_i2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }

// EPI 1 = update / reset gates (NOUT = 2 O), 2 = candidate + blend (NOUT = O).  NCT = NOUT / 32 column tiles.
template <int EPI, int NCT>
__global__ __launch_bounds__(FUSED_THREADS) void dconv_fused_kernel(FusedArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[FUSED_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, C = a.C, Kp = (C + 3) & ~3, CP = Kp + 2, NP = C >> 1;     // NP column pairs per row
  const int NOUT = NCT * 32;
  float* sA = reinterpret_cast<float*>(smem);
  float* sB = sA + N * CP;
  int2* cv_o = reinterpret_cast<int2*>(sB + N * CP);
  int2* cv_i = cv_o + a.nnz_o;
  int* rp_o = reinterpret_cast<int*>(cv_i + a.nnz_i);
  int* rp_i = rp_o + (N + 1);
  // operators -> LDS (once per workgroup), padding columns -> 0
  for (int q = tid; q < a.nnz_o; q += FUSED_THREADS) cv_o[q] = mk_i2(a.col_o[q], i_as(a.val_o[q]));
  for (int q = tid; q < a.nnz_i; q += FUSED_THREADS) cv_i[q] = mk_i2(a.col_i[q], i_as(a.val_i[q]));
  for (int q = tid; q <= N; q += FUSED_THREADS) { rp_o[q] = a.rp_o[q]; rp_i[q] = a.rp_i[q]; }
  for (int q = tid; q < N * (CP - C); q += FUSED_THREADS) {
    const int r = q / (CP - C), c = C + q % (CP - C);
    sA[r * CP + c] = 0.f;
    sB[r * CP + c] = 0.f;
  }
  const int n_tasks = N * NP;
  const int64_t blk = (int64_t)N * C;
  // tiles of this wave
  constexpr int RSLOTS = (FUSED_THREADS / 64) / NCT;   // row slots: wave w owns row tiles w / NCT + j * RSLOTS
  constexpr int TPW = 8 / RSLOTS;                      // accumulator tiles per wavefront (N <= 256: 8 row tiles)
  const int ct = wave % NCT, rs = wave / NCT;
  const int n_rt = (N + 31) >> 5;
  const int lo = lane & 31, hi = lane >> 5;
  const int col = ct * 32 + lo;

  for (int b = (int)blockIdx.x; b < a.n_samples; b += (int)gridDim.x) {
    float* T0g = a.TS + (int64_t)b * blk;
    __syncthreads();                                   // previous sample's epilogue / MFMA reads are done
    for (int t = tid; t < n_tasks; t += FUSED_THREADS) {
      const int r = t / NP, p = t - r * NP;
      *reinterpret_cast<float2*>(sA + r * CP + 2 * p) = *reinterpret_cast<const float2*>(T0g + 2 * t);
    }
    __syncthreads();
    pgt_f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Weight rows of one chunk (four groups of four k = 16 rows of the term) for this lane's output column: rows
    // 4g + 2 hi and 4g + 2 hi + 1 of every group.  L2-resident: every workgroup reads the same [5 C, NOUT] block.
    // A term's first chunk is requested BEFORE the gather phase that precedes its MFMAs, the following chunks one
    // chunk (>= 32 MFMAs per wavefront) ahead of their use.
    const int n_groups = Kp >> 2;
    auto load_w = [&](int seg, int g0, float (&bb)[8]) {
      const float* Wb = a.W + (int64_t)seg * C * NOUT + col;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kb = 4 * (g0 + u) + 2 * hi;
        bb[2 * u] = kb < C ? Wb[kb * NOUT] : 0.f;
        bb[2 * u + 1] = kb + 1 < C ? Wb[(kb + 1) * NOUT] : 0.f;
      }
    };
    // acc += buf[N, Kp] @ W[seg]; bn holds the term's first chunk on entry
    auto mma_term = [&](const float* buf, int seg, float (&bn)[8]) {
      const int my_tiles = rs < n_rt ? (n_rt - rs + RSLOTS - 1) / RSLOTS : 0;     // wave-uniform
      if (my_tiles == 0 || (a.dbg & 1)) return;
      const float* pa[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int row = (rs + j * RSLOTS) * 32 + lo;
        pa[j] = buf + (row < N ? row : N - 1) * CP + 2 * hi;
      }
#pragma unroll 1
      for (int g0 = 0; g0 < n_groups; g0 += 4) {
        float bc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bc[u] = bn[u];
        if (g0 + 4 < n_groups) load_w(seg, g0 + 4, bn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (g0 + u < n_groups) {
            float2 ac[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) ac[j] = *reinterpret_cast<const float2*>(pa[j] + 4 * (g0 + u));
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
              if (j < my_tiles) {
                acc[j] = PGT_MFMA_32x32x2(ac[j].x, bc[2 * u], acc[j]);
                acc[j] = PGT_MFMA_32x32x2(ac[j].y, bc[2 * u + 1], acc[j]);
              }
            }
          }
        }
      }
    };
    // sum over the slots of row r of (val * src[col, pair p]); four slots' LDS reads in flight
    auto gather = [&](const int* rp, const int2* cv, const float* src, int r, int p) {
      float2 s = make_float2(0.f, 0.f);
      if (a.dbg & 2) return s;
      int q = rp[r];
      const int e = rp[r + 1];
      for (; q + 4 <= e; q += 4) {
        int2 s4[4];
        float2 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = cv[q + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float2*>(src + s4[u].x * CP + 2 * p);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float v = f_as(s4[u].y); s.x = fmaf(v, x[u].x, s.x); s.y = fmaf(v, x[u].y, s.y); }
      }
      for (; q < e; ++q) {
        const int2 s1 = cv[q];
        const float2 x = *reinterpret_cast<const float2*>(src + s1.x * CP + 2 * p);
        const float v = f_as(s1.y);
        s.x = fmaf(v, x.x, s.x); s.y = fmaf(v, x.y, s.y);
      }
      return s;
    };
    float bw[8];
    load_w(0, 0, bw);
    mma_term(sA, 0, bw);
#pragma unroll 1
    for (int d = 0; d < 2; ++d) {
      const int* rp = d ? rp_i : rp_o;
      const int2* cv = d ? cv_i : cv_o;
      const int seg1 = 1 + d, seg2 = 3 + d;
      if (d == 1) {
        // the first direction's second hop overwrote T_0 in place: fetch it again (L2-hot, read moments ago)
        for (int t = tid; t < n_tasks; t += FUSED_THREADS) {
          const int r = t / NP, p = t - r * NP;
          *reinterpret_cast<float2*>(sA + r * CP + 2 * p) = *reinterpret_cast<const float2*>(T0g + 2 * t);
        }
        __syncthreads();
      }
      load_w(seg1, 0, bw);                            // in flight during the first hop
      // first hop: T_1 = P T_0 -> sB (and the stack)
      float* g1 = a.TS + (int64_t)seg1 * a.seg_stride + (int64_t)b * blk;
      for (int t = tid; t < n_tasks; t += FUSED_THREADS) {
        const int r = t / NP, p = t - r * NP;
        const float2 v = gather(rp, cv, sA, r, p);
        *reinterpret_cast<float2*>(sB + r * CP + 2 * p) = v;
        if (!(a.dbg & 4)) *reinterpret_cast<float2*>(g1 + 2 * t) = v;
      }
      __syncthreads();
      mma_term(sB, seg1, bw);
      load_w(seg2, 0, bw);                            // in flight during the second hop
      // second hop: T_2 = 2 P T_1 - T_0 written IN PLACE over T_0 (each element of sA is read and rewritten by its own
      // task only; every other access of this phase — the gathers and the MFMAs of T_1 — reads sB)
      float* g2 = a.TS + (int64_t)seg2 * a.seg_stride + (int64_t)b * blk;
      for (int t = tid; t < n_tasks; t += FUSED_THREADS) {
        const int r = t / NP, p = t - r * NP;
        const float2 s2 = gather(rp, cv, sB, r, p);
        float2* own = reinterpret_cast<float2*>(sA + r * CP + 2 * p);
        const float2 x0 = *own;
        const float2 v = make_float2(2.f * s2.x + -1.f * x0.x, 2.f * s2.y + -1.f * x0.y);
        *own = v;
        if (!(a.dbg & 4)) *reinterpret_cast<float2*>(g2 + 2 * t) = v;
      }
      __syncthreads();
      mma_term(sA, seg2, bw);
      __syncthreads();                                 // sA / sB are free again
    }

    // ---- epilogue straight from the accumulators: lane = output column, 16 rows per tile per lane.  One base address
    // per tile and output; the 16 rows sit at compile-time multiples of the (compile-time) row stride.
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int rt = rs + j * RSLOTS;
      if (rt >= n_rt) continue;
      const int row0 = rt * 32 + 4 * hi;                              // row of accumulator register 0
      const unsigned m0 = (unsigned)(b * N + row0);
      if constexpr (EPI == 1) {
        float* zr = a.ZR + (size_t)m0 * NOUT + col;
        const bool rgate = col >= a.O;
        const int o = a.Fin + col - a.O;
        float* xhr = a.XHR + (size_t)m0 * a.ldxhr + o;
        const float* hsrc = T0g + (size_t)row0 * C + o;               // H: columns Fin.. of T_0 (L2-hot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < N) {
            const float v = pgt_sigmoidf(acc[j][r] + bv);
            zr[dr * NOUT] = v;
            if (rgate) xhr[(size_t)dr * a.ldxhr] = hsrc[dr * C] * v;
          }
        }
      } else {
        float* ht = a.HT + (size_t)m0 * NOUT + col;
        const float* zp = a.ZR + (size_t)m0 * 2 * NOUT + col;          // Z = ZR[:, :O], O = NOUT here
        const float* hp = a.Hp + (size_t)m0 * a.ldhp + col;
        float* ho = a.Hout + (size_t)m0 * a.ldo + col;
        float* hn2 = a.Hnext ? a.Hnext + (size_t)m0 * a.ldn + col : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (row0 + dr < N) {
            const float v = tanhf(acc[j][r] + bv);
            ht[dr * NOUT] = v;
            const float hn = pgt_gru_blend(zp[dr * 2 * NOUT], hp[(size_t)dr * a.ldhp], v);
            ho[(size_t)dr * a.ldo] = hn;
            if (hn2) hn2[(size_t)dr * a.ldn] = hn;
          }
        }
      }
    }
  }
}

int g_fused_dbg = 0;     // see FusedArgs::dbg

static size_t fused_lds_bytes(int64_t N, int64_t C, int64_t nnz_o, int64_t nnz_i) {
  const int64_t CP = ((C + 3) & ~(int64_t)3) + 2;
  return (size_t)(2 * N * CP * 4 + (nnz_o + nnz_i) * 8 + 2 * (N + 1) * 4);
}

static bool fused_fits(int64_t N, int64_t C, int64_t O, int64_t nnz_o, int64_t nnz_i, int nout) {
  if (N < 1 || N > 256 || C < 2 || C % 2 != 0 || C > 128 || O < 32 || O % 32 != 0) return false;
  if (nout != 64 && nout != 128) return false;
  return fused_lds_bytes(N, C, nnz_o, nnz_i) <= (size_t)FUSED_LDS;
}

static int fused_common(const char* who, const pgt_csr* fo, const pgt_csr* fi, int64_t nnz_o, int64_t nnz_i, int64_t N,
                        int64_t n_samples, int64_t C, int64_t Fin, int64_t O, float* TS, int64_t seg_stride,
                        const float* W, int nout, FusedArgs* g) {
  PGT_REQUIRE(fo && fi && fo->rowptr && fi->rowptr && TS && W, "%s: null pointer", who);
  PGT_REQUIRE(n_samples >= 0 && Fin >= 0 && Fin + O == C, "%s: C must equal Fin + O", who);
  PGT_REQUIRE(fused_fits(N, C, O, nnz_o, nnz_i, nout), "%s: shape not covered (see pgt_dconv_fused_fits)", who);
  PGT_REQUIRE(seg_stride >= n_samples * N * C && seg_stride % 2 == 0 && pgt_aligned(TS, 8), "%s: bad stack layout", who);
  *g = FusedArgs{fo->rowptr, fo->col, fo->val, fi->rowptr, fi->col, fi->val, (int)N, (int)C, (int)Fin, (int)O, (int)nnz_o,
                 (int)nnz_i, (int)n_samples, TS, seg_stride, W, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr,
                 0, nullptr, 0, g_fused_dbg};
  return PGT_OK;
}

int g_fused_wgs = 256;   // workgroups launched (one per CU); pgt_tune("dconv_fused_wgs") lowers it for the CPU test double

}  // namespace

int pgt_fused_tune(const char* key, int value) {
  if (strcmp(key, "dconv_fused_wgs") == 0) { g_fused_wgs = value > 0 ? value : 256; return 1; }
  if (strcmp(key, "dconv_fused_dbg") == 0) { g_fused_dbg = value; return 1; }
  return 0;
}

extern "C" int pgt_dconv_fused_fits(int64_t N, int64_t C, int64_t O, int64_t K, int64_t nnz_o, int64_t nnz_i) {
  // K = 3 (five terms), hidden width 64: the gate product is 128 columns wide (four column tiles), the candidate's 64 (two)
  return (K == 3 && O == 64 && fused_fits(N, C, O, nnz_o, nnz_i, 128) && fused_fits(N, C, O, nnz_o, nnz_i, 64)) ? 1 : 0;
}

extern "C" int pgt_dconv_fused_zr_f32(const pgt_csr* fwd_o, const pgt_csr* fwd_i, int64_t nnz_o, int64_t nnz_i, int64_t N,
                                      int64_t n_samples, int64_t C, int64_t Fin, int64_t O, float* TS, int64_t seg_stride,
                                      const float* W, const float* bias, float* ZR, float* XHR, int64_t ldxhr,
                                      pgt_stream_t stream) {
  if (n_samples == 0) return PGT_OK;
  FusedArgs g;
  if (int rc = fused_common("pgt_dconv_fused_zr_f32", fwd_o, fwd_i, nnz_o, nnz_i, N, n_samples, C, Fin, O, TS, seg_stride,
                            W, (int)(2 * O), &g)) return rc;
  PGT_REQUIRE(ZR && XHR && ldxhr >= C, "pgt_dconv_fused_zr_f32: null output");
  g.bias = bias; g.ZR = ZR; g.XHR = XHR; g.ldxhr = ldxhr;
  const unsigned wgs = (unsigned)(n_samples < g_fused_wgs ? n_samples : g_fused_wgs);
  PGT_REQUIRE(O == 64, "pgt_dconv_fused_zr_f32: hidden width 64 only");
  PGT_LAUNCH((dconv_fused_kernel<1, 4>), dim3(wgs), dim3(FUSED_THREADS), stream, g);
  return pgt_check_launch("pgt_dconv_fused_zr_f32");
}

extern "C" int pgt_dconv_fused_h_f32(const pgt_csr* fwd_o, const pgt_csr* fwd_i, int64_t nnz_o, int64_t nnz_i, int64_t N,
                                     int64_t n_samples, int64_t C, int64_t Fin, int64_t O, float* TS, int64_t seg_stride,
                                     const float* W, const float* bias, float* HT, const float* ZR, const float* Hp,
                                     int64_t ldhp, float* Hout, int64_t ldo, float* Hnext, int64_t ldn,
                                     pgt_stream_t stream) {
  if (n_samples == 0) return PGT_OK;
  FusedArgs g;
  if (int rc = fused_common("pgt_dconv_fused_h_f32", fwd_o, fwd_i, nnz_o, nnz_i, N, n_samples, C, Fin, O, TS, seg_stride, W,
                            (int)O, &g)) return rc;
  PGT_REQUIRE(HT && ZR && Hp && Hout, "pgt_dconv_fused_h_f32: null pointer");
  g.bias = bias; g.HT = HT; g.ZR = const_cast<float*>(ZR); g.Hp = Hp; g.ldhp = ldhp; g.Hout = Hout; g.ldo = ldo;
  g.Hnext = Hnext; g.ldn = ldn;
  const unsigned wgs = (unsigned)(n_samples < g_fused_wgs ? n_samples : g_fused_wgs);
  PGT_REQUIRE(O == 64, "pgt_dconv_fused_h_f32: hidden width 64 only");
  PGT_LAUNCH((dconv_fused_kernel<2, 2>), dim3(wgs), dim3(FUSED_THREADS), stream, g);
  return pgt_check_launch("pgt_dconv_fused_h_f32");
}
