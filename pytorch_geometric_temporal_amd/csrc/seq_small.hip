// Whole DCRNN sequences of SMALL graphs in one workgroup per sample — the launch-bound regime of the reference's own
// configuration: BatchedDCRNN(2, 2, K = 3) on 64 windows of 12 steps over 207 nodes (examples/indexBatching/DCRNN/pems_ddp.py:80,
// pems_bay_main.py), and the DCRNN cell with K > 1 on graphs of tens of nodes (test/recurrent_test.py:274-315, Chickenpox).
//
// A sample's state is N x (in + out) floats (207 x 4 = 3.3 KB): the general path spends ~220 launches of ~10 us per training step
// on operands that never needed to leave a CU.  Here ONE workgroup owns a sample for the whole sequence:
//   forward (dcrnn.py:429-475, cell :194-219, gates :172-192, diffusion :85-106): both CSR operators (P_o, P_i: packed
//     (col, val) slots) and the diffusion stack [S = 2K - 1][N][C] live in LDS; per step: stage [X_t | H], K - 1 hop phases
//     (both directions at once), the z | r product + sigmoid, H * R into segment 0, the hops again, the candidate product +
//     tanh + blend, H_t -> out[b, t] — 2K + 2 barriers per step, one launch for B x T cell steps.
//   backward (hand-written BPTT): the TRANSPOSED operators in LDS; per step the gate adjoints, the stack gradients d_pre W^T, the
//     adjoint hops, the weight-gradient sums (each (k, n) entry owned by one thread of the sample's workgroup and accumulated
//     over the steps in a per-sample buffer: no atomics, the samples are summed by the caller in index order).
// Global memory is never a hand-off between the threads of a launch (X, the saved activations and the output are read or written,
// never both; a weight-gradient partial belongs to one thread): every barrier is an LDS-only barrier (PGT_LDS_BARRIER), so the
// stores of a step drain while the next phases run instead of being acknowledged at each of the 2K + 2 barriers (1.02 -> 0.985 ms
// per training step of the reference's BatchedDCRNN(2, 2, K = 3) at 64 windows).  Measured and dropped (profiles/r04l_*): the
// reads of a step requested a phase ahead into registers and the weight-gradient sums kept in registers — 1.035 ms, and 100
// more VGPRs: a phase is bound by its dependent LDS chains at two wavefronts per SIMD, not by its few global round trips.
// The forward saves, per (sample, step), both stacks, Z | R and the candidate (2 S N C + 3 N O floats: 40 KB at the METR-LA
// shape) for the backward pass; sums run in the order of the general path (slot order per row, k order per product).
#include <type_traits>

#include "pgt_common.h"

// lab/seq_small_lab.hip defines this to record a per-workgroup phase timeline (step, slot); a no-op in the library
#ifndef PGT_SEQ_MARK
#define PGT_SEQ_MARK(step, slot) do { } while (0)
#endif

namespace {

constexpr int SQ_THREADS = 512;
constexpr int SQ_LDS = 150 * 1024;

struct SeqArgs {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;
  int N, E_o, E_i, Fin, O, K, T, B;
  const float* X;  int64_t x_sb, x_st;      // X[b, t] = X + b * x_sb + t * x_st : [N, Fin] rows of Fin floats
  const float* H0;                          // [B, N, O] | null (zeros)
  const float* Wzr; const float* bzr; const float* Wh; const float* bh;
  float* out;  int64_t o_sb, o_st;          // out[b, t] = out + b * o_sb + t * o_st : [N, O]
  float* save;                              // [B][T][2 S N C + 3 N O] | null (inference)
  // backward only
  const float* dOut; int64_t g_sb, g_st;
  float* dX; float* dH0; float* dWpart;     // dX [B, T, N, Fin] (x strides) | null; dH0 [B, N, O] | null; dWpart [B][S C 3 O + 3 O]
  int w_lds;                                // the two stacked weights are parked in LDS for the whole launch (they fit)
};

__device__ __forceinline__ float sq_as_float(int v) { union { int i; float f; } u; u.i = v; return u.f; }
__device__ __forceinline__ int sq_as_int(float v) { union { int i; float f; } u; u.f = v; return u.i; }

struct SqLds {
  int* rp_o; int* rp_i; int2* cv_o; int2* cv_i;
  float* TS;      // [S][N][C]
  float* TV;      // backward: the saved stack of the product at hand [S][N][C]
  float* H;       // [N][O]  current state (forward) / previous state (backward)
  float* ZR;      // [N][2 O]
  float* HT;      // [N][O]   (backward)
  float* dH;      // [N][O]   running state gradient (backward)
  float* dP;      // [N][3 O] pre-activation gradients: zr | h (backward)
  float* red;     // [SQ_THREADS] partial sums of the weight-gradient entries (backward)
  const float* Wzr; const float* Wh;        // the weights as the kernel reads them: LDS copies, or the global arrays
};

static size_t sq_lds_bytes(int64_t N, int64_t E_o, int64_t E_i, int64_t C, int64_t O, int64_t K, bool bwd, bool w_lds = false) {
  const int64_t S = 2 * K - 1;
  size_t b = 2 * (size_t)(N + 1) * 4 + (size_t)(E_o + E_i) * 8 + (size_t)S * N * C * 4 + (size_t)N * O * 4 + (size_t)N * 2 * O * 4;
  if (bwd) b += (size_t)S * N * C * 4 + (size_t)N * O * 4 * 2 + (size_t)N * 3 * O * 4 + (size_t)SQ_THREADS * 4;
  if (w_lds) b += (size_t)S * C * 3 * O * 4;
  return b + 64;
}

__device__ __forceinline__ SqLds sq_carve(char* base, const SeqArgs& a, bool bwd) {
  SqLds s;
  const int C = a.Fin + a.O, S = 2 * a.K - 1;
  char* p = base;
  s.cv_o = reinterpret_cast<int2*>(p); p += (size_t)a.E_o * 8;
  s.cv_i = reinterpret_cast<int2*>(p); p += (size_t)a.E_i * 8;
  s.rp_o = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  s.rp_i = reinterpret_cast<int*>(p); p += (size_t)(a.N + 1) * 4;
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
  s.TS = reinterpret_cast<float*>(p); p += (size_t)S * a.N * C * 4;
  s.H = reinterpret_cast<float*>(p); p += (size_t)a.N * a.O * 4;
  s.ZR = reinterpret_cast<float*>(p); p += (size_t)a.N * 2 * a.O * 4;
  s.TV = s.HT = s.dH = s.dP = s.red = nullptr;
  if (bwd) {
    s.TV = reinterpret_cast<float*>(p); p += (size_t)S * a.N * C * 4;
    s.HT = reinterpret_cast<float*>(p); p += (size_t)a.N * a.O * 4;
    s.dH = reinterpret_cast<float*>(p); p += (size_t)a.N * a.O * 4;
    s.dP = reinterpret_cast<float*>(p); p += (size_t)a.N * 3 * a.O * 4;
    s.red = reinterpret_cast<float*>(p); p += (size_t)SQ_THREADS * 4;
  }
  s.Wzr = a.Wzr; s.Wh = a.Wh;
  if (a.w_lds) {
    // (a product entry is S C multiply-adds against a column of W: read from global memory every element is its own dependent
    // round trip — 280 us per Chickenpox snapshot in the first version; out of LDS the products run at LDS rate)
    float* wz = reinterpret_cast<float*>(p); p += (size_t)S * C * 2 * a.O * 4;
    float* wh = reinterpret_cast<float*>(p);
    for (int i = threadIdx.x; i < S * C * 2 * a.O; i += SQ_THREADS) wz[i] = a.Wzr[i];
    for (int i = threadIdx.x; i < S * C * a.O; i += SQ_THREADS) wh[i] = a.Wh[i];
    s.Wzr = wz; s.Wh = wh;
  }
  return s;
}

__device__ __forceinline__ void sq_stage_csr(const SeqArgs& a, const SqLds& s, int tid) {
  for (int i = tid; i <= a.N; i += SQ_THREADS) { s.rp_o[i] = a.rp_o[i]; s.rp_i[i] = a.rp_i[i]; }
  for (int i = tid; i < a.E_o; i += SQ_THREADS) { int2 t; t.x = a.col_o[i]; t.y = sq_as_int(a.val_o[i]); s.cv_o[i] = t; }
  for (int i = tid; i < a.E_i; i += SQ_THREADS) { int2 t; t.x = a.col_i[i]; t.y = sq_as_int(a.val_i[i]); s.cv_i[i] = t; }
}

// sum over the slots of row n of one operator, sequential fma chain in slot order (as the general path's kernels)
__device__ __forceinline__ float sq_row(const int* rp, const int2* cv, const float* src, int n, int c, int C) {
  float acc = 0.f;
  int q = rp[n];
  const int e = rp[n + 1];
  for (; q + 4 <= e; q += 4) {                        // four slots' LDS reads in flight; the sum stays in slot order
    int2 s4[4];
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = cv[q + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = src[s4[u].x * C + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fmaf(sq_as_float(s4[u].y), x[u], acc);
  }
  for (; q < e; ++q) {
    const int2 s = cv[q];
    acc = fmaf(sq_as_float(s.y), src[s.x * C + c], acc);
  }
  return acc;
}

// the same row sum for the four channels of a quad at once (C a multiple of 4: the reference's BatchedDCRNN(2, 2) has C = 4, a
// node's whole row): a slot is read ONCE per quad instead of once per channel and the source as one ds_read_b128 — a quarter of the
// LDS instructions, and a hop level is one element per thread instead of four in a row.  Per channel the same fmaf chain.
__device__ __forceinline__ pgt_f4 sq_row4(const int* rp, const int2* cv, const float* src, int n, int c4, int C) {
  pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
  int q = rp[n];
  const int e = rp[n + 1];
  for (; q < e; q += 4) {                              // up to four slots in flight; a short last batch re-reads slot q (never added)
    int2 s4[4];
    pgt_f4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = cv[q + u < e ? q + u : q];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const pgt_f4*>(src + s4[u].x * C + c4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = sq_as_float(s4[u].y);
      if (q + u < e) acc = pgt_mk4(fmaf(v, x[u].x, acc.x), fmaf(v, x[u].y, acc.y), fmaf(v, x[u].z, acc.z), fmaf(v, x[u].w, acc.w));
    }
  }
  return acc;
}
__device__ __forceinline__ pgt_f4 sq_ld4(const float* p) { return *reinterpret_cast<const pgt_f4*>(p); }
__device__ __forceinline__ void sq_st4(float* p, pgt_f4 v) { *reinterpret_cast<pgt_f4*>(p) = v; }

// the K - 1 hop levels on TS (segment 0 given): T1d = P_d T0, Tkd = 2 P_d T(k-1)d - T0   (dcrnn.py:85-106; Tx_0 is never advanced)
__device__ __forceinline__ void sq_hops(const SeqArgs& a, const SqLds& s, int tid) {
  const int C = a.Fin + a.O, NC = a.N * C;
  for (int k = 1; k < a.K; ++k) {
    if ((C & 3) == 0) {
      const int Q = C >> 2, NQ = a.N * Q;
      for (int e = tid; e < 2 * NQ; e += SQ_THREADS) {
        const int d = e >= NQ, r = e - d * NQ, n = r / Q, off = n * C + 4 * (r - n * Q);
        const float* src = s.TS + (size_t)(k == 1 ? 0 : 2 * (k - 1) - 1 + d) * NC;
        const pgt_f4 g = sq_row4(d ? s.rp_i : s.rp_o, d ? s.cv_i : s.cv_o, src, n, off - n * C, C);
        const pgt_f4 t0 = sq_ld4(s.TS + off);
        sq_st4(s.TS + (size_t)(2 * k - 1 + d) * NC + off,
               k == 1 ? g : pgt_mk4(2.f * g.x - t0.x, 2.f * g.y - t0.y, 2.f * g.z - t0.z, 2.f * g.w - t0.w));
      }
    } else {
      for (int e = tid; e < 2 * NC; e += SQ_THREADS) {
        const int d = e >= NC, r = e - d * NC, n = r / C, c = r - n * C;
        const float* src = s.TS + (size_t)(k == 1 ? 0 : 2 * (k - 1) - 1 + d) * NC;
        const float g = d ? sq_row(s.rp_i, s.cv_i, src, n, c, C) : sq_row(s.rp_o, s.cv_o, src, n, c, C);
        s.TS[(size_t)(2 * k - 1 + d) * NC + r] = k == 1 ? g : 2.f * g - s.TS[r];
      }
    }
    PGT_LDS_BARRIER();
  }
}

// pre[n][j] = bias[j] + sum_{s, c} TS[s][n][c] W[(s C + c) ldw + j], k ascending (the order of the MFMA chain of the general path)
__device__ __forceinline__ float sq_dot(const SeqArgs& a, const float* TS, const float* __restrict__ W, int ldw, int n, int j) {
  const int C = a.Fin + a.O, S = 2 * a.K - 1, NC = a.N * C;
  float acc = 0.f;
  for (int sg = 0; sg < S; ++sg) {
    const float* t = TS + (size_t)sg * NC + n * C;
    const float* w = W + (size_t)sg * C * ldw + j;
#pragma unroll 4
    for (int c = 0; c < C; ++c) acc = fmaf(t[c], w[(size_t)c * ldw], acc);
  }
  return acc;
}

// LDS_BYTES: the static LDS of the instantiation (three sizes: small samples leave room for 2 - 4 resident workgroups per CU)
// (Round 4 also built "one thread per NODE" forms of the two products — a node's stack row read once, the weight rows as broadcast
// vector reads.  Measured in round 5, they lose everywhere: B = 64 / 256 / 1024 at hidden 2: 0.565 / 0.596 / 1.480 ms per step
// against 0.591 / 0.628 / 2.008 with them, profiles/r05a_seq_vdot_ab.txt — 207 busy threads of 512 and 30 more VGPRs.  Deleted.)
template <int LDS_BYTES>
__global__ __launch_bounds__(SQ_THREADS) void dcrnn_seq_small_fwd_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int C = a.Fin + a.O, O = a.O, S = 2 * a.K - 1, NC = a.N * C, NO = a.N * O;
  SqLds s = sq_carve(smem, a, false);
  sq_stage_csr(a, s, tid);
  const int64_t per_step = (int64_t)2 * S * NC + 3 * NO;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    for (int e = tid; e < NO; e += SQ_THREADS) s.H[e] = a.H0 ? a.H0[(int64_t)b * NO + e] : 0.f;
    PGT_LDS_BARRIER();
    for (int t = 0; t < a.T; ++t) {
      const float* x = a.X + b * a.x_sb + t * a.x_st;
      float* sv = a.save ? a.save + ((int64_t)b * a.T + t) * per_step : nullptr;
      for (int e = tid; e < NC; e += SQ_THREADS) {          // segment 0 = [X_t | H]
        const int n = e / C, c = e - n * C;
        s.TS[e] = c < a.Fin ? x[n * a.Fin + c] : s.H[n * O + (c - a.Fin)];
      }
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(t, 0);                                    // [X_t | H] staged
      sq_hops(a, s, tid);
      PGT_SEQ_MARK(t, 1);                                    // hops of the z | r stack
      for (int e = tid; e < 2 * NO; e += SQ_THREADS) {       // Z | R = sigmoid(stack Wzr + bzr)
        const int n = e / (2 * O), j = e - n * 2 * O;
        const float v = pgt_sigmoidf(sq_dot(a, s.TS, s.Wzr, 2 * O, n, j) + (a.bzr ? a.bzr[j] : 0.f));
        s.ZR[e] = v;
        if (sv) sv[(int64_t)2 * S * NC + e] = v;
      }
      if (sv) for (int e = tid; e < S * NC; e += SQ_THREADS) sv[e] = s.TS[e];
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(t, 2);                                    // z | r product, sigmoid, stack saved
      for (int e = tid; e < NO; e += SQ_THREADS) {           // segment 0 = [X_t | H * R]
        const int n = e / O, o = e - n * O;
        s.TS[n * C + a.Fin + o] = s.H[e] * s.ZR[n * 2 * O + O + o];
      }
      PGT_LDS_BARRIER();
      sq_hops(a, s, tid);
      PGT_SEQ_MARK(t, 3);                                    // H * R, hops of the candidate's stack
      float* o_t = a.out + b * a.o_sb + t * a.o_st;
      for (int e = tid; e < NO; e += SQ_THREADS) {           // candidate, blend, H_t
        const int n = e / O, o = e - n * O;
        const float ht = tanhf(sq_dot(a, s.TS, s.Wh, O, n, o) + (a.bh ? a.bh[o] : 0.f));
        const float hn = pgt_gru_blend(s.ZR[n * 2 * O + o], s.H[e], ht);
        if (sv) sv[(int64_t)2 * S * NC + 2 * NO + e] = ht;
        o_t[e] = hn;
        s.H[e] = hn;                                          // (n, o) is read by this thread only in this phase
      }
      if (sv) for (int e = tid; e < S * NC; e += SQ_THREADS) sv[(int64_t)S * NC + e] = s.TS[e];
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(t, 4);                                    // candidate product, blend, H_t out, stack saved
    }
  }
}

// adjoint of sq_hops on the gradient stack G (all S segments given): leaves d/dT0 in segment 0.  tp_* are the TRANSPOSED operators.
//   Tkd = 2 P_d T(k-1)d - T0 (k >= 2)  ->  G(k-1)d += 2 P_d^T Gkd,  G0 -= Gkd;     T1d = P_d T0  ->  G0 += P_d^T G1d
__device__ __forceinline__ void sq_hops_adjoint(const SeqArgs& a, const SqLds& s, float* G, int tid) {
  const int C = a.Fin + a.O, NC = a.N * C;
  const bool quads = (C & 3) == 0;
  const int Q = C >> 2, NQ = a.N * Q;
  for (int k = a.K - 1; k >= 2; --k) {
    if (quads) {
      for (int e = tid; e < 2 * NQ; e += SQ_THREADS) {
        const int d = e >= NQ, r = e - d * NQ, n = r / Q, c4 = 4 * (r - n * Q);
        const float* src = G + (size_t)(2 * k - 1 + d) * NC;
        const pgt_f4 g = sq_row4(d ? s.rp_i : s.rp_o, d ? s.cv_i : s.cv_o, src, n, c4, C);
        float* at = G + (size_t)(2 * (k - 1) - 1 + d) * NC + n * C + c4;
        const pgt_f4 old = sq_ld4(at);
        sq_st4(at, pgt_mk4(old.x + 2.f * g.x, old.y + 2.f * g.y, old.z + 2.f * g.z, old.w + 2.f * g.w));
      }
    } else {
      for (int e = tid; e < 2 * NC; e += SQ_THREADS) {
        const int d = e >= NC, r = e - d * NC, n = r / C, c = r - n * C;
        const float* src = G + (size_t)(2 * k - 1 + d) * NC;
        const float g = d ? sq_row(s.rp_i, s.cv_i, src, n, c, C) : sq_row(s.rp_o, s.cv_o, src, n, c, C);
        G[(size_t)(2 * (k - 1) - 1 + d) * NC + r] += 2.f * g;
      }
    }
    for (int r = tid; r < NC; r += SQ_THREADS) G[r] -= G[(size_t)(2 * k - 1) * NC + r] + G[(size_t)(2 * k) * NC + r];
    PGT_LDS_BARRIER();
  }
  if (a.K >= 2) {
    if (quads) {
      for (int r = tid; r < NQ; r += SQ_THREADS) {
        const int n = r / Q, c4 = 4 * (r - n * Q);
        const pgt_f4 go = sq_row4(s.rp_o, s.cv_o, G + NC, n, c4, C);
        PGT_SCHED_FENCE();                                   // one direction after the other: interleaved they only cost registers
        const pgt_f4 gi = sq_row4(s.rp_i, s.cv_i, G + 2 * (size_t)NC, n, c4, C);
        float* at = G + n * C + c4;
        const pgt_f4 old = sq_ld4(at);
        sq_st4(at, pgt_mk4(old.x + (go.x + gi.x), old.y + (go.y + gi.y), old.z + (go.z + gi.z), old.w + (go.w + gi.w)));
      }
    } else {
      for (int r = tid; r < NC; r += SQ_THREADS) {
        const int n = r / C, c = r - n * C;
        G[r] += sq_row(s.rp_o, s.cv_o, G + NC, n, c, C) + sq_row(s.rp_i, s.cv_i, G + 2 * (size_t)NC, n, c, C);
      }
    }
    PGT_LDS_BARRIER();
  }
}

// one gate product's adjoint: weight-gradient sums into the sample's buffer (dW[(s C + c) ldw + j] += sum_n TV[s][n][c] dP[n][j0 + j]),
// bias sums, and the stack gradient G[s][n][c] = sum_j dP[n][j0 + j] W[(s C + c) ldw + j]
__device__ __forceinline__ void sq_product_adjoint(const SeqArgs& a, const SqLds& s, const float* __restrict__ W, int ldw, int j0,
                                                   float* dW, float* db, int tid) {
  const int C = a.Fin + a.O, O = a.O, S = 2 * a.K - 1, NC = a.N * C;
  const int nE = S * C * ldw, nAll = nE + ldw;               // weight entries, then the bias entries
  int P = SQ_THREADS / nAll;
  if (P > 16) P = 16;
  if (P >= 2) {
    // narrow cells (the reference's BatchedDCRNN(2, 2, K = 3): 80 + 4 entries): one thread per entry would walk all N nodes in a
    // dependent chain while 400 threads idle — the nodes are cut into P runs, a thread sums one run of one entry, the runs of an
    // entry are added in run order (a fixed order: deterministic)
    // (unroll 2, not 4: with the deeper unroll the kernel needs 142 VGPRs and only ONE workgroup fits a CU — B = 1024 loses 12 %)
    const int run = (a.N + P - 1) / P;
    if (tid < nAll * P) {
      const int p = tid / nAll, e = tid - p * nAll;
      const int n0 = p * run, n1 = n0 + run < a.N ? n0 + run : a.N;
      float acc = 0.f;
      if (e < nE) {
        const int sc = e / ldw, j = e - sc * ldw, sg = sc / C, c = sc - sg * C;
        const float* tv = s.TV + (size_t)sg * NC + c;
#pragma unroll 2
        for (int n = n0; n < n1; ++n) acc = fmaf(tv[n * C], s.dP[n * 3 * O + j0 + j], acc);
      } else {
#pragma unroll 2
        for (int n = n0; n < n1; ++n) acc += s.dP[n * 3 * O + j0 + (e - nE)];
      }
      s.red[tid] = acc;
    }
    PGT_LDS_BARRIER();
    if (tid < nAll) {
      float t = 0.f;
      for (int p = 0; p < P; ++p) t += s.red[p * nAll + tid];
      if (tid < nE) dW[tid] += t;
      else db[tid - nE] += t;
    }
  } else {
    for (int e = tid; e < nE; e += SQ_THREADS) {
      const int sc = e / ldw, j = e - sc * ldw, sg = sc / C, c = sc - sg * C;
      const float* tv = s.TV + (size_t)sg * NC + c;
      float acc = 0.f;
#pragma unroll 4
      for (int n = 0; n < a.N; ++n) acc = fmaf(tv[n * C], s.dP[n * 3 * O + j0 + j], acc);
      dW[e] += acc;
    }
    for (int j = tid; j < ldw; j += SQ_THREADS) {
      float acc = 0.f;
#pragma unroll 4
      for (int n = 0; n < a.N; ++n) acc += s.dP[n * 3 * O + j0 + j];
      db[j] += acc;
    }
  }
  for (int e = tid; e < S * NC; e += SQ_THREADS) {
    const int sg = e / NC, r = e - sg * NC, n = r / C, c = r - n * C;
    const float* w = W + ((size_t)sg * C + c) * ldw;
    const float* g = s.dP + n * 3 * O + j0;
    float acc = 0.f;
#pragma unroll 4
    for (int j = 0; j < ldw; ++j) acc = fmaf(g[j], w[j], acc);
    s.TS[e] = acc;
  }
}

template <int LDS_BYTES>
__global__ __launch_bounds__(SQ_THREADS) void dcrnn_seq_small_bwd_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int C = a.Fin + a.O, O = a.O, S = 2 * a.K - 1, NC = a.N * C, NO = a.N * O;
  SqLds s = sq_carve(smem, a, true);
  sq_stage_csr(a, s, tid);                                   // (the caller passes the transposed operators)
  const int64_t per_step = (int64_t)2 * S * NC + 3 * NO;
  const int64_t nW = (int64_t)S * C * 3 * O + 3 * O;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float* dWzr = a.dWpart + b * nW;
    float* dWh = dWzr + (int64_t)S * C * 2 * O;
    float* dbzr = dWh + (int64_t)S * C * O;
    float* dbh = dbzr + 2 * O;
    for (int e = tid; e < NO; e += SQ_THREADS) s.dH[e] = 0.f;
    PGT_LDS_BARRIER();
    for (int t = a.T - 1; t >= 0; --t) {
      const float* sv = a.save + ((int64_t)b * a.T + t) * per_step;
      const float* g_t = a.dOut + b * a.g_sb + t * a.g_st;
      const float* hprev = t > 0 ? a.out + b * a.o_sb + (t - 1) * a.o_st : (a.H0 ? a.H0 + (int64_t)b * NO : nullptr);
      // ---- gate adjoints of the blend and the candidate; the candidate's stack
      for (int e = tid; e < NO; e += SQ_THREADS) {
        const int n = e / O, o = e - n * O;
        const float z = sv[(int64_t)2 * S * NC + n * 2 * O + o], r = sv[(int64_t)2 * S * NC + n * 2 * O + O + o];
        const float ht = sv[(int64_t)2 * S * NC + 2 * NO + e], h = hprev ? hprev[e] : 0.f;
        const float g = g_t[e] + s.dH[e];
        s.ZR[n * 2 * O + o] = z; s.ZR[n * 2 * O + O + o] = r;
        s.HT[e] = ht; s.H[e] = h;
        s.dP[n * 3 * O + 2 * O + o] = g * (1.f - z) * (1.f - ht * ht);
        s.dP[n * 3 * O + o] = g * (h - ht) * z * (1.f - z);
        s.dH[e] = g * z;
      }
      for (int e = tid; e < S * NC; e += SQ_THREADS) s.TV[e] = sv[(int64_t)S * NC + e];
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(a.T - 1 - t, 0);                          // gate adjoints, the candidate's saved stack in LDS
      sq_product_adjoint(a, s, s.Wh, O, 2 * O, dWh, dbh, tid);
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(a.T - 1 - t, 1);                          // candidate product adjoint (weight sums + stack gradient)
      sq_hops_adjoint(a, s, s.TS, tid);
      PGT_SEQ_MARK(a.T - 1 - t, 2);                          // adjoint hops
      // ---- d(H R): the reset gate's pre-activation, the state; the input columns of this stack's d/dT0
      float* dx = a.dX ? a.dX + b * a.x_sb + t * a.x_st : nullptr;
      for (int e = tid; e < NO; e += SQ_THREADS) {
        const int n = e / O, o = e - n * O;
        const float dhr = s.TS[n * C + a.Fin + o], r = s.ZR[n * 2 * O + O + o];
        s.dP[n * 3 * O + O + o] = dhr * s.H[e] * r * (1.f - r);
        s.dH[e] += dhr * r;
      }
      if (dx) for (int e = tid; e < a.N * a.Fin; e += SQ_THREADS) { const int n = e / a.Fin, f = e - n * a.Fin; dx[e] = s.TS[n * C + f]; }
      for (int e = tid; e < S * NC; e += SQ_THREADS) s.TV[e] = sv[e];
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(a.T - 1 - t, 3);                          // d(H R), the z | r stack in LDS
      sq_product_adjoint(a, s, s.Wzr, 2 * O, 0, dWzr, dbzr, tid);
      PGT_LDS_BARRIER();
      PGT_SEQ_MARK(a.T - 1 - t, 4);                          // z | r product adjoint
      sq_hops_adjoint(a, s, s.TS, tid);
      PGT_SEQ_MARK(a.T - 1 - t, 5);                          // adjoint hops
      for (int e = tid; e < NO; e += SQ_THREADS) { const int n = e / O, o = e - n * O; s.dH[e] += s.TS[n * C + a.Fin + o]; }
      if (dx) for (int e = tid; e < a.N * a.Fin; e += SQ_THREADS) { const int n = e / a.Fin, f = e - n * a.Fin; dx[e] += s.TS[n * C + f]; }
      PGT_LDS_BARRIER();
    }
    if (a.dH0) for (int e = tid; e < NO; e += SQ_THREADS) a.dH0[(int64_t)b * NO + e] = s.dH[e];
    PGT_LDS_BARRIER();
  }
}

static int sq_take(const char* who, const pgt_csr* op_o, const pgt_csr* op_i, int64_t E_o, int64_t E_i, int64_t N, int64_t B,
                   int64_t T, int64_t Fin, int64_t O, int64_t K, bool bwd, SeqArgs* a) {
  PGT_REQUIRE(op_o && op_i && op_o->rowptr && op_i->rowptr, "%s: null operator", who);
  PGT_REQUIRE(N >= 1 && B >= 0 && T >= 0 && Fin >= 0 && O >= 1 && K >= 1 && E_o >= 0 && E_i >= 0, "%s: bad size", who);
  PGT_REQUIRE(E_o == 0 || (op_o->col && op_o->val), "%s: null operator arrays", who);
  PGT_REQUIRE(E_i == 0 || (op_i->col && op_i->val), "%s: null operator arrays", who);
  PGT_REQUIRE(sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, bwd) <= (size_t)SQ_LDS && N * (Fin + O) * (2 * K - 1) < (1 << 24),
              "%s: the sample does not fit a workgroup's LDS (see pgt_dcrnn_seq_small_fits)", who);
  a->rp_o = op_o->rowptr; a->col_o = op_o->col; a->val_o = op_o->val;
  a->rp_i = op_i->rowptr; a->col_i = op_i->col; a->val_i = op_i->val;
  a->N = (int)N; a->E_o = (int)E_o; a->E_i = (int)E_i; a->Fin = (int)Fin; a->O = (int)O; a->K = (int)K; a->T = (int)T; a->B = (int)B;
  return PGT_OK;
}

}  // namespace

extern "C" int pgt_dcrnn_seq_small_fits(int64_t N, int64_t E_o, int64_t E_i, int64_t Fin, int64_t O, int64_t K) {
  if (N < 1 || Fin < 0 || O < 1 || K < 1 || E_o < 0 || E_i < 0 || N > 65535) return 0;
  if (N * (Fin + O) * (2 * K - 1) >= (1 << 24)) return 0;
  return sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, true) <= (size_t)SQ_LDS ? 1 : 0;
}

extern "C" int64_t pgt_dcrnn_seq_small_save_floats(int64_t N, int64_t Fin, int64_t O, int64_t K) {
  return 2 * (2 * K - 1) * N * (Fin + O) + 3 * N * O;
}

extern "C" int pgt_dcrnn_seq_small_f32(const pgt_csr* op_o, const pgt_csr* op_i, int64_t E_o, int64_t E_i, int64_t N,
                                       const float* X, int64_t x_stride_b, int64_t x_stride_t, const float* H0, const float* Wzr,
                                       const float* bzr, const float* Wh, const float* bh, int64_t B, int64_t T, int64_t Fin,
                                       int64_t O, int64_t K, float* out, int64_t out_stride_b, int64_t out_stride_t, float* save,
                                       pgt_stream_t stream) {
  SeqArgs a{};
  if (int rc = sq_take("pgt_dcrnn_seq_small_f32", op_o, op_i, E_o, E_i, N, B, T, Fin, O, K, false, &a)) return rc;
  if (B == 0 || T == 0) return PGT_OK;
  PGT_REQUIRE((Fin == 0 || X) && Wzr && Wh && out, "pgt_dcrnn_seq_small_f32: null pointer");
  a.X = X; a.x_sb = x_stride_b; a.x_st = x_stride_t; a.H0 = H0; a.Wzr = Wzr; a.bzr = bzr; a.Wh = Wh; a.bh = bh;
  a.out = out; a.o_sb = out_stride_b; a.o_st = out_stride_t; a.save = save;
  const int64_t wgs = B < 2048 ? B : 2048;
  a.w_lds = sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, false, true) <= (size_t)SQ_LDS ? 1 : 0;
  const size_t need = sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, false, a.w_lds != 0);
#define SQ_GO(KERN, BYTES) PGT_LAUNCH((KERN<BYTES>), dim3((unsigned)wgs), dim3(SQ_THREADS), stream, a)
  if (need <= 38 * 1024) SQ_GO(dcrnn_seq_small_fwd_kernel, 38 * 1024);
  else if (need <= 78 * 1024) SQ_GO(dcrnn_seq_small_fwd_kernel, 78 * 1024);
  else SQ_GO(dcrnn_seq_small_fwd_kernel, SQ_LDS);
  return pgt_check_launch("pgt_dcrnn_seq_small_f32");
}

extern "C" int pgt_dcrnn_seq_small_bwd_f32(const pgt_csr* tp_o, const pgt_csr* tp_i, int64_t E_o, int64_t E_i, int64_t N,
                                           const float* dOut, int64_t g_stride_b, int64_t g_stride_t, const float* out,
                                           int64_t out_stride_b, int64_t out_stride_t, const float* H0, const float* save,
                                           const float* Wzr, const float* Wh, int64_t B, int64_t T, int64_t Fin, int64_t O, int64_t K,
                                           float* dX, int64_t x_stride_b, int64_t x_stride_t, float* dH0, float* dWpart,
                                           pgt_stream_t stream) {
  SeqArgs a{};
  if (int rc = sq_take("pgt_dcrnn_seq_small_bwd_f32", tp_o, tp_i, E_o, E_i, N, B, T, Fin, O, K, true, &a)) return rc;
  if (B == 0 || T == 0) return PGT_OK;
  PGT_REQUIRE(dOut && out && save && Wzr && Wh && dWpart, "pgt_dcrnn_seq_small_bwd_f32: null pointer");
  a.dOut = dOut; a.g_sb = g_stride_b; a.g_st = g_stride_t; a.out = const_cast<float*>(out); a.o_sb = out_stride_b; a.o_st = out_stride_t;
  a.H0 = H0; a.save = const_cast<float*>(save); a.Wzr = Wzr; a.Wh = Wh;
  a.dX = dX; a.x_sb = x_stride_b; a.x_st = x_stride_t; a.dH0 = dH0; a.dWpart = dWpart;
  const int64_t wgs = B < 2048 ? B : 2048;
  a.w_lds = sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, true, true) <= (size_t)SQ_LDS ? 1 : 0;
  const size_t need = sq_lds_bytes(N, E_o, E_i, Fin + O, O, K, true, a.w_lds != 0);
  if (need <= 38 * 1024) SQ_GO(dcrnn_seq_small_bwd_kernel, 38 * 1024);
  else if (need <= 78 * 1024) SQ_GO(dcrnn_seq_small_bwd_kernel, 78 * 1024);
  else SQ_GO(dcrnn_seq_small_bwd_kernel, SQ_LDS);
#undef SQ_GO
  return pgt_check_launch("pgt_dcrnn_seq_small_bwd_f32");
}
