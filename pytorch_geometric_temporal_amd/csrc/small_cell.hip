// The DCRNN cell without diffusion (K = 1) in one launch per direction (dcrnn.py:79-82 + 172-192).
//
// With K = 1 the reference's DConv is `X @ W[0,0] + X @ W[1,0] + b` (dcrnn.py:79-82: the hop loop never runs, the graph only
// feeds degrees that nothing reads), so the whole cell is dense and its rows are independent:
//     Z = sig([X,H] Wz + bz)    R = sig([X,H] Wr + br)    Hc = tanh([X, R*H] Wh + bh)    H' = Z*H + (1-Z)*Hc
// This is BASELINE configs[0] (examples/recurrent/dcrnn_example.py:19-28: DCRNN(4, 32, 1) on 20 Chickenpox counties, 103
// snapshots per epoch, H = None every call): through the general path one snapshot is a weight re-stacking, a staging
// copy, two products with gate epilogues and their adjoints — a dozen launches of a few microseconds for 20 x 36 numbers.
// Here: one launch forward (32-row groups through LDS, the three parameter tensors read in their own [2, 1, C, O] layout,
// H = None as a null pointer), one workgroup backward (gate chain, d/dX, d/dH, the three weight gradients written to both
// halves of [2, 1, C, O], the bias gradients).  fmaf dot products in index order: deterministic, no atomics.
#include "pgt_common.h"

namespace {

constexpr int SC_ROWS = 32;            // rows of one pass through LDS
constexpr int SC_MAX_C = 128;          // in_channels + out_channels
constexpr int SC_MAX_O = 64;           // out_channels
constexpr int SC_FWD_THREADS = 256;
constexpr int SC_BWD_THREADS = 512;
constexpr int SC_BWD_MAX_ROWS = 4096;  // the adjoint is ONE workgroup (its weight gradients sum over all rows)

struct CellK1Args {
  const float* X; int64_t ldx;
  const float* H; int64_t ldh;                        // null: zeros (DCRNN._set_hidden_state, dcrnn.py:167-170)
  const float* Wz; const float* Wr; const float* Wh;  // [2][1][C][O] each
  const float* bz; const float* br; const float* bh;  // [O] or null
  float* Hnew; int64_t ldo;
  float* saved;                                       // [N][3 O]: Z | R | Hc
  int N, Fin, O;
};

__global__ __launch_bounds__(SC_FWD_THREADS) void cell_k1_fwd_kernel(CellK1Args a) {
  __shared__ float s_xh[SC_ROWS * SC_MAX_C];          // [r][C]: X | H
  __shared__ float s_rh[SC_ROWS * SC_MAX_O];          // R * H
  __shared__ float s_z[SC_ROWS * SC_MAX_O];
  const int tid = threadIdx.x, Fin = a.Fin, O = a.O, C = Fin + O, CO = C * O;
  const int r0 = blockIdx.x * SC_ROWS, nr = (a.N - r0 < SC_ROWS) ? a.N - r0 : SC_ROWS;
  for (int e = tid; e < nr * C; e += SC_FWD_THREADS) {
    const int r = e / C, c = e - r * C;
    float v;
    if (c < Fin) v = a.X[(int64_t)(r0 + r) * a.ldx + c];
    else v = a.H ? a.H[(int64_t)(r0 + r) * a.ldh + (c - Fin)] : 0.f;
    s_xh[r * C + c] = v;
  }
  __syncthreads();
  for (int e = tid; e < nr * O; e += SC_FWD_THREADS) {
    const int r = e / O, o = e - r * O;
    const float* x = s_xh + r * C;
    float z0 = 0.f, z1 = 0.f, q0 = 0.f, q1 = 0.f;      // the two halves are separate products in the reference
    for (int c = 0; c < C; ++c) {
      const float xv = x[c];
      z0 = fmaf(xv, a.Wz[c * O + o], z0);
      z1 = fmaf(xv, a.Wz[CO + c * O + o], z1);
      q0 = fmaf(xv, a.Wr[c * O + o], q0);
      q1 = fmaf(xv, a.Wr[CO + c * O + o], q1);
    }
    const float Z = pgt_sigmoidf(z0 + z1 + (a.bz ? a.bz[o] : 0.f));
    const float R = pgt_sigmoidf(q0 + q1 + (a.br ? a.br[o] : 0.f));
    s_z[r * O + o] = Z;
    s_rh[r * O + o] = R * x[Fin + o];
    float* sv = a.saved + (int64_t)(r0 + r) * 3 * O;
    sv[o] = Z;
    sv[O + o] = R;
  }
  __syncthreads();
  for (int e = tid; e < nr * O; e += SC_FWD_THREADS) {
    const int r = e / O, o = e - r * O;
    const float* x = s_xh + r * C;
    const float* rh = s_rh + r * O;
    float p0 = 0.f, p1 = 0.f;
    for (int c = 0; c < Fin; ++c) {
      p0 = fmaf(x[c], a.Wh[c * O + o], p0);
      p1 = fmaf(x[c], a.Wh[CO + c * O + o], p1);
    }
    for (int j = 0; j < O; ++j) {
      p0 = fmaf(rh[j], a.Wh[(Fin + j) * O + o], p0);
      p1 = fmaf(rh[j], a.Wh[CO + (Fin + j) * O + o], p1);
    }
    const float Hc = tanhf(p0 + p1 + (a.bh ? a.bh[o] : 0.f));
    a.saved[(int64_t)(r0 + r) * 3 * O + 2 * O + o] = Hc;
    a.Hnew[(int64_t)(r0 + r) * a.ldo + o] = pgt_gru_blend(s_z[r * O + o], x[Fin + o], Hc);
  }
}

struct CellK1BwdArgs {
  const float* G; int64_t ldg;                        // d/dH'
  const float* X; int64_t ldx;
  const float* H; int64_t ldh;                        // null: zeros
  const float* Wz; const float* Wr; const float* Wh;
  const float* saved;                                 // [N][3 O]
  float* dX; int64_t lddx;                            // null: not wanted
  float* dH; int64_t lddh;                            // null: not wanted
  float* dWz; float* dWr; float* dWh;                 // [2][1][C][O] each (both halves receive the same gradient)
  float* dbz; float* dbr; float* dbh;                 // [O] or null
  float* dP;                                          // scratch [N][3 O]: d/d(pre-activation) of z | r | h
  int N, Fin, O;
};

__global__ __launch_bounds__(SC_BWD_THREADS) void cell_k1_bwd_kernel(CellK1BwdArgs a) {
  __shared__ float s_dpz[SC_ROWS * SC_MAX_O], s_dpr[SC_ROWS * SC_MAX_O], s_dph[SC_ROWS * SC_MAX_O];
  __shared__ float s_dh[SC_ROWS * SC_MAX_O];          // d/dH through the blend and through R*H
  __shared__ float s_h[SC_ROWS * SC_MAX_O], s_r[SC_ROWS * SC_MAX_O];
  __shared__ float s_dx[SC_ROWS * SC_MAX_C];          // d/dX through the candidate product
  const int tid = threadIdx.x, Fin = a.Fin, O = a.O, C = Fin + O, CO = C * O;
  const bool want_in = a.dX != nullptr || a.dH != nullptr;
  for (int r0 = 0; r0 < a.N; r0 += SC_ROWS) {
    const int nr = (a.N - r0 < SC_ROWS) ? a.N - r0 : SC_ROWS;
    for (int e = tid; e < nr * O; e += SC_BWD_THREADS) {
      const int r = e / O, o = e - r * O;
      const int64_t row = r0 + r;
      const float* sv = a.saved + row * 3 * O;
      const float g = a.G[row * a.ldg + o], Z = sv[o], R = sv[O + o], Hc = sv[2 * O + o];
      const float h = a.H ? a.H[row * a.ldh + o] : 0.f;
      const float dph = g * (1.f - Z) * (1.f - Hc * Hc);
      const float dpz = g * (h - Hc) * Z * (1.f - Z);
      s_dph[e] = dph;
      s_dpz[e] = dpz;
      s_dh[e] = g * Z;
      s_h[e] = h;
      s_r[e] = R;
      a.dP[row * 3 * O + o] = dpz;
      a.dP[row * 3 * O + 2 * O + o] = dph;
    }
    __syncthreads();
    // through the candidate product: hidden columns (always: they carry d/dR), then the feature columns if wanted
    for (int e = tid; e < nr * O; e += SC_BWD_THREADS) {   // the same (r, j) ownership as above: s_dh[e] is this thread's
      const int r = e / O, j = e - r * O;
      const float* dp = s_dph + r * O;
      const float* w0 = a.Wh + (Fin + j) * O;
      float d = 0.f;
      for (int o = 0; o < O; ++o) d = fmaf(dp[o], w0[o] + w0[CO + o], d);
      const float R = s_r[e];
      const float dpr = d * s_h[e] * R * (1.f - R);
      s_dh[e] = fmaf(d, R, s_dh[e]);
      s_dpr[e] = dpr;
      a.dP[(int64_t)(r0 + r) * 3 * O + O + j] = dpr;
    }
    if (a.dX)
      for (int e = tid; e < nr * Fin; e += SC_BWD_THREADS) {
        const int r = e / Fin, c = e - r * Fin;
        const float* dp = s_dph + r * O;
        const float* w0 = a.Wh + c * O;
        float d = 0.f;
        for (int o = 0; o < O; ++o) d = fmaf(dp[o], w0[o] + w0[CO + o], d);
        s_dx[r * Fin + c] = d;
      }
    __syncthreads();
    if (want_in)
      for (int e = tid; e < nr * C; e += SC_BWD_THREADS) {
        const int r = e / C, c = e - r * C;
        if (c < Fin ? a.dX == nullptr : a.dH == nullptr) continue;
        const float* dz = s_dpz + r * O;
        const float* dr = s_dpr + r * O;
        const float* wz = a.Wz + c * O;
        const float* wr = a.Wr + c * O;
        float d = 0.f;
        for (int o = 0; o < O; ++o) {
          d = fmaf(dz[o], wz[o] + wz[CO + o], d);
          d = fmaf(dr[o], wr[o] + wr[CO + o], d);
        }
        if (c < Fin) a.dX[(int64_t)(r0 + r) * a.lddx + c] = s_dx[r * Fin + c] + d;
        else a.dH[(int64_t)(r0 + r) * a.lddh + (c - Fin)] = s_dh[r * O + (c - Fin)] + d;
      }
    __syncthreads();
  }
  // weight gradients: dW_g[c][o] = sum over rows of S_g[row][c] * dP_g[row][o], S = [X, H] for z and r, [X, R*H] for the
  // candidate; rows in index order.  dP was written by this workgroup (visible after the barrier above).
  for (int e = tid; e < 3 * CO; e += SC_BWD_THREADS) {
    const int g = e / CO, rem = e - g * CO, c = rem / O, o = rem - c * O;
    float acc = 0.f;
    if (c < Fin) {
      for (int row = 0; row < a.N; ++row)
        acc = fmaf(a.X[(int64_t)row * a.ldx + c], a.dP[(int64_t)row * 3 * O + g * O + o], acc);
    } else if (a.H) {
      const int j = c - Fin;
      for (int row = 0; row < a.N; ++row) {
        float s = a.H[(int64_t)row * a.ldh + j];
        if (g == 2) s *= a.saved[(int64_t)row * 3 * O + O + j];
        acc = fmaf(s, a.dP[(int64_t)row * 3 * O + g * O + o], acc);
      }
    }
    float* dW = g == 0 ? a.dWz : (g == 1 ? a.dWr : a.dWh);
    dW[rem] = acc;
    dW[CO + rem] = acc;
  }
  for (int e = tid; e < 3 * O; e += SC_BWD_THREADS) {
    const int g = e / O;
    float* db = g == 0 ? a.dbz : (g == 1 ? a.dbr : a.dbh);
    if (!db) continue;
    float acc = 0.f;
    for (int row = 0; row < a.N; ++row) acc += a.dP[(int64_t)row * 3 * O + e];
    db[e - g * O] = acc;
  }
}

}  // namespace

extern "C" int pgt_dcrnn_cell_k1_fits(int64_t N, int64_t Fin, int64_t O) {
  return N >= 1 && N <= SC_BWD_MAX_ROWS && Fin >= 1 && O >= 1 && O <= SC_MAX_O && Fin + O <= SC_MAX_C;
}

extern "C" int pgt_dcrnn_cell_k1_f32(const float* X, int64_t ldx, const float* H, int64_t ldh, const float* Wz,
                                     const float* Wr, const float* Wh, const float* bz, const float* br, const float* bh,
                                     float* Hnew, int64_t ldo, float* saved, int64_t N, int64_t Fin, int64_t O,
                                     pgt_stream_t stream) {
  PGT_REQUIRE(pgt_dcrnn_cell_k1_fits(N, Fin, O), "pgt_dcrnn_cell_k1_f32: N = %lld, in = %lld, out = %lld outside 1..%d rows, "
              "out <= %d, in + out <= %d", (long long)N, (long long)Fin, (long long)O, SC_BWD_MAX_ROWS, SC_MAX_O, SC_MAX_C);
  PGT_REQUIRE(X && Wz && Wr && Wh && Hnew && saved, "pgt_dcrnn_cell_k1_f32: null pointer");
  PGT_REQUIRE(ldx >= Fin && ldo >= O && (!H || ldh >= O), "pgt_dcrnn_cell_k1_f32: row stride below the row width");
  CellK1Args a{X, ldx, H, ldh, Wz, Wr, Wh, bz, br, bh, Hnew, ldo, saved, (int)N, (int)Fin, (int)O};
  PGT_LAUNCH(cell_k1_fwd_kernel, dim3((unsigned)pgt_cdiv(N, SC_ROWS)), dim3(SC_FWD_THREADS), stream, a);
  return pgt_check_launch("pgt_dcrnn_cell_k1_f32");
}

extern "C" int pgt_dcrnn_cell_k1_bwd_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* H,
                                         int64_t ldh, const float* Wz, const float* Wr, const float* Wh,
                                         const float* saved, float* dX, int64_t lddx, float* dH, int64_t lddh, float* dWz,
                                         float* dWr, float* dWh, float* dbz, float* dbr, float* dbh, float* dP, int64_t N,
                                         int64_t Fin, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(pgt_dcrnn_cell_k1_fits(N, Fin, O), "pgt_dcrnn_cell_k1_bwd_f32: N = %lld, in = %lld, out = %lld outside 1..%d rows, "
              "out <= %d, in + out <= %d", (long long)N, (long long)Fin, (long long)O, SC_BWD_MAX_ROWS, SC_MAX_O, SC_MAX_C);
  PGT_REQUIRE(G && X && Wz && Wr && Wh && saved && dWz && dWr && dWh && dP, "pgt_dcrnn_cell_k1_bwd_f32: null pointer");
  PGT_REQUIRE(ldg >= O && ldx >= Fin && (!H || ldh >= O) && (!dX || lddx >= Fin) && (!dH || lddh >= O),
              "pgt_dcrnn_cell_k1_bwd_f32: row stride below the row width");
  CellK1BwdArgs a{G, ldg, X, ldx, H, ldh, Wz, Wr, Wh, saved, dX, lddx, dH, lddh, dWz, dWr, dWh, dbz, dbr, dbh, dP,
                  (int)N, (int)Fin, (int)O};
  PGT_LAUNCH(cell_k1_bwd_kernel, dim3(1), dim3(SC_BWD_THREADS), stream, a);
  return pgt_check_launch("pgt_dcrnn_cell_k1_bwd_f32");
}
