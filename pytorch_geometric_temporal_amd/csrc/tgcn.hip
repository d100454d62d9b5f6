// T-GCN cell (torch_geometric_temporal/nn/recurrent/temporalgcn.py:82-130) parameter plumbing.
//
// A gate of the reference is  linear_g(cat[conv_g(X), H'])  with conv_g(X) = A_hat (X Wc_g^T) + bc_g (PyG GCNConv) and
// H' = H (z, r) or H * R (candidate).  Both maps are linear, so with AX = A_hat X (ONE aggregation at the INPUT width,
// shared by the three gates)
//     pre_g = [AX | H'] W'_g + b'_g,   W'_g = [ Wc_g^T L_g[:, :O]^T ; L_g[:, O:]^T ]  (Fin + O rows),   b'_g = lb_g + L_g[:, :O] bc_g
// and the cell is exactly the two gate products of a GRU on the operand [AX | H'] — the fused-epilogue entry points
// pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32 — instead of the reference's 3 aggregations at width O, 3 + 3 products, 2
// concatenations and ~8 elementwise passes per step.  These two kernels build (W'_zr, b'_zr, W'_h, b'_h) from the module's
// parameters and carry the gradients back, ONE launch each way (torch's slice / matmul / cat graph of the same arithmetic is
// ~25 tiny launches per step of the sequence loop and twice that backward).
#include "pgt_common.h"

namespace {

struct TgcnParams {
  const float* Wc[3];   // conv_{z,r,h}.lin.weight  [O, Fin]
  const float* bc[3];   // conv_{z,r,h}.bias        [O] | null
  const float* L[3];    // linear_{z,r,h}.weight    [O, 2 O]
  const float* lb[3];   // linear_{z,r,h}.bias      [O] | null
};
struct TgcnGrads {
  float* dWc[3]; float* dbc[3]; float* dL[3]; float* dlb[3];
};

__global__ __launch_bounds__(256) void tgcn_pack_kernel(TgcnParams p, int Fin, int O, float* __restrict__ Wzr,
                                                        float* __restrict__ bzr, float* __restrict__ Wh, float* __restrict__ bh) {
  const int C = Fin + O;
  const int64_t nW = (int64_t)C * 3 * O;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nW + 3 * O) return;
  const int col = (int)(e < nW ? e % (3 * O) : e - nW);
  const int gate = col / O, o = col - gate * O;
  const float* L = p.L[gate] + (int64_t)o * 2 * O;
  float v;
  if (e < nW) {
    const int c = (int)(e / (3 * O));
    if (c < Fin) {
      v = 0.f;
#pragma unroll 8
      for (int j = 0; j < O; ++j) v = fmaf(p.Wc[gate][(int64_t)j * Fin + c], L[j], v);
    } else {
      v = L[O + (c - Fin)];
    }
    if (gate < 2) Wzr[(int64_t)c * 2 * O + gate * O + o] = v;
    else Wh[(int64_t)c * O + o] = v;
  } else {
    v = p.lb[gate] ? p.lb[gate][o] : 0.f;
    if (p.bc[gate]) {
#pragma unroll 8
      for (int j = 0; j < O; ++j) v = fmaf(p.bc[gate][j], L[j], v);
    }
    if (gate < 2) bzr[gate * O + o] = v;
    else bh[o] = v;
  }
}

// adjoint of the packing: every output element is one thread's sum in index order (deterministic)
__global__ __launch_bounds__(256) void tgcn_unpack_kernel(TgcnParams p, TgcnGrads d, int Fin, int O,
                                                          const float* __restrict__ dWzr, const float* __restrict__ dbzr,
                                                          const float* __restrict__ dWh, const float* __restrict__ dbh) {
  const int64_t nWc = (int64_t)O * Fin, nL = (int64_t)O * 2 * O;
  const int64_t per = nWc + O + nL + O;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= 3 * per) return;
  const int gate = (int)(e / per);
  const int64_t r = e - gate * per;
  auto dW = [&](int c, int o) { return gate < 2 ? dWzr[(int64_t)c * 2 * O + gate * O + o] : dWh[(int64_t)c * O + o]; };
  auto db = [&](int o) { return gate < 2 ? dbzr[gate * O + o] : dbh[o]; };
  const float* L = p.L[gate];
  if (r < nWc) {                                            // dWc[j, f] = sum_o dW'[f, o] L[o, j]
    const int j = (int)(r / Fin), f = (int)(r - (int64_t)j * Fin);
    float v = 0.f;
#pragma unroll 8
    for (int o = 0; o < O; ++o) v = fmaf(dW(f, o), L[(int64_t)o * 2 * O + j], v);
    d.dWc[gate][r] = v;
  } else if (r < nWc + O) {                                 // dbc[j] = sum_o db'[o] L[o, j]
    const int j = (int)(r - nWc);
    if (d.dbc[gate]) {
      float v = 0.f;
#pragma unroll 8
      for (int o = 0; o < O; ++o) v = fmaf(db(o), L[(int64_t)o * 2 * O + j], v);
      d.dbc[gate][j] = v;
    }
  } else if (r < nWc + O + nL) {
    const int64_t q = r - nWc - O;
    const int o = (int)(q / (2 * O)), j = (int)(q - (int64_t)o * 2 * O);
    float v;
    if (j < O) {                                            // dL[o, j] = sum_f dW'[f, o] Wc[j, f] + db'[o] bc[j]
      v = p.bc[gate] ? db(o) * p.bc[gate][j] : 0.f;
      for (int f = 0; f < Fin; ++f) v = fmaf(dW(f, o), p.Wc[gate][(int64_t)j * Fin + f], v);
    } else {
      v = dW(Fin + (j - O), o);
    }
    d.dL[gate][q] = v;
  } else {
    const int o = (int)(r - nWc - O - nL);
    if (d.dlb[gate]) d.dlb[gate][o] = db(o);
  }
}

}  // namespace

extern "C" int pgt_tgcn_pack_weights_f32(const float* const Wc[3], const float* const bc[3], const float* const L[3],
                                         const float* const lb[3], int64_t Fin, int64_t O, float* Wzr, float* bzr, float* Wh,
                                         float* bh, pgt_stream_t stream) {
  PGT_REQUIRE(Fin >= 1 && O >= 1 && Fin < (1 << 20) && O < (1 << 14), "pgt_tgcn_pack_weights_f32: bad size");
  PGT_REQUIRE(Wc && bc && L && lb && Wzr && bzr && Wh && bh, "pgt_tgcn_pack_weights_f32: null pointer");
  TgcnParams p;
  for (int g = 0; g < 3; ++g) {
    PGT_REQUIRE(Wc[g] && L[g], "pgt_tgcn_pack_weights_f32: null weight");
    p.Wc[g] = Wc[g]; p.bc[g] = bc[g]; p.L[g] = L[g]; p.lb[g] = lb[g];
  }
  const int64_t total = (Fin + O) * 3 * O + 3 * O;
  dim3 grid((unsigned)pgt_cdiv(total, 256));
  PGT_LAUNCH(tgcn_pack_kernel, grid, dim3(256), stream, p, (int)Fin, (int)O, Wzr, bzr, Wh, bh);
  return pgt_check_launch("pgt_tgcn_pack_weights_f32");
}

extern "C" int pgt_tgcn_unpack_weight_grads_f32(const float* dWzr, const float* dbzr, const float* dWh, const float* dbh,
                                                const float* const Wc[3], const float* const bc[3], const float* const L[3],
                                                int64_t Fin, int64_t O, float* const dWc[3], float* const dbc[3],
                                                float* const dL[3], float* const dlb[3], pgt_stream_t stream) {
  PGT_REQUIRE(Fin >= 1 && O >= 1 && Fin < (1 << 20) && O < (1 << 14), "pgt_tgcn_unpack_weight_grads_f32: bad size");
  PGT_REQUIRE(dWzr && dbzr && dWh && dbh && Wc && bc && L && dWc && dbc && dL && dlb,
              "pgt_tgcn_unpack_weight_grads_f32: null pointer");
  TgcnParams p;
  TgcnGrads d;
  for (int g = 0; g < 3; ++g) {
    PGT_REQUIRE(Wc[g] && L[g] && dWc[g] && dL[g], "pgt_tgcn_unpack_weight_grads_f32: null weight");
    p.Wc[g] = Wc[g]; p.bc[g] = bc[g]; p.L[g] = L[g]; p.lb[g] = nullptr;
    d.dWc[g] = dWc[g]; d.dbc[g] = dbc[g]; d.dL[g] = dL[g]; d.dlb[g] = dlb[g];
  }
  const int64_t total = 3 * (O * Fin + O + O * 2 * O + O);
  dim3 grid((unsigned)pgt_cdiv(total, 256));
  PGT_LAUNCH(tgcn_unpack_kernel, grid, dim3(256), stream, p, d, (int)Fin, (int)O, dWzr, dbzr, dWh, dbh);
  return pgt_check_launch("pgt_tgcn_unpack_weight_grads_f32");
}
