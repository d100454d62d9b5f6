// GRU gate chains (dcrnn.py:172-192 / temporalgcn.py:82-102) fused into single passes, their backward twins,
// and the small strided movers that replace torch.cat / permute on the path.  All HBM-streaming: each lane moves
// V consecutive floats (V = 4/2/1 picked from pointer + stride alignment), consecutive lanes on consecutive V-groups.
#include "pgt_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return pgt_sigmoidf(x); }

template <int V>
__global__ __launch_bounds__(256) void gru_zr_kernel(float* pre_zr, const float* __restrict__ H, int64_t ldh,
                                                      float* xhr, int64_t ldxhr, int f_in, int64_t M, int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float* p = pre_zr + m * 2 * O;
  float z[V], r[V], h[V], hr[V];
  pgt_ldv<V>(p + o, z);
  pgt_ldv<V>(p + O + o, r);
  pgt_ldv<V>(H + m * ldh + o, h);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    z[i] = sigmoidf_(z[i]);
    r[i] = sigmoidf_(r[i]);
    hr[i] = h[i] * r[i];
  }
  pgt_stv<V>(p + o, z);
  pgt_stv<V>(p + O + o, r);
  pgt_stv<V>(xhr + m * ldxhr + f_in + o, hr);
}

template <int V>
__global__ __launch_bounds__(256) void gru_h_kernel(float* pre_h, const float* __restrict__ zr,
                                                     const float* __restrict__ H, int64_t ldh, float* out0,
                                                     int64_t ld0, pgt_rowmap map0, float* out1, int64_t ld1, int64_t M,
                                                     int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float t[V], z[V], h[V], hn[V];
  pgt_ldv<V>(pre_h + m * O + o, t);
  pgt_ldv<V>(zr + m * 2 * O + o, z);
  pgt_ldv<V>(H + m * ldh + o, h);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    t[i] = tanhf(t[i]);
    hn[i] = pgt_gru_blend(z[i], h[i], t[i]);
  }
  pgt_stv<V>(pre_h + m * O + o, t);
  pgt_stv<V>(out0 + pgt_row_off(m, ld0, map0.period, map0.stride_hi) + o, hn);
  if (out1) pgt_stv<V>(out1 + m * ld1 + o, hn);
}

template <int V>
__global__ __launch_bounds__(256) void gru_h_bwd_kernel(const float* __restrict__ dHn, int64_t lddh, pgt_rowmap map_dh,
                                                         const float* dHn2, int64_t lddh2,
                                                         const float* __restrict__ dHn3, int64_t lddh3,
                                                         const float* __restrict__ zr,
                                                         const float* __restrict__ H, int64_t ldh, pgt_rowmap map_h,
                                                         const float* __restrict__ ht, float* d_pre_h,
                                                         float* d_pre_zr, float* dH, int64_t lddhp, int acc,
                                                         int64_t M, int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float g[V], g2[V], z[V], h[V], t[V], dph[V], dpz[V], dh[V];
  pgt_ldv<V>(dHn + pgt_row_off(m, lddh, map_dh.period, map_dh.stride_hi) + o, g);
  if (dHn2) {
    pgt_ldv<V>(dHn2 + m * lddh2 + o, g2);
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] += g2[i];
  }
  if (dHn3) {
    pgt_ldv<V>(dHn3 + m * lddh3 + o, g2);
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] += g2[i];
  }
  pgt_ldv<V>(zr + m * 2 * O + o, z);
  pgt_ldv<V>(H + pgt_row_off(m, ldh, map_h.period, map_h.stride_hi) + o, h);
  pgt_ldv<V>(ht + m * O + o, t);
  float* q = dH + m * lddhp + o;
  if (acc) pgt_ldv<V>(q, dh);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    dph[i] = g[i] * (1.f - z[i]) * (1.f - t[i] * t[i]);
    dpz[i] = g[i] * (h[i] - t[i]) * z[i] * (1.f - z[i]);
    dh[i] = acc ? dh[i] + g[i] * z[i] : g[i] * z[i];
  }
  pgt_stv<V>(d_pre_h + m * O + o, dph);
  pgt_stv<V>(d_pre_zr + m * 2 * O + o, dpz);
  pgt_stv<V>(q, dh);
}

template <int V>
__global__ __launch_bounds__(256) void gru_zr_bwd_kernel(const float* __restrict__ dxhr, int64_t lddxhr, int f_in,
                                                          const float* __restrict__ zr,
                                                          const float* __restrict__ H, int64_t ldh, pgt_rowmap map_h,
                                                          float* d_pre_zr, float* dH, int64_t lddhp, int64_t M,
                                                          int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float g[V], r[V], h[V], dpr[V], dh[V];
  pgt_ldv<V>(dxhr + m * lddxhr + f_in + o, g);
  pgt_ldv<V>(zr + m * 2 * O + O + o, r);
  pgt_ldv<V>(H + pgt_row_off(m, ldh, map_h.period, map_h.stride_hi) + o, h);
  float* q = dH + m * lddhp + o;
  pgt_ldv<V>(q, dh);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    dpr[i] = g[i] * h[i] * r[i] * (1.f - r[i]);
    dh[i] += g[i] * r[i];
  }
  pgt_stv<V>(d_pre_zr + m * 2 * O + O + o, dpr);
  pgt_stv<V>(q, dh);
}

// ---- peephole LSTM gate chain (gconv_lstm.py:138-172; gc_lstm.py:138-169 with wci = wcf = wco = NULL)
//   P [M, 4O] holds the pre-activations i | f | c | o (every bias already folded in by the GEMM that produced it).
//   I = s(P_i + wci*C)  F = s(P_f + wcf*C)  T = tanh(P_c)  C' = F*C + I*T  Og = s(P_o + wco*C')  H = Og*tanh(C')
//   The activated gates overwrite P (saved for the backward pass).
template <int V>
__global__ __launch_bounds__(256) void lstm_gates_kernel(float* P, const float* __restrict__ C, int64_t ldc,
                                                          const float* __restrict__ wci,
                                                          const float* __restrict__ wcf,
                                                          const float* __restrict__ wco, float* Hn, int64_t ldh,
                                                          float* Cn, int64_t ldcn, int64_t M, int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float* p = P + m * 4 * O + o;
  float gi[V], gf[V], gt[V], go[V], c[V], hn[V], cn[V], pi[V], pf[V], po[V];
  pgt_ldv<V>(p, gi);
  pgt_ldv<V>(p + O, gf);
  pgt_ldv<V>(p + 2 * O, gt);
  pgt_ldv<V>(p + 3 * O, go);
  pgt_ldv<V>(C + m * ldc + o, c);
#pragma unroll
  for (int i = 0; i < V; ++i) { pi[i] = wci ? wci[o + i] : 0.f; pf[i] = wcf ? wcf[o + i] : 0.f; po[i] = wco ? wco[o + i] : 0.f; }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    gi[i] = sigmoidf_(gi[i] + pi[i] * c[i]);
    gf[i] = sigmoidf_(gf[i] + pf[i] * c[i]);
    gt[i] = tanhf(gt[i]);
    cn[i] = gf[i] * c[i] + gi[i] * gt[i];
    go[i] = sigmoidf_(go[i] + po[i] * cn[i]);
    hn[i] = go[i] * tanhf(cn[i]);
  }
  pgt_stv<V>(p, gi);
  pgt_stv<V>(p + O, gf);
  pgt_stv<V>(p + 2 * O, gt);
  pgt_stv<V>(p + 3 * O, go);
  pgt_stv<V>(Cn + m * ldcn + o, cn);
  pgt_stv<V>(Hn + m * ldh + o, hn);
}

// backward: gates [M,4O] (activated i|f|t|o), C (previous cell), Cn (new cell), dH, dCn (may be NULL) ->
//   dP [M,4O] (pre-activation gradients), dC [M,O] (w.r.t. the previous cell), dw [3,O] += peephole gradients
//   (column sums: one wavefront-partial atomicAdd per column group).
template <int V>
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ G, const float* __restrict__ C,
                                                              int64_t ldc, const float* __restrict__ Cn,
                                                              int64_t ldcn, const float* __restrict__ wci,
                                                              const float* __restrict__ wcf,
                                                              const float* __restrict__ wco,
                                                              const float* __restrict__ dH, int64_t lddh,
                                                              const float* dCn, int64_t lddcn, float* dP, float* dC,
                                                              int64_t lddc, float* dw, int64_t M, int O,
                                                              int rows_per_block) {
  // block = (column group = threadIdx.x % OV ... ) : threads are laid out [rows][OV] so a column's partial sums meet in
  // one thread column; rows_per_block rows are walked by each thread row.
  const int OV = O / V;
  const int tcol = threadIdx.x % OV, trow = threadIdx.x / OV, nrow_t = 256 / OV;
  const int o = tcol * V;
  float sw_i[V], sw_f[V], sw_o[V];
#pragma unroll
  for (int i = 0; i < V; ++i) sw_i[i] = sw_f[i] = sw_o[i] = 0.f;
  float pi[V], pf[V], po[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { pi[i] = wci ? wci[o + i] : 0.f; pf[i] = wcf ? wcf[o + i] : 0.f; po[i] = wco ? wco[o + i] : 0.f; }
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  if (trow < nrow_t) {
    for (int64_t m = m0 + trow; m < m0 + rows_per_block && m < M; m += nrow_t) {
      const float* gp = G + m * 4 * O + o;
      float gi[V], gf[V], gt[V], go[V], c[V], cn[V], dh[V], dcn[V], dpi[V], dpf[V], dpt[V], dpo[V], dc[V];
      pgt_ldv<V>(gp, gi);
      pgt_ldv<V>(gp + O, gf);
      pgt_ldv<V>(gp + 2 * O, gt);
      pgt_ldv<V>(gp + 3 * O, go);
      pgt_ldv<V>(C + m * ldc + o, c);
      pgt_ldv<V>(Cn + m * ldcn + o, cn);
      pgt_ldv<V>(dH + m * lddh + o, dh);
      if (dCn) pgt_ldv<V>(dCn + m * lddcn + o, dcn);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float tc = tanhf(cn[i]);
        dpo[i] = dh[i] * tc * go[i] * (1.f - go[i]);
        const float dct = (dCn ? dcn[i] : 0.f) + dh[i] * go[i] * (1.f - tc * tc) + dpo[i] * po[i];
        dpi[i] = dct * gt[i] * gi[i] * (1.f - gi[i]);
        dpf[i] = dct * c[i] * gf[i] * (1.f - gf[i]);
        dpt[i] = dct * gi[i] * (1.f - gt[i] * gt[i]);
        dc[i] = dct * gf[i] + dpi[i] * pi[i] + dpf[i] * pf[i];
        sw_i[i] += dpi[i] * c[i];
        sw_f[i] += dpf[i] * c[i];
        sw_o[i] += dpo[i] * cn[i];
      }
      float* dp = dP + m * 4 * O + o;
      pgt_stv<V>(dp, dpi);
      pgt_stv<V>(dp + O, dpf);
      pgt_stv<V>(dp + 2 * O, dpt);
      pgt_stv<V>(dp + 3 * O, dpo);
      pgt_stv<V>(dC + m * lddc + o, dc);
    }
  }
  if (dw != nullptr && trow < nrow_t) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (wci) atomicAdd(dw + o + i, sw_i[i]);
      if (wcf) atomicAdd(dw + O + o + i, sw_f[i]);
      if (wco) atomicAdd(dw + 2 * O + o + i, sw_o[i]);
    }
  }
}

// mode 0: dst = x ; 1: dst += x ; 2: dst = a*x + b*y
template <int V>
__global__ __launch_bounds__(256) void mover2d_kernel(float* dst, int64_t ldd, const float* __restrict__ x,
                                                       int64_t ldx, float a, const float* y, int64_t ldy, float b,
                                                       int64_t M, int64_t W, int mode) {
  const int64_t WV = W / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * WV) return;
  const int64_t m = idx / WV;
  const int64_t w = (idx - m * WV) * V;
  float xv[V], dv[V], yv[V];
  pgt_ldv<V>(x + m * ldx + w, xv);
  float* d = dst + m * ldd + w;
  if (mode == 1) {
    pgt_ldv<V>(d, dv);
#pragma unroll
    for (int i = 0; i < V; ++i) dv[i] += xv[i];
  } else if (mode == 2) {
    if (y) {
      pgt_ldv<V>(y + m * ldy + w, yv);
#pragma unroll
      for (int i = 0; i < V; ++i) dv[i] = a * xv[i] + b * yv[i];
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) dv[i] = a * xv[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) dv[i] = xv[i];
  }
  pgt_stv<V>(d, dv);
}

template <int V>
__global__ __launch_bounds__(256) void swap01_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                      int64_t D0, int64_t D1, int64_t W) {
  const int64_t WV = W / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= D0 * D1 * WV) return;
  const int64_t w = (idx % WV) * V;
  const int64_t d1 = (idx / WV) % D1;
  const int64_t d0 = idx / (WV * D1);
  float v[V];
  pgt_ldv<V>(src + (d0 * D1 + d1) * W + w, v);
  pgt_stv<V>(dst + (d1 * D0 + d0) * W + w, v);
}


// Index-batch window gather (signal/index_dataset.py:32-57: x = data[idx : idx + h], y = data[idx + h : idx + 2 h] for
// every sample of the batch) from the HBM-resident series in ONE launch: both windows of all B samples, optionally
// written time-major ([h][B][W] instead of [B][h][W]) — the row order the recurrent layers consume, which saves the
// [B, T] -> [T, B] transposition that would follow.
template <int V>
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ data, const int64_t* __restrict__ idx,
                                                             int64_t B, int64_t h, int64_t W, int64_t T_total,
                                                             float* __restrict__ X, float* __restrict__ Y, int time_major) {
  const int64_t WV = W / V;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * B * h * WV) return;
  const int64_t w = (e % WV) * V;
  const int64_t t = (e / WV) % h;
  const int64_t b = (e / (WV * h)) % B;
  const int which = (int)(e / (WV * h * B));         // 0: the input window, 1: the target window
  int64_t row = idx[b] + t + (which ? h : 0);
  row = row < 0 ? 0 : (row < T_total ? row : T_total - 1);   // never read outside the series: the CALLER guarantees the range (IndexDataset.gather checks it once)
  float v[V];
  pgt_ldv<V>(data + row * W + w, v);
  float* out = which ? Y : X;
  pgt_stv<V>(out + (time_major ? (t * B + b) : (b * h + t)) * W + w, v);
}

// ---- DCRNN cell weights (dcrnn.py:26-37, :138-160): the three DConv weights [2, K, C, O] -> the stacked operands of the
// two gate products, [(2K-1) C, 2O] (z | r) and [(2K-1) C, O] (candidate); segment 0 = W[0,0] + W[1,0] (the reference
// computes X W[0,0] + X W[1,0], dcrnn.py:81-83), segment 2k-1+d = W[d,k]; the z | r bias is the concatenation.  ONE launch
// each way instead of the ~8 / ~10 torch slice / add / cat launches a forward / backward pass paid for it — which is
// what a per-snapshot loop over a small graph (Chickenpox: 20 nodes) spends its time on.
__global__ __launch_bounds__(256) void dcrnn_pack_weights_kernel(const float* __restrict__ Wz, const float* __restrict__ Wr,
                                                                 const float* __restrict__ Wh, const float* __restrict__ bz,
                                                                 const float* __restrict__ br, int K, int C, int O,
                                                                 float* __restrict__ Wzr, float* __restrict__ bzr,
                                                                 float* __restrict__ Whs) {
  const int S = 2 * K - 1;
  const int64_t total = (int64_t)S * C * 3 * O;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < 2 * O && bzr != nullptr) bzr[e] = e < O ? bz[e] : br[e - O];
  if (e >= total) return;
  const int o3 = (int)(e % (3 * O));
  const int64_t sc = e / (3 * O);
  const int c = (int)(sc % C), seg = (int)(sc / C);
  const int gate = o3 / O, o = o3 - gate * O;                       // 0 = z, 1 = r, 2 = candidate
  const float* W = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);          // [2][K][C][O]
  const int64_t KCO = (int64_t)K * C * O;
  float v;
  if (seg == 0) v = W[(int64_t)c * O + o] + W[KCO + (int64_t)c * O + o];
  else {
    const int k = (seg + 1) >> 1, d = (seg + 1) & 1;                // seg = 2k - 1 + d
    v = W[d * KCO + ((int64_t)k * C + c) * O + o];
  }
  if (gate < 2) Wzr[((int64_t)seg * C + c) * 2 * O + gate * O + o] = v;
  else Whs[((int64_t)seg * C + c) * O + o] = v;
}
// adjoint: dW[0,0] = dW[1,0] = d segment 0; dW[d,k] = d segment 2k-1+d; dbz | dbr = the halves of dbzr
__global__ __launch_bounds__(256) void dcrnn_unpack_grads_kernel(const float* __restrict__ dWzr, const float* __restrict__ dbzr,
                                                                 const float* __restrict__ dWhs, int K, int C, int O,
                                                                 float* __restrict__ dWz, float* __restrict__ dWr,
                                                                 float* __restrict__ dWh, float* dbz, float* dbr) {
  const int64_t KCO = (int64_t)K * C * O;
  const int64_t total = 3 * 2 * KCO;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < 2 * O && dbzr != nullptr) { if (e < O) dbz[e] = dbzr[e]; else dbr[e - O] = dbzr[e]; }
  if (e >= total) return;
  const int gate = (int)(e / (2 * KCO));
  const int64_t r = e - gate * 2 * KCO;
  const int d = (int)(r / KCO);
  const int64_t kco = r - d * KCO;
  const int o = (int)(kco % O), c = (int)((kco / O) % C), k = (int)(kco / ((int64_t)C * O));
  const int seg = k == 0 ? 0 : 2 * k - 1 + d;
  float v;
  if (gate < 2) v = dWzr != nullptr ? dWzr[((int64_t)seg * C + c) * 2 * O + gate * O + o] : 0.f;
  else v = dWhs != nullptr ? dWhs[((int64_t)seg * C + c) * O + o] : 0.f;
  (gate == 0 ? dWz : (gate == 1 ? dWr : dWh))[e - gate * 2 * KCO] = v;
}

// ---- DCRNN sequence inputs: the input columns of segment 0 of both stacks for all T steps and the initial state into
// step 0 of the gate stack — one launch for what was three strided copies
__global__ __launch_bounds__(256) void dcrnn_stage_kernel(const float* __restrict__ X, const float* __restrict__ H0, int64_t TM,
                                                          int64_t M, int Fin, int O, float* __restrict__ TSzr0,
                                                          float* __restrict__ TSh0) {
  const int C = Fin + O;
  const int64_t nx = TM * Fin, total = nx + M * O;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  if (e < nx) {
    const int64_t m = e / Fin;
    const int f = (int)(e - m * Fin);
    const float v = X[e];
    TSzr0[m * C + f] = v;
    TSh0[m * C + f] = v;
  } else {
    const int64_t q = e - nx, m = q / O;
    TSzr0[m * C + Fin + (int)(q - m * O)] = H0[q];
  }
}

inline int grid_for(int64_t total, const char* what, dim3* grid) {
  const int64_t nb = pgt_cdiv(total, 256);
  if (nb >= ((int64_t)1 << 31)) {
    pgt_set_error("%s: grid too large", what);
    return PGT_ERR_INVALID;
  }
  *grid = dim3((unsigned)nb);
  return PGT_OK;
}

#define PGT_VDISPATCH(v, KERN, grid, block, stream, ...)                       \
  do {                                                                         \
    if ((v) == 4) PGT_LAUNCH((KERN<4>), grid, block, stream, __VA_ARGS__);     \
    else if ((v) == 2) PGT_LAUNCH((KERN<2>), grid, block, stream, __VA_ARGS__); \
    else PGT_LAUNCH((KERN<1>), grid, block, stream, __VA_ARGS__);              \
  } while (0)

}  // namespace

extern "C" int pgt_gru_zr_f32(float* pre_zr, const float* H, int64_t ldh, float* xhr, int64_t ldxhr,
                              int64_t f_in, int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0 && f_in >= 0, "pgt_gru_zr_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(pre_zr && H && xhr, "pgt_gru_zr_f32: null pointer");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(pre_zr, 2 * O); pick.operand(pre_zr + O, 2 * O); pick.operand(H, ldh); pick.operand(xhr + f_in, ldxhr);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_zr_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_zr_kernel, grid, block, stream, pre_zr, H, ldh, xhr, ldxhr, (int)f_in, M, (int)O);
  return pgt_check_launch("pgt_gru_zr_f32");
}

extern "C" int pgt_gru_h_f32(float* pre_h, const float* zr, const float* H, int64_t ldh, float* out0,
                             int64_t ld0, const pgt_rowmap* map0, float* out1, int64_t ld1, int64_t M, int64_t O,
                             pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(pre_h && zr && H && out0, "pgt_gru_h_f32: null pointer");
  pgt_rowmap m0;
  PGT_REQUIRE(pgt_rowmap_take(map0, M, &m0), "pgt_gru_h_f32: row map out of range");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(pre_h, O); pick.operand(zr, 2 * O); pick.operand(H, ldh); pick.operand(out0, ld0, m0);
  pick.operand(out1, ld1);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_h_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_h_kernel, grid, block, stream, pre_h, zr, H, ldh, out0, ld0, m0, out1, ld1, M, (int)O);
  return pgt_check_launch("pgt_gru_h_f32");
}

extern "C" int pgt_gru_h_bwd_f32(const float* dHnew, int64_t lddh, const pgt_rowmap* map_dh, const float* dHnew2,
                                 int64_t lddh2, const float* dHnew3, int64_t lddh3, const float* zr, const float* H,
                                 int64_t ldh, const pgt_rowmap* map_h,
                                 const float* ht, float* d_pre_h, float* d_pre_zr, float* dH, int64_t lddhp,
                                 int accumulate_dh, int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dHnew && zr && H && ht && d_pre_h && d_pre_zr && dH, "pgt_gru_h_bwd_f32: null pointer");
  pgt_rowmap mg, mh;
  PGT_REQUIRE(pgt_rowmap_take(map_dh, M, &mg) && pgt_rowmap_take(map_h, M, &mh), "pgt_gru_h_bwd_f32: row map out of range");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(dHnew, lddh, mg); pick.operand(dHnew2, lddh2); pick.operand(dHnew3, lddh3); pick.operand(zr, 2 * O);
  pick.operand(H, ldh, mh); pick.operand(ht, O); pick.operand(d_pre_h, O); pick.operand(d_pre_zr, 2 * O);
  pick.operand(dH, lddhp);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_h_bwd_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_h_bwd_kernel, grid, block, stream, dHnew, lddh, mg, dHnew2, lddh2, dHnew3, lddh3, zr, H, ldh,
                mh, ht, d_pre_h, d_pre_zr, dH, lddhp, accumulate_dh, M, (int)O);
  return pgt_check_launch("pgt_gru_h_bwd_f32");
}

extern "C" int pgt_gru_zr_bwd_f32(const float* dxhr, int64_t lddxhr, int64_t f_in, const float* zr,
                                  const float* H, int64_t ldh, const pgt_rowmap* map_h, float* d_pre_zr, float* dH,
                                  int64_t lddhp, int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0 && f_in >= 0, "pgt_gru_zr_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dxhr && zr && H && d_pre_zr && dH, "pgt_gru_zr_bwd_f32: null pointer");
  pgt_rowmap mh;
  PGT_REQUIRE(pgt_rowmap_take(map_h, M, &mh), "pgt_gru_zr_bwd_f32: row map out of range");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(dxhr + f_in, lddxhr); pick.operand(zr + O, 2 * O); pick.operand(H, ldh, mh);
  pick.operand(d_pre_zr + O, 2 * O); pick.operand(dH, lddhp);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_zr_bwd_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_zr_bwd_kernel, grid, block, stream, dxhr, lddxhr, (int)f_in, zr, H, ldh, mh, d_pre_zr, dH,
                lddhp, M, (int)O);
  return pgt_check_launch("pgt_gru_zr_bwd_f32");
}

static int mover(const char* what, float* dst, int64_t ldd, const float* x, int64_t ldx, float a, const float* y,
                 int64_t ldy, float b, int64_t M, int64_t W, int mode, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && W >= 0, "%s: negative size", what);
  if (M == 0 || W == 0) return PGT_OK;
  PGT_REQUIRE(dst && x, "%s: null pointer", what);
  PgtVecPick pick;
  pick.width(W);
  pick.operand(dst, ldd); pick.operand(x, ldx); pick.operand(y, ldy);
  dim3 grid, block(256);
  if (int e = grid_for(M * (W / pick.v), what, &grid)) return e;
  PGT_VDISPATCH(pick.v, mover2d_kernel, grid, block, stream, dst, ldd, x, ldx, a, y, ldy, b, M, W, mode);
  return pgt_check_launch(what);
}

extern "C" int pgt_lstm_gates_f32(float* P, const float* C, int64_t ldc, const float* wci, const float* wcf,
                                  const float* wco, float* Hn, int64_t ldh, float* Cn, int64_t ldcn, int64_t M,
                                  int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_lstm_gates_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(P && C && Hn && Cn, "pgt_lstm_gates_f32: null pointer");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(P, 4 * O); pick.operand(P + O, 4 * O); pick.operand(C, ldc); pick.operand(Hn, ldh); pick.operand(Cn, ldcn);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_lstm_gates_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, lstm_gates_kernel, grid, block, stream, P, C, ldc, wci, wcf, wco, Hn, ldh, Cn, ldcn, M, (int)O);
  return pgt_check_launch("pgt_lstm_gates_f32");
}

extern "C" int pgt_lstm_gates_bwd_f32(const float* gates, const float* C, int64_t ldc, const float* Cn,
                                      int64_t ldcn, const float* wci, const float* wcf, const float* wco,
                                      const float* dH, int64_t lddh, const float* dCn, int64_t lddcn, float* dP,
                                      float* dC, int64_t lddc, float* dw, int64_t M, int64_t O,
                                      pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_lstm_gates_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(gates && C && Cn && dH && dP && dC, "pgt_lstm_gates_bwd_f32: null pointer");
  PGT_REQUIRE(!(wci || wcf || wco) || dw, "pgt_lstm_gates_bwd_f32: peephole weights given but dw is null");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(gates, 4 * O); pick.operand(gates + O, 4 * O); pick.operand(C, ldc); pick.operand(Cn, ldcn);
  pick.operand(dH, lddh); pick.operand(dCn, lddcn); pick.operand(dP, 4 * O); pick.operand(dP + O, 4 * O);
  pick.operand(dC, lddc);
  const int v = pick.v;
  PGT_REQUIRE(O / v <= 256, "pgt_lstm_gates_bwd_f32: more than 1024 channels are not supported");
  const int64_t nrow_t = 256 / (O / v);
  // ~2048 workgroups; each thread row walks rows_per_block / nrow_t rows
  int64_t rpb = pgt_cdiv(pgt_cdiv(M, 2048), nrow_t) * nrow_t;
  if (rpb < nrow_t) rpb = nrow_t;
  const int64_t nblk = pgt_cdiv(M, rpb);
  PGT_REQUIRE(nblk < ((int64_t)1 << 31), "pgt_lstm_gates_bwd_f32: grid too large");
  dim3 grid((unsigned)nblk), block(256);
  PGT_VDISPATCH(v, lstm_gates_bwd_kernel, grid, block, stream, gates, C, ldc, Cn, ldcn, wci, wcf, wco, dH, lddh, dCn,
                lddcn, dP, dC, lddc, dw, M, (int)O, (int)rpb);
  return pgt_check_launch("pgt_lstm_gates_bwd_f32");
}

extern "C" int pgt_dcrnn_pack_weights_f32(const float* Wz, const float* Wr, const float* Wh, const float* bz, const float* br,
                                          int64_t K, int64_t C, int64_t O, float* Wzr, float* bzr, float* Whs,
                                          pgt_stream_t stream) {
  PGT_REQUIRE(K >= 1 && C >= 1 && O >= 1, "pgt_dcrnn_pack_weights_f32: bad size");
  PGT_REQUIRE(Wz && Wr && Wh && Wzr && Whs, "pgt_dcrnn_pack_weights_f32: null pointer");
  PGT_REQUIRE((bzr == nullptr) == (bz == nullptr) && (bz == nullptr) == (br == nullptr),
              "pgt_dcrnn_pack_weights_f32: the gate biases go together");
  dim3 grid;
  const int64_t total = (2 * K - 1) * C * 3 * O;
  if (int e = grid_for(total > 2 * O ? total : 2 * O, "pgt_dcrnn_pack_weights_f32", &grid)) return e;
  PGT_LAUNCH(dcrnn_pack_weights_kernel, grid, dim3(256), stream, Wz, Wr, Wh, bz, br, (int)K, (int)C, (int)O, Wzr, bzr, Whs);
  return pgt_check_launch("pgt_dcrnn_pack_weights_f32");
}

extern "C" int pgt_dcrnn_unpack_weight_grads_f32(const float* dWzr, const float* dbzr, const float* dWhs, int64_t K, int64_t C,
                                                 int64_t O, float* dWz, float* dWr, float* dWh, float* dbz, float* dbr,
                                                 pgt_stream_t stream) {
  PGT_REQUIRE(K >= 1 && C >= 1 && O >= 1, "pgt_dcrnn_unpack_weight_grads_f32: bad size");
  PGT_REQUIRE(dWz && dWr && dWh, "pgt_dcrnn_unpack_weight_grads_f32: null pointer");
  PGT_REQUIRE(dbzr == nullptr || (dbz && dbr), "pgt_dcrnn_unpack_weight_grads_f32: null bias gradient");
  dim3 grid;
  const int64_t total = 6 * K * C * O;
  if (int e = grid_for(total > 2 * O ? total : 2 * O, "pgt_dcrnn_unpack_weight_grads_f32", &grid)) return e;
  PGT_LAUNCH(dcrnn_unpack_grads_kernel, grid, dim3(256), stream, dWzr, dbzr, dWhs, (int)K, (int)C, (int)O, dWz, dWr, dWh, dbz, dbr);
  return pgt_check_launch("pgt_dcrnn_unpack_weight_grads_f32");
}

extern "C" int pgt_dcrnn_stage_f32(const float* X, const float* H0, int64_t T, int64_t M, int64_t Fin, int64_t O, float* TSzr0,
                                   float* TSh0, pgt_stream_t stream) {
  PGT_REQUIRE(T >= 0 && M >= 0 && Fin >= 0 && O >= 1, "pgt_dcrnn_stage_f32: bad size");
  if (T == 0 || M == 0) return PGT_OK;
  PGT_REQUIRE((Fin == 0 || X) && H0 && TSzr0 && TSh0, "pgt_dcrnn_stage_f32: null pointer");
  dim3 grid;
  if (int e = grid_for(T * M * Fin + M * O, "pgt_dcrnn_stage_f32", &grid)) return e;
  PGT_LAUNCH(dcrnn_stage_kernel, grid, dim3(256), stream, X, H0, T * M, M, (int)Fin, (int)O, TSzr0, TSh0);
  return pgt_check_launch("pgt_dcrnn_stage_f32");
}

extern "C" int pgt_copy2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t M, int64_t W,
                              pgt_stream_t stream) {
  return mover("pgt_copy2d_f32", dst, ldd, src, lds, 1.f, nullptr, 0, 0.f, M, W, 0, stream);
}

extern "C" int pgt_add2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t M, int64_t W,
                             pgt_stream_t stream) {
  return mover("pgt_add2d_f32", dst, ldd, src, lds, 1.f, nullptr, 0, 0.f, M, W, 1, stream);
}

extern "C" int pgt_axpby2d_f32(float* dst, int64_t ldd, const float* x, int64_t ldx, float a, const float* y,
                               int64_t ldy, float b, int64_t M, int64_t W, pgt_stream_t stream) {
  return mover("pgt_axpby2d_f32", dst, ldd, x, ldx, a, y, ldy, b, M, W, 2, stream);
}

extern "C" int pgt_swap01_f32(float* dst, const float* src, int64_t D0, int64_t D1, int64_t W,
                              pgt_stream_t stream) {
  PGT_REQUIRE(D0 >= 0 && D1 >= 0 && W >= 0, "pgt_swap01_f32: negative size");
  if (D0 == 0 || D1 == 0 || W == 0) return PGT_OK;
  PGT_REQUIRE(dst && src && dst != src, "pgt_swap01_f32: null or aliased pointer");
  PgtVecPick pick;
  pick.width(W);
  pick.operand(dst, W); pick.operand(src, W);
  dim3 grid, block(256);
  if (int e = grid_for(D0 * D1 * (W / pick.v), "pgt_swap01_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, swap01_kernel, grid, block, stream, dst, src, D0, D1, W);
  return pgt_check_launch("pgt_swap01_f32");
}

extern "C" int pgt_window_gather_f32(const float* data, int64_t T_total, int64_t W, const int64_t* idx, int64_t B,
                                     int64_t h, float* X, float* Y, int time_major, pgt_stream_t stream) {
  PGT_REQUIRE(T_total >= 0 && W >= 0 && B >= 0 && h >= 0, "pgt_window_gather_f32: negative size");
  if (B == 0 || h == 0 || W == 0) return PGT_OK;
  PGT_REQUIRE(data && idx && X && Y, "pgt_window_gather_f32: null pointer");
  PGT_REQUIRE(T_total >= 2 * h, "pgt_window_gather_f32: the series (%lld steps) is shorter than two windows of %lld",
              (long long)T_total, (long long)h);
  PgtVecPick pick;
  pick.width(W);
  pick.operand(data, W); pick.operand(X, W); pick.operand(Y, W);
  dim3 grid, block(256);
  if (int e = grid_for(2 * B * h * (W / pick.v), "pgt_window_gather_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, window_gather_kernel, grid, block, stream, data, idx, B, h, W, T_total, X, Y, time_major ? 1 : 0);
  return pgt_check_launch("pgt_window_gather_f32");
}

// ---- Adam over one flat parameter buffer (torch.optim.Adam's update, examples/indexBatching/DCRNN/pems_ddp.py:86, on the
// concatenation of all parameters: dp.FlatParameters).  torch's fused / foreach Adam hands a single 76 k-element tensor to TWO
// workgroups (96 us per step, whatever the batch); this is one thread per element.  The step count lives on the device so that the
// update can be captured in a hipGraph: a one-thread launch advances it, the update launch reads it.
namespace {
__global__ void adam_tick_kernel(float* step) { *step += 1.f; }
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ step, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float t = *step;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  float gi = g[i];
  const float pi = p[i];
  if (wd != 0.f) gi = fmaf(wd, pi, gi);
  const float mi = m[i] + (gi - m[i]) * (1.f - b1);             // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = fmaf(1.f - b2, gi * gi, b2 * v[i]);           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}
}  // namespace

extern "C" int pgt_adam_f32(float* p, const float* g, float* m, float* v, float* step, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, pgt_stream_t stream) {
  PGT_REQUIRE(n >= 0, "pgt_adam_f32: negative size");
  if (n == 0) return PGT_OK;
  PGT_REQUIRE(p && g && m && v && step, "pgt_adam_f32: null pointer");
  PGT_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), stream, step);
  dim3 grid, block(256);
  if (int e = grid_for(n, "pgt_adam_f32", &grid)) return e;
  PGT_LAUNCH(adam_kernel, grid, block, stream, p, g, m, v, step, n, lr, beta1, beta2, eps, weight_decay);
  return pgt_check_launch("pgt_adam_f32");
}
