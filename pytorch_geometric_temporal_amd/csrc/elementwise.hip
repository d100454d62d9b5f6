// GRU gate chains (dcrnn.py:172-192 / temporalgcn.py:82-102) fused into single passes, their backward twins,
// and the small strided movers that replace torch.cat / permute on the path.  All HBM-streaming: each lane moves
// V consecutive floats (V = 4/2/1 picked from pointer + stride alignment), consecutive lanes on consecutive V-groups.
#include "pgt_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

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
                                                     int64_t ld0, float* out1, int64_t ld1, int64_t M, int O) {
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
    hn[i] = z[i] * h[i] + (1.f - z[i]) * t[i];
  }
  pgt_stv<V>(pre_h + m * O + o, t);
  pgt_stv<V>(out0 + m * ld0 + o, hn);
  if (out1) pgt_stv<V>(out1 + m * ld1 + o, hn);
}

template <int V>
__global__ __launch_bounds__(256) void gru_h_bwd_kernel(const float* __restrict__ dHn, int64_t lddh,
                                                         const float* dHn2, int64_t lddh2,
                                                         const float* __restrict__ zr,
                                                         const float* __restrict__ H, int64_t ldh,
                                                         const float* __restrict__ ht, float* d_pre_h,
                                                         float* d_pre_zr, float* dH, int64_t lddhp, int acc,
                                                         int64_t M, int O) {
  const int OV = O / V;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * OV) return;
  const int64_t m = idx / OV;
  const int o = (int)(idx - m * OV) * V;
  float g[V], g2[V], z[V], h[V], t[V], dph[V], dpz[V], dh[V];
  pgt_ldv<V>(dHn + m * lddh + o, g);
  if (dHn2) {
    pgt_ldv<V>(dHn2 + m * lddh2 + o, g2);
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] += g2[i];
  }
  pgt_ldv<V>(zr + m * 2 * O + o, z);
  pgt_ldv<V>(H + m * ldh + o, h);
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
                                                          const float* __restrict__ H, int64_t ldh,
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
  pgt_ldv<V>(H + m * ldh + o, h);
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
                             int64_t ld0, float* out1, int64_t ld1, int64_t M, int64_t O,
                             pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(pre_h && zr && H && out0, "pgt_gru_h_f32: null pointer");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(pre_h, O); pick.operand(zr, 2 * O); pick.operand(H, ldh); pick.operand(out0, ld0);
  pick.operand(out1, ld1);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_h_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_h_kernel, grid, block, stream, pre_h, zr, H, ldh, out0, ld0, out1, ld1, M, (int)O);
  return pgt_check_launch("pgt_gru_h_f32");
}

extern "C" int pgt_gru_h_bwd_f32(const float* dHnew, int64_t lddh, const float* dHnew2, int64_t lddh2,
                                 const float* zr, const float* H, int64_t ldh, const float* ht, float* d_pre_h,
                                 float* d_pre_zr, float* dH, int64_t lddhp, int accumulate_dh, int64_t M,
                                 int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dHnew && zr && H && ht && d_pre_h && d_pre_zr && dH, "pgt_gru_h_bwd_f32: null pointer");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(dHnew, lddh); pick.operand(dHnew2, lddh2); pick.operand(zr, 2 * O); pick.operand(H, ldh);
  pick.operand(ht, O); pick.operand(d_pre_h, O); pick.operand(d_pre_zr, 2 * O); pick.operand(dH, lddhp);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_h_bwd_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_h_bwd_kernel, grid, block, stream, dHnew, lddh, dHnew2, lddh2, zr, H, ldh, ht, d_pre_h,
                d_pre_zr, dH, lddhp, accumulate_dh, M, (int)O);
  return pgt_check_launch("pgt_gru_h_bwd_f32");
}

extern "C" int pgt_gru_zr_bwd_f32(const float* dxhr, int64_t lddxhr, int64_t f_in, const float* zr,
                                  const float* H, int64_t ldh, float* d_pre_zr, float* dH, int64_t lddhp,
                                  int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0 && f_in >= 0, "pgt_gru_zr_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dxhr && zr && H && d_pre_zr && dH, "pgt_gru_zr_bwd_f32: null pointer");
  PgtVecPick pick;
  pick.width(O);
  pick.operand(dxhr + f_in, lddxhr); pick.operand(zr + O, 2 * O); pick.operand(H, ldh);
  pick.operand(d_pre_zr + O, 2 * O); pick.operand(dH, lddhp);
  dim3 grid, block(256);
  if (int e = grid_for(M * (O / pick.v), "pgt_gru_zr_bwd_f32", &grid)) return e;
  PGT_VDISPATCH(pick.v, gru_zr_bwd_kernel, grid, block, stream, dxhr, lddxhr, (int)f_in, zr, H, ldh, d_pre_zr, dH,
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
