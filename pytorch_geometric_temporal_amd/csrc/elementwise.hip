// GRU gate chains (dcrnn.py:172-192 / temporalgcn.py:82-102) fused into single passes, their backward twins,
// and the small strided movers that replace torch.cat / permute on the path.  All HBM-streaming, one element per
// lane, consecutive lanes on consecutive floats.
#include "pgt_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_zr_kernel(float* pre_zr, const float* __restrict__ H, int64_t ldh,
                                                      float* xhr, int64_t ldxhr, int f_in, int64_t M, int O) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * O) return;
  const int64_t m = idx / O;
  const int o = (int)(idx - m * O);
  float* p = pre_zr + m * 2 * O;
  const float z = sigmoidf_(p[o]);
  const float r = sigmoidf_(p[O + o]);
  p[o] = z;
  p[O + o] = r;
  xhr[m * ldxhr + f_in + o] = H[m * ldh + o] * r;
}

__global__ __launch_bounds__(256) void gru_h_kernel(float* pre_h, const float* __restrict__ zr,
                                                     const float* __restrict__ H, int64_t ldh, float* out0,
                                                     int64_t ld0, float* out1, int64_t ld1, int64_t M, int O) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * O) return;
  const int64_t m = idx / O;
  const int o = (int)(idx - m * O);
  const float ht = tanhf(pre_h[idx]);
  pre_h[idx] = ht;
  const float z = zr[m * 2 * O + o];
  const float h = H[m * ldh + o];
  const float hn = z * h + (1.f - z) * ht;
  out0[m * ld0 + o] = hn;
  if (out1) out1[m * ld1 + o] = hn;
}

__global__ __launch_bounds__(256) void gru_h_bwd_kernel(const float* __restrict__ dHn, int64_t lddh,
                                                         const float* __restrict__ zr,
                                                         const float* __restrict__ H, int64_t ldh,
                                                         const float* __restrict__ ht, float* d_pre_h,
                                                         float* d_pre_zr, float* dH, int64_t lddhp, int acc,
                                                         int64_t M, int O) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * O) return;
  const int64_t m = idx / O;
  const int o = (int)(idx - m * O);
  const float g = dHn[m * lddh + o];
  const float z = zr[m * 2 * O + o];
  const float h = H[m * ldh + o];
  const float t = ht[idx];
  d_pre_h[idx] = g * (1.f - z) * (1.f - t * t);
  d_pre_zr[m * 2 * O + o] = g * (h - t) * z * (1.f - z);
  float* q = dH + m * lddhp + o;
  *q = acc ? (*q + g * z) : (g * z);
}

__global__ __launch_bounds__(256) void gru_zr_bwd_kernel(const float* __restrict__ dxhr, int64_t lddxhr, int f_in,
                                                          const float* __restrict__ zr,
                                                          const float* __restrict__ H, int64_t ldh,
                                                          float* d_pre_zr, float* dH, int64_t lddhp, int64_t M,
                                                          int O) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * O) return;
  const int64_t m = idx / O;
  const int o = (int)(idx - m * O);
  const float g = dxhr[m * lddxhr + f_in + o];
  const float r = zr[m * 2 * O + O + o];
  const float h = H[m * ldh + o];
  d_pre_zr[m * 2 * O + O + o] = g * h * r * (1.f - r);
  dH[m * lddhp + o] += g * r;
}

// mode 0: dst = x ; 1: dst += x ; 2: dst = a*x + b*y
__global__ __launch_bounds__(256) void mover2d_kernel(float* dst, int64_t ldd, const float* __restrict__ x,
                                                       int64_t ldx, float a, const float* y, int64_t ldy, float b,
                                                       int64_t M, int64_t W, int mode) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * W) return;
  const int64_t m = idx / W;
  const int64_t w = idx - m * W;
  const float xv = x[m * ldx + w];
  float* d = dst + m * ldd + w;
  if (mode == 0) *d = xv;
  else if (mode == 1) *d += xv;
  else *d = a * xv + (y ? b * y[m * ldy + w] : 0.f);
}

__global__ __launch_bounds__(256) void swap01_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                      int64_t D0, int64_t D1, int64_t W) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= D0 * D1 * W) return;
  const int64_t w = idx % W;
  const int64_t d1 = (idx / W) % D1;
  const int64_t d0 = idx / (W * D1);
  dst[(d1 * D0 + d0) * W + w] = src[idx];
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

}  // namespace

extern "C" int pgt_gru_zr_f32(float* pre_zr, const float* H, int64_t ldh, float* xhr, int64_t ldxhr,
                              int64_t f_in, int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0 && f_in >= 0, "pgt_gru_zr_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(pre_zr && H && xhr, "pgt_gru_zr_f32: null pointer");
  dim3 grid, block(256);
  if (int e = grid_for(M * O, "pgt_gru_zr_f32", &grid)) return e;
  PGT_LAUNCH(gru_zr_kernel, grid, block, stream, pre_zr, H, ldh, xhr, ldxhr, (int)f_in, M, (int)O);
  return pgt_check_launch("pgt_gru_zr_f32");
}

extern "C" int pgt_gru_h_f32(float* pre_h, const float* zr, const float* H, int64_t ldh, float* out0,
                             int64_t ld0, float* out1, int64_t ld1, int64_t M, int64_t O,
                             pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(pre_h && zr && H && out0, "pgt_gru_h_f32: null pointer");
  dim3 grid, block(256);
  if (int e = grid_for(M * O, "pgt_gru_h_f32", &grid)) return e;
  PGT_LAUNCH(gru_h_kernel, grid, block, stream, pre_h, zr, H, ldh, out0, ld0, out1, ld1, M, (int)O);
  return pgt_check_launch("pgt_gru_h_f32");
}

extern "C" int pgt_gru_h_bwd_f32(const float* dHnew, int64_t lddh, const float* zr, const float* H,
                                 int64_t ldh, const float* ht, float* d_pre_h, float* d_pre_zr, float* dH,
                                 int64_t lddhp, int accumulate_dh, int64_t M, int64_t O,
                                 pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0, "pgt_gru_h_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dHnew && zr && H && ht && d_pre_h && d_pre_zr && dH, "pgt_gru_h_bwd_f32: null pointer");
  dim3 grid, block(256);
  if (int e = grid_for(M * O, "pgt_gru_h_bwd_f32", &grid)) return e;
  PGT_LAUNCH(gru_h_bwd_kernel, grid, block, stream, dHnew, lddh, zr, H, ldh, ht, d_pre_h, d_pre_zr, dH, lddhp,
             accumulate_dh, M, (int)O);
  return pgt_check_launch("pgt_gru_h_bwd_f32");
}

extern "C" int pgt_gru_zr_bwd_f32(const float* dxhr, int64_t lddxhr, int64_t f_in, const float* zr,
                                  const float* H, int64_t ldh, float* d_pre_zr, float* dH, int64_t lddhp,
                                  int64_t M, int64_t O, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && O >= 0 && f_in >= 0, "pgt_gru_zr_bwd_f32: negative size");
  if (M == 0 || O == 0) return PGT_OK;
  PGT_REQUIRE(dxhr && zr && H && d_pre_zr && dH, "pgt_gru_zr_bwd_f32: null pointer");
  dim3 grid, block(256);
  if (int e = grid_for(M * O, "pgt_gru_zr_bwd_f32", &grid)) return e;
  PGT_LAUNCH(gru_zr_bwd_kernel, grid, block, stream, dxhr, lddxhr, (int)f_in, zr, H, ldh, d_pre_zr, dH, lddhp, M,
             (int)O);
  return pgt_check_launch("pgt_gru_zr_bwd_f32");
}

static int mover(const char* what, float* dst, int64_t ldd, const float* x, int64_t ldx, float a, const float* y,
                 int64_t ldy, float b, int64_t M, int64_t W, int mode, pgt_stream_t stream) {
  PGT_REQUIRE(M >= 0 && W >= 0, "%s: negative size", what);
  if (M == 0 || W == 0) return PGT_OK;
  PGT_REQUIRE(dst && x, "%s: null pointer", what);
  dim3 grid, block(256);
  if (int e = grid_for(M * W, what, &grid)) return e;
  PGT_LAUNCH(mover2d_kernel, grid, block, stream, dst, ldd, x, ldx, a, y, ldy, b, M, W, mode);
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
  dim3 grid, block(256);
  if (int e = grid_for(D0 * D1 * W, "pgt_swap01_f32", &grid)) return e;
  PGT_LAUNCH(swap01_kernel, grid, block, stream, dst, src, D0, D1, W);
  return pgt_check_launch("pgt_swap01_f32");
}
