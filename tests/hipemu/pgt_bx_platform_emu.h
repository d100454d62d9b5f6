// TEST INFRASTRUCTURE (CPU test double only, never part of libpgt_hip.so): the platform layer of csrc/gemm_bx.hip in plain
// C++.  On the device these operations are hand-written gfx950 instructions (buffer loads / stores through a descriptor,
// hand-counted s_waitcnt, the LDS-only barrier, v_cvt_pk_bf16_f32, v_perm_b32, v_mfma_f32_32x32x16_bf16, exp / rcp); here a
// descriptor is a (base, size) pair with the raw-buffer range rule, waits are no-ops, the flag poll yields to the other
// fibers and the MFMA runs on the wavefront's 64 fibers.  Included inside gemm_bx.hip's anonymous namespace.
#pragma once
typedef uint32_t bx_u32x4 __attribute__((vector_size(16)));
typedef uint32_t bx_u32x2 __attribute__((vector_size(8)));
struct BxRsrc { const unsigned char* base; uint32_t bytes; };
static inline BxRsrc bx_make_rsrc(const void* p, int64_t bytes) { return BxRsrc{static_cast<const unsigned char*>(p), (uint32_t)bytes}; }
static inline BxRsrc bx_select_rsrc(bool c, const BxRsrc& a, const BxRsrc& b) { return c ? a : b; }
static inline bool bx_in_range(const BxRsrc& r, uint64_t off, unsigned size) { return off + size <= r.bytes; }   // raw buffer rule
#define BX_LOAD2(dst, voff, rs) do { uint64_t o_ = (voff); if (bx_in_range(rs, o_, 8)) memcpy(&(dst), (rs).base + o_, 8); else memset(&(dst), 0, 8); } while (0)
#define BX_LOAD2S(dst, voff, rs, soff) do { uint64_t o_ = (uint64_t)(uint32_t)(voff) + (uint32_t)(soff); if (bx_in_range(rs, o_, 8)) memcpy(&(dst), (rs).base + o_, 8); else memset(&(dst), 0, 8); } while (0)
#define BX_LOAD1(dst, voff, rs) do { uint64_t o_ = (voff); if (bx_in_range(rs, o_, 4)) memcpy(&(dst), (rs).base + o_, 4); else memset(&(dst), 0, 4); } while (0)
#define BX_LOAD1S(dst, voff, rs, soff) do { uint64_t o_ = (uint64_t)(uint32_t)(voff) + (uint32_t)(soff); if (bx_in_range(rs, o_, 4)) memcpy(&(dst), (rs).base + o_, 4); else memset(&(dst), 0, 4); } while (0)
#define BX_STORE1S(val, voff, rs, soff) do { uint64_t o_ = (uint64_t)(uint32_t)(voff) + (uint32_t)(soff); float v_ = (val); if (bx_in_range(rs, o_, 4)) memcpy(const_cast<unsigned char*>((rs).base) + o_, &v_, 4); } while (0)
#define BX_LOAD4(dst, voff, rs) do { uint64_t o_ = (uint32_t)(voff); if (bx_in_range(rs, o_, 16)) memcpy(&(dst), (rs).base + o_, 16); else memset(&(dst), 0, 16); } while (0)
#define BX_STORE4(val, voff, rs) do { uint64_t o_ = (uint32_t)(voff); bx_u32x4 v_ = (val); if (bx_in_range(rs, o_, 16)) memcpy(const_cast<unsigned char*>((rs).base) + o_, &v_, 16); } while (0)
#define BX_WAIT(n, reg) ((void)0)
#define BX_WAIT2(n, r0, r1) ((void)0)
#define BX_WAIT8(n, a0, a1, a2, a3, a4, a5, a6, a7) ((void)0)
#define BX_WAIT_PLAIN(n) ((void)0)
#define BX_DRAIN() ((void)0)
#define BX_FENCE() ((void)0)
#define BX_SGPR(x) (x)
#define BX_SETPRIO(n) ((void)0)
#define BX_YIELD() pgt_emu::yield_()
typedef volatile int bx_lds_vint;
static inline void bx_barrier() { __syncthreads(); }
static inline float bx_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bx_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t bx_bf16_rne(float x) {                         // v_cvt_pk_bf16_f32 on one value: bits of the bf16
  uint32_t u = bx_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // nan stays nan
  return (uint32_t)(((uint64_t)u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline uint32_t bx_pack(float x, float y) { return (bx_bf16_rne(x) & 0xffffu) | (bx_bf16_rne(y) << 16); }
static inline uint32_t bx_perm_hi16(uint32_t y, uint32_t x) { return (y & 0xffff0000u) | (x >> 16); }
static inline float bx_rcp(float x) { return 1.f / x; }
static inline float bx_exp(float x) { return expf(x); }
// v_mfma_f32_32x32x16_bf16 on fibers: lane l holds A[i = l % 32][k = 8 (l / 32) .. + 7] and B[k = 8 (l / 32) .. + 7][j = l % 32];
// D as pgt_emu_mfma_32x32x2
static uint32_t bx_emu_a[16][64][4], bx_emu_b[16][64][4];
static inline pgt_f32x16 bx_mfma(bx_u32x4 a, bx_u32x4 b, pgt_f32x16 c) {
  pgt_emu::State& st = pgt_emu::S();
  const unsigned w = st.cur / 64, lane = st.cur % 64;
  for (int q = 0; q < 4; ++q) { bx_emu_a[w][lane][q] = a[q]; bx_emu_b[w][lane][q] = b[q]; }
  pgt_emu::wave_barrier();
  const unsigned col = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int kk = 0; kk < 16; ++kk) {
      const unsigned kh = kk >> 3, t = kk & 7;
      const uint32_t wa = bx_emu_a[w][row + 32 * kh][t >> 1], wb = bx_emu_b[w][col + 32 * kh][t >> 1];
      const float av = bx_as_float((t & 1) ? (wa & 0xffff0000u) : (wa << 16)), bv = bx_as_float((t & 1) ? (wb & 0xffff0000u) : (wb << 16));
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  pgt_emu::wave_barrier();
  return c;
}
