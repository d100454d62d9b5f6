// TEST INFRASTRUCTURE (CPU test double only, never part of libpgt_hip.so): the platform layer of csrc/seq64.hip in plain C++.
// On the device these operations are gfx950 instructions (v_cvt_pk_bf16_f32, v_perm_b32, v_mfma_f32_16x16x32_bf16, exp / rcp,
// the LDS-only barrier); here the MFMA runs on the wavefront's 64 fibers.  Included inside seq64.hip's anonymous namespace.
#pragma once
typedef uint32_t sq_u32x4 __attribute__((vector_size(16)));
struct sq_f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
static inline float sq_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t sq_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t sq_bf16_rne(float x) {                         // v_cvt_pk_bf16_f32 on one value: bits of the bf16
  uint32_t u = sq_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // nan stays nan
  return (uint32_t)(((uint64_t)u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline uint32_t sq_pack(float x, float y) { return (sq_bf16_rne(x) & 0xffffu) | (sq_bf16_rne(y) << 16); }
static inline uint32_t sq_perm_hi16(uint32_t y, uint32_t x) { return (y & 0xffff0000u) | (x >> 16); }
static inline float sq_rcp(float x) { return 1.f / x; }
static inline float sq_exp(float x) { return expf(x); }
static inline void sq_barrier() { __syncthreads(); }
#define SQ_GLOAD4(dst, voff, sbase, imm) ((dst) = *reinterpret_cast<const sq_u32x4*>(reinterpret_cast<const char*>(sbase) + (voff) + (imm)))
static inline uint64_t sq_uniform64(uint64_t x) { return x; }
#define SQ_VMWAIT4(n, r0, r1, r2, r3) ((void)0)
#define SQ_VMWAIT6(n, r) ((void)0)
#define SQ_TOUCH(sink, ptr) ((void)(ptr))
#define SQ_TOUCH_DONE(sink) ((void)(sink))
// v_mfma_f32_16x16x32_bf16 on fibers: lane l holds A[i = l % 16][k = 8 (l / 16) .. + 7] and B[k = 8 (l / 16) .. + 7][j = l % 16];
// D: col = l % 16, row = 4 (l / 16) + reg
static uint32_t sq_emu_a[16][64][4], sq_emu_b[16][64][4];
static inline sq_f32x4 sq_mfma16(sq_u32x4 a, sq_u32x4 b, sq_f32x4 c) {
  pgt_emu::State& st = pgt_emu::S();
  const unsigned w = st.cur / 64, lane = st.cur % 64;
  for (int q = 0; q < 4; ++q) { sq_emu_a[w][lane][q] = a[q]; sq_emu_b[w][lane][q] = b[q]; }
  pgt_emu::wave_barrier();
  const unsigned col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const unsigned row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int kk = 0; kk < 32; ++kk) {
      const unsigned kg = kk >> 3, t = kk & 7;
      const uint32_t wa = sq_emu_a[w][row + 16 * kg][t >> 1], wb = sq_emu_b[w][col + 16 * kg][t >> 1];
      const float av = sq_as_float((t & 1) ? (wa & 0xffff0000u) : (wa << 16)), bv = sq_as_float((t & 1) ? (wb & 0xffff0000u) : (wb << 16));
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  pgt_emu::wave_barrier();
  return c;
}
// v_mfma_f32_16x16x4_f32 on fibers: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]; D as above; fmaf in k order
static float sq_emu_a4[16][64], sq_emu_b4[16][64];
static inline sq_f32x4 sq_mfma4(float a, float b, sq_f32x4 c) {
  pgt_emu::State& st = pgt_emu::S();
  const unsigned w = st.cur / 64, lane = st.cur % 64;
  sq_emu_a4[w][lane] = a;
  sq_emu_b4[w][lane] = b;
  pgt_emu::wave_barrier();
  const unsigned col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const unsigned row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(sq_emu_a4[w][row + 16 * k], sq_emu_b4[w][col + 16 * k], acc);
    c[r] = acc;
  }
  pgt_emu::wave_barrier();
  return c;
}
