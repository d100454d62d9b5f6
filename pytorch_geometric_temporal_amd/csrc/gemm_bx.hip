// Dense feature transform through the bf16 matrix pipe with fp32 accuracy ("split-bf16").
//
// Why: gfx950 has no TF32, and v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (157 vs 2 500 TFLOP/s), so the fp32
// tile kernels of gemm.hip are matrix-pipe bound at the DCRNN shapes (M = N_nodes * B rows, K = 330, 128 / 64 columns:
// 0.34 - 0.59 of the fp32 peak) although the operands would stream from HBM in a third of the time.  Here every fp32
// operand is split into three bf16 pieces, x = x1 + x2 + x3 (8 significant bits each = the 24 bits of an fp32
// significand), and a product is accumulated from the six largest piece products
//     x1 y1  +  (x1 y2 + x2 y1 + x2 y2 + x1 y3 + x3 y1)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the big term and the corrections in separate accumulators, added
// once at the end).  Each bf16 x bf16 product is exact in fp32; the dropped terms (x2 y3, x3 y2, x3 y3) are below
// 2^-24 |x y|.  Measured against an fp64 product (K = 330, lab/gemm_bx_lab.hip): mean error 3.7e-8, max 3.9e-7 — the
// fp32 MFMA kernel of gemm.hip on the same inputs: 1.4e-7 / 1.6e-6.  Six bf16 MFMAs are 2.67x the fp32 MFMA peak.
// Non-finite operands: an infinite x gives x - x1 = nan, so its output row is nan where the fp32 product has +-inf or
// nan (nan operands propagate as usual).
//
// Schedule: one persistent 512-thread workgroup per CU = two wavefronts per SIMD.
//   * wavefronts 0 .. 3 ("consumers") own 32 * WN output columns each and the first KA k-steps (16 deep) of K, and run
//     the epilogue (bias, GRU gate chain of pgt_gemm_gru_zr/h_f32, stores straight from the accumulator layout: one
//     register = one 128-byte row piece per half-wavefront);
//   * wavefronts 4 .. 7 ("producers") own the same columns for the remaining k-steps and bring the next 32-row block of A
//     in: global fp32 (8-byte loads through a buffer descriptor that ends at the last valid row: ragged and
//     out-of-range blocks read zeros, no branch) -> three bf16 planes in the other LDS buffer;
//   * the B slice of a wavefront (its columns x its part of K, three planes) stays in registers for the whole launch;
//   * the two K parts meet in LDS: the producer leaves its partial sums there and moves on to the next block, the
//     consumer adds them and stores.  One workgroup barrier per block (LDS counters only: loads and stores stay in
//     flight across it) plus an LDS flag that orders the reuse of the partial-sum buffer.
// The producers' loads are issued with inline asm and waited for by hand (s_waitcnt vmcnt(EPT - 1)): the compiler's
// counter insertion drains vmcnt at every loop back-edge, which serialises a load with the k-steps it should hide
// behind (measured: 157 -> 128 us).  For the same reason nothing in the steady state uses FLAT or scratch accesses.
#include "pgt_common.h"

namespace {

int g_bx = 1;   // pgt_tune("gemm_bx"): 1 = where it applies (>= 8192 rows), 2 = at any size (tests), 0 = never
int g_bx_sym = 1;   // pgt_tune("gemm_bx_sym"): 0 = short-K products on the K-split kernel instead of the symmetric one (A/B)
int g_bx_sym_pc = 1;   // pgt_tune("gemm_bx_sym_pc"): 1 = K <= 64 products on specialised wavefronts (gemm_bx_sym_pc_kernel), 0 = all alike
int g_bx_tn_pc = 1;   // pgt_tune("gemm_bx_tn_pc"): 1 = the weight gradient on specialised wavefronts (gemm_bx_tn_pc_kernel), 0 = all alike, 2 = twelve wavefronts at N = 128 too

// ---- platform layer: the handful of operations below are hand-written gfx950 instructions.  The CPU test double compiles
// the SAME kernel bodies against tests/hipemu/pgt_bx_platform_emu.h, which spells these operations in plain C++ (fibers,
// no vmcnt, descriptor bounds as a range check); this translation unit holds gfx950 code only.
#ifdef PGT_EMU
#include "pgt_bx_platform_emu.h"   // tests/hipemu: the same operations in plain C++ for the CPU test double (never shipped)
#else
typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t bx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bx_u32x2 __attribute__((ext_vector_type(2)));
// buffer descriptor (raw buffer, stride 0): accesses past `bytes` read zero / are dropped
typedef bx_u32x4 BxRsrc;
__device__ __forceinline__ BxRsrc bx_make_rsrc(const void* p, int64_t bytes) {
  const uint64_t base = reinterpret_cast<uint64_t>(p);
  bx_u32x4 r = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) & 0xffffu,
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bytes), 0x00020000u};
  return r;
}
// one of two descriptors by a wave-uniform condition, the result in scalar registers
__device__ __forceinline__ BxRsrc bx_select_rsrc(bool c, const BxRsrc& a, const BxRsrc& b) {
  bx_u32x4 r = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(c ? a[0] : b[0])), (uint32_t)__builtin_amdgcn_readfirstlane((int)(c ? a[1] : b[1])),
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(c ? a[2] : b[2])), (uint32_t)__builtin_amdgcn_readfirstlane((int)(c ? a[3] : b[3]))};
  return r;
}
#define BX_LOAD2(dst, voff, rs) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory")
#define BX_LOAD2S(dst, voff, rs, soff) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory")
#define BX_LOAD1(dst, voff, rs) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory")
#define BX_LOAD1S(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory")
#define BX_STORE1S(val, voff, rs, soff) asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(val), "v"(voff), "s"(rs), "s"(soff) : "memory")
#define BX_LOAD4(dst, voff, rs) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory")
// (a store of more than 8 bytes reads its data registers over several cycles: a VALU write to them in the next two issue
// slots corrupts the stored value — the compiler pads its own stores for this hazard and cannot see into the asm)
#define BX_STORE4(val, voff, rs) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(val), "v"(voff), "s"(rs) : "memory")
// hand-counted waits, tied to the registers they release so that the consumer cannot be scheduled above them
#define BX_WAIT(n, reg) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n))
#define BX_WAIT2(n, r0, r1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(n))
#define BX_WAIT8(n, a0, a1, a2, a3, a4, a5, a6, a7) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "n"(n))
#define BX_WAIT_PLAIN(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#define BX_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define BX_FENCE() asm volatile("" ::: "memory")
#define BX_SGPR(x) __builtin_amdgcn_readfirstlane(x)
#define BX_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define BX_YIELD() ((void)0)
typedef __attribute__((address_space(3))) volatile int bx_lds_vint;   // an LDS access (a generic pointer would be a FLAT load,
                                                                      // and FLAT waits drain vmcnt)
// LDS-only workgroup barrier: planes and partial sums travel through LDS (lgkmcnt); the loads of the blocks ahead and the
// epilogue's stores stay in flight across it (__syncthreads would drain vmcnt as well)
__device__ __forceinline__ void bx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float bx_as_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t bx_as_uint(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint32_t bx_pack(float x, float y) {     // v_cvt_pk_bf16_f32: round to nearest even
  bx_f32x2 v = {x, y};
  bx_bf16x2 r = __builtin_convertvector(v, bx_bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t bx_perm_hi16(uint32_t y, uint32_t x) { return __builtin_amdgcn_perm(y, x, 0x07060302u); }
__device__ __forceinline__ float bx_rcp(float x) { return __frcp_rn(x); }
__device__ __forceinline__ float bx_exp(float x) { return __expf(x); }
__device__ __forceinline__ pgt_f32x16 bx_mfma(bx_u32x4 a, bx_u32x4 b, pgt_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bx_bf16x8, a), __builtin_bit_cast(bx_bf16x8, b), c, 0, 0, 0);
}
#endif

// (x, y) -> three packed bf16 pairs (low half = x's piece), every piece rounded to nearest: the resident operand
__device__ __forceinline__ void bx_split2(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = bx_pack(x, y);
  float rx = x - bx_as_float(p1 << 16), ry = y - bx_as_float(p1 & 0xffff0000u);
  rx = (fabsf(rx) <= 3.0e38f) ? rx : 0.f;      // inf / nan: the first piece carries it, the others are zero
  ry = (fabsf(ry) <= 3.0e38f) ? ry : 0.f;
  p2 = bx_pack(rx, ry);
  rx -= bx_as_float(p2 << 16);
  ry -= bx_as_float(p2 & 0xffff0000u);
  p3 = bx_pack(rx, ry);
}
// the streaming operand: first piece rounded to nearest, the other two cut off (x - x1 has at most 16 significant bits,
// the second cut leaves at most 9): 9 instructions per pair
__device__ __forceinline__ void bx_split2_fast(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = bx_pack(x, y);
  float rx = x - bx_as_float(p1 << 16), ry = y - bx_as_float(p1 & 0xffff0000u);
  p2 = bx_perm_hi16(bx_as_uint(ry), bx_as_uint(rx));
  rx -= bx_as_float(p2 << 16);
  ry -= bx_as_float(p2 & 0xffff0000u);
  p3 = bx_perm_hi16(bx_as_uint(ry), bx_as_uint(rx));
}

// sigmoid / tanh on the hardware exp / reciprocal (1 ulp each): the gate chain runs on the four consumer wavefronts only,
// sixteen elements per lane and block, and the library expf + IEEE division (~50 instructions per element) was a third of
// the fused kernel's time (182 -> 168 us at M = 211 968)
__device__ __forceinline__ float bx_sigmoidf(float x) { return bx_rcp(1.f + bx_exp(-x)); }
// 1 - 2 / (1 + e^2x) has an ABSOLUTE error of ~1e-7, i.e. a poor relative one near 0: there the odd series takes over
// (|x| < 0.04: the x^7 term is below 1e-11 |x|)
__device__ __forceinline__ float bx_tanhf(float x) {
  const float x2 = x * x;
  const float small = x * fmaf(x2, fmaf(x2, 0.13333334f, -0.33333334f), 1.f);
  const float big = 1.f - 2.f * bx_rcp(1.f + bx_exp(2.f * x));
  return fabsf(x) < 0.04f ? small : big;
}

// ---- non-finite operands.  An infinite or nan element of the streaming operand turns its whole output ROW into nan
// (x - x1 = nan reaches every column), a non-finite weight its whole column (0 * inf in the zero-padded k-steps), where
// the reference's fp32 product (torch.matmul, dcrnn.py:81-105) has +-inf or nan element by element.  Every wavefront that
// finds a nan among the sums of a 32 x 32 tile recomputes the tile as an fp32 fmaf chain over the ORIGINAL operands — the
// arithmetic of gemm.hip's kernels — so that inf / nan land exactly where the reference puts them.  Rare path (plain
// loads, ~0.1 ms per tile); rows past M are clamped (their sums are never stored).
__device__ __forceinline__ bool bx_tile_has_nan(const float (&acc)[16]) {
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) bad |= acc[r] != acc[r];
  return __ballot(bad) != 0;
}
// `ra`: the buffer descriptor of the tile's 32-row block of A (ends with the last valid row of the last segment; rows past
// M of the other segments read whatever follows — their sums are never stored).  One lane register per load and a
// uniform row offset: no 64-bit address per row, so the rare path costs the hot loop no registers.
__device__ __forceinline__ void bx_exact_tile(const PgtGemmArgs& g, const BxRsrc& ra, int col, int hi, float bias, float (&acc)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool col_ok = col < g.N;
  const float* bcol = g.Bw + (int64_t)(col_ok ? col : 0) * g.sbn;
  const uint32_t lda4 = (uint32_t)(g.lda * 4);
  for (int seg = 0; seg < g.n_seg; ++seg) {
    uint32_t voff = (uint32_t)((seg * g.a_seg_stride + 4 * hi * g.lda) * 4);
    for (int kk = 0; kk < g.seg_k; ++kk, voff += 4) {
      const float b = col_ok ? bcol[(int64_t)(seg * g.seg_k + kk) * g.sbk] : 0.f;
      float a[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = BX_SGPR((int)(((r & 3) + 8 * (r >> 2)) * lda4));
        BX_LOAD1S(a[r], voff, ra, soff);
      }
      BX_DRAIN();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        BX_WAIT(0, a[r]);
        acc[r] = fmaf(a[r], b, acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += bias;
}

// compile-time integers handed to lambdas (s_waitcnt takes an immediate)
template <int V> struct BxInt { static constexpr int value = V; };
// the row-sliced epilogue's descriptors: a 2 GB window over an array, and the offset that lies outside it
constexpr uint32_t BX_WIN = 0x80000000u, BX_OOR = 0xc0000000u;

// lab/gemm_bx_trace_lab.hip defines this to record a per-wavefront timeline of the K-split kernel; a no-op in the library
#ifndef BX_TRACE
#define BX_TRACE(slot) do { } while (0)
#endif
// ... and this to take the kernel apart (1: no MFMAs, 4: no epilogue stores, 8: no loads of A, 16: no gate-operand loads)
#ifndef BX_LAB_SKIP
#define BX_LAB_SKIP(bit) false
#endif

// KSTEPS: 16-deep k-steps covering K (zero padded); WN: 32-column blocks per wavefront; EPI: 0 bias, 1 / 2 the GRU
// epilogues of PgtGemmArgs.  A: n_seg segments of seg_k (even) columns, consumed as one [M, n_seg * seg_k] operand.
// Q4 (N <= 64): only two column blocks exist, so K is cut four ways instead — consumers 0, 1 (epilogue) and 2, 3, producers
// 0, 1 and 2, 3 each take a quarter of the k-steps of column block (wavefront & 1) and three partial sums meet in LDS.
template <int KSTEPS, int WN, int EPI, bool Q4>
__global__ __launch_bounds__(512, 1) void gemm_bx_kernel(PgtGemmArgs g, int n_blocks) {
  constexpr int BM = 32, KP = KSTEPS * 16, SROW = KP * 2 + 16, PLANE = BM * SROW, BUF = 3 * PLANE;
  constexpr int EPT = (KP / 2) / 8;                       // float pairs per producer thread and block (8 threads per row)
  constexpr int PART = 64 * 16 * WN * 4;                  // a column's partial sums, accumulator layout
  constexpr int NPART = Q4 ? 4 : 2, NCOL = Q4 ? 2 : 4;     // parts of K x column blocks = the eight wavefronts
  // k-steps of the parts.  The producers also convert the next block (round 3's timeline: their k-loop took 2.6 - 4.1 us
  // against the consumers' 1.5 - 3.1, and the consumers waited for them at the barrier), so they get FEWER k-steps:
  // four parts (registers to spare): consumers 0, 1 KC0 = 8 of 21, consumers 2, 3 KC1 = 9, producers KPR = 2 each (same box,
  // inside the step: 130 - 133 us; 7 / 8 / 3 / 3: 128 - 147, two modes; the even split of round 2: 131 - 146); two parts:
  // consumers KC0 = 11, producers KPR = 10 (136 -> 132 us) — 12 / 9 would balance them, but 144 registers of B spill.
  constexpr int KPR = Q4 ? (KSTEPS / 10 > 0 ? KSTEPS / 10 : 1) : (KSTEPS * 10 / 21 > 0 ? KSTEPS * 10 / 21 : 1);
  constexpr int KC0 = Q4 ? (KSTEPS - 2 * KPR) / 2 : KSTEPS - KPR, KC1 = Q4 ? KSTEPS - 2 * KPR - KC0 : 0;
  constexpr int KMAX = KC0 > KC1 ? (KC0 > KPR ? KC0 : KPR) : (KC1 > KPR ? KC1 : KPR);
  constexpr int NREG = (NPART - 1) * NCOL;                 // partial-sum regions
  // ---- WN == 1: the block's sums leave through a ROW-SLICED epilogue on all eight wavefronts.  (Round 3's timeline,
  // lab/gemm_bx_trace_lab.hip: with the epilogue on the part-0 wavefronts alone — 16 rows x 1 .. 3 arrays of 128-byte row
  // pieces per lane, 64-bit address chains — a block's stores took 2.8 - 3.3 us of a 6.3 - 7.4 us iteration while the other
  // wavefronts waited at the barrier.)  The part-0 wavefronts add the partial sums as before and leave the block's sums in
  // LDS ([column block][row][32]); after a second barrier wavefront w owns rows 4w .. 4w + 3: a lane is a column (two at 128
  // columns), every load / store instruction is one whole row piece (256 contiguous bytes) through a per-row buffer
  // descriptor — scalar row addressing, masking by the descriptor's range, no branches, no vector address arithmetic.
  constexpr bool ROWS = WN == 1;
  constexpr int CPL = ROWS ? NCOL / 2 : 1;                 // 64-column groups of the block = columns per lane
  // gate operands live in ONE 64-column group: the candidate gate has a single group; of the z | r gates' 2 O columns only
  // the reset half needs H, and that half is the LAST group (N = 2 O = 128 at two groups, N <= 64 at one)
  constexpr int OPG = CPL - 1;                             // the group that carries gate operands
  constexpr int NOP = !ROWS || EPI == 0 ? 0 : (EPI == 1 ? 1 : 2);   // gate-operand loads (16 bytes per lane) per wavefront and block
  static_assert((KP / 2) % 8 == 0 && KPR >= 1 && KC0 >= 1 && (!Q4 || KC1 >= 1) && (!Q4 || WN == 1) && (EPI == 0 || WN == 1), "shape");
  static_assert(EPT - 1 + NOP < 64, "vmcnt is six bits");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + NREG * PART + 64];
  unsigned char* const stage_part = lds + 2 * BUF;
  bx_lds_vint* const part_seen = (bx_lds_vint*)(lds + 2 * BUF + NREG * PART);
  bx_lds_vint* const tile_seen = part_seen + 4;            // [8]: wavefront w has taken its rows of block n_iter - 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = BX_SGPR(tid >> 6);
  const int wc = wave & 3;
  const bool producer = wave >= 4;
  const int cb = Q4 ? (wc & 1) : wc;                                       // column block
  const int part = (producer ? NPART / 2 : 0) + (Q4 ? (wc >> 1) : 0);      // part of K
  const int kbase = producer ? KC0 + KC1 + (Q4 ? (wc >> 1) * KPR : 0) : (part == 0 ? 0 : KC0);
  const int ksteps = producer ? KPR : (part == 0 ? KC0 : KC1);
  const int nwg = gridDim.x;
  const int Ktot = g.n_seg * g.seg_k;
  // ---- B slice -> registers (this wavefront's columns x its part of K); every piece rounded to nearest
  bx_u32x4 bf[KMAX][WN][3];
  {
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = (cb * WN + j) * 32 + (lane & 31), k0 = (kbase + i) * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int k = k0 + t;
          v[t] = (k < Ktot && col < g.N && i < ksteps) ? g.Bw[(int64_t)k * g.sbk + (int64_t)col * g.sbn] : 0.f;
        }
        uint32_t p[3][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bx_split2(v[2 * t], v[2 * t + 1], p[0][t], p[1][t], p[2][t]);
#pragma unroll
        for (int q = 0; q < 3; ++q) { bx_u32x4 f = {p[q][0], p[q][1], p[q][2], p[q][3]}; bf[i][j][q] = f; }
      }
  }
  // ---- zero both A buffers once (the K padding columns are never written again)
  for (int i = tid; i < 2 * BUF / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 12) part_seen[tid] = 0;
  int rb = blockIdx.x;
  if (rb >= n_blocks) return;
  __syncthreads();
  const int arow = (lane & 31) * SROW + 16 * (lane >> 5);
  int n_iter = 0;

  // ---- the row-sliced epilogue (ROWS), 16-byte accesses.  Wavefront w owns rows 4w .. 4w + 3 of the block; lane l is row
  // 4w + (l >> 4) and the four columns 4 (l & 15) .. + 3 of each 64-column group: ONE buffer_load / buffer_store_dwordx4
  // moves four whole 256-byte row pieces (1 KB per instruction — a CU keeps a bounded number of vector-memory
  // instructions in flight, so bytes per instruction is what its stream rate is made of).  Offsets are absolute inside
  // an array (the host side checks that they fit 32 bits); rows past M / columns past N / the update half of the z | r
  // gates get an offset past the descriptor (a 2 GB window; the offset 3 GB: no wrap-around inside the 16-byte range check):
  // read zero, are dropped.
  const int r4 = lane >> 4, q4 = lane & 15;
  uint32_t ccol[CPL];                   // byte offset of the quad inside a row of C, its segment folded in
  bool live[CPL];
#pragma unroll
  for (int s = 0; s < CPL; ++s) {
    const int col0 = 64 * s + 4 * q4;
    live[s] = ROWS && col0 < g.N;
    const int js = live[s] ? col0 / g.c_seg_n : 0;
    ccol[s] = (uint32_t)(((int64_t)js * g.c_seg_stride + (col0 - js * g.c_seg_n)) * 4);
  }
  const int ocol = 64 * OPG + 4 * q4;                                  // the quad that carries gate operands
  const bool olive = ROWS && EPI != 0 && ocol < g.N && (EPI == 2 || ocol >= g.eO);
  const uint32_t hcol4 = (uint32_t)((EPI == 1 ? ocol - g.eO : ocol) * 4);
  const BxRsrc rs_c = bx_make_rsrc(g.C, BX_WIN);
  const BxRsrc rs_h = bx_make_rsrc(EPI != 0 ? g.eH : nullptr, EPI != 0 ? BX_WIN : 0);
  const BxRsrc rs_z = bx_make_rsrc(EPI == 2 ? g.eZ : nullptr, EPI == 2 ? BX_WIN : 0);
  const BxRsrc rs_x = bx_make_rsrc(EPI == 1 ? g.eX + g.efin : nullptr, EPI == 1 ? BX_WIN : 0);
  const BxRsrc rs_0 = bx_make_rsrc(EPI == 2 ? g.eO0 : nullptr, EPI == 2 ? BX_WIN : 0);
  const BxRsrc rs_1 = bx_make_rsrc(EPI == 2 ? g.eO1 : nullptr, (EPI == 2 && g.eO1) ? BX_WIN : 0);
  bx_u32x4 eh = {0, 0, 0, 0}, ez = {0, 0, 0, 0};
  // gate operands of this lane's row of block `b`: always NOP load instructions (the producers' hand-counted waits count them)
  auto e_issue_rows = [&](int b) {
    if (BX_LAB_SKIP(16)) return;
    if constexpr (NOP > 0) {
      const int gm = b * BM + 4 * wave + r4;
      const bool ok = olive && gm < g.M;
      BX_LOAD4(eh, ok ? (uint32_t)gm * (uint32_t)(g.eldh * 4) + hcol4 : BX_OOR, rs_h);
      if constexpr (EPI == 2) BX_LOAD4(ez, ok ? (uint32_t)gm * (uint32_t)(g.eO * 8) + hcol4 : BX_OOR, rs_z);
    }
  };
  // NY: loads of this wavefront that are younger than its gate operands when they are due (a producer: the EPT loads of the
  // block after next; everyone else: none)
  auto row_epilogue = [&](int b, auto NY) {
    constexpr int ny = decltype(NY)::value;
    BX_TRACE(5);
    if constexpr (NOP > 0) {
      if (!BX_LAB_SKIP(32)) {
        BX_WAIT(ny, eh);
        if constexpr (EPI == 2) BX_WAIT(ny, ez);
      }
    }
    const int gm = b * BM + 4 * wave + r4;
    const bool row_ok = gm < g.M;
#pragma unroll
    for (int s = 0; s < CPL; ++s) {
      const bool ok = row_ok && live[s] && !BX_LAB_SKIP(4);
      // (the gate products write one segment, c_seg_n = N: the quad's offset is its column, no register kept for it)
      const uint32_t cq = EPI != 0 ? (uint32_t)((64 * s + 4 * q4) * 4) : ccol[s];
      const uint32_t co = ok ? (uint32_t)gm * (uint32_t)(g.ldc * 4) + cq : BX_OOR;
      const int col0 = 64 * s + 4 * q4;
      const float4 v = *reinterpret_cast<const float4*>(stage_part + (col0 >> 5) * PART + ((4 * wave + r4) * 32 + (col0 & 31)) * 4);
      if (s == CPL - 1 && lane == 0) tile_seen[wave] = n_iter + 1;   // after the last read of the block's sums: a wavefront's
                                                                     // LDS operations complete in order
      float x[4] = {v.x, v.y, v.z, v.w};
      if constexpr (EPI == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = bx_sigmoidf(x[i]);
      } else if constexpr (EPI == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = bx_tanhf(x[i]);
      }
      const bx_u32x4 xo = {bx_as_uint(x[0]), bx_as_uint(x[1]), bx_as_uint(x[2]), bx_as_uint(x[3])};
      BX_STORE4(xo, co, rs_c);
      if constexpr (EPI != 0) {
        if (s == OPG) {
          const bool sok = ok && olive;
          bx_u32x4 so;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float h = bx_as_float(eh[i]);
            so[i] = bx_as_uint(EPI == 1 ? h * x[i] : pgt_gru_blend(bx_as_float(ez[i]), h, x[i]));
          }
          if constexpr (EPI == 1) {
            BX_STORE4(so, sok ? (uint32_t)gm * (uint32_t)(g.eldx * 4) + hcol4 : BX_OOR, rs_x);
          } else {
            // out0 in a two-level row layout (pgt_rowmap: H_t straight into the [B, T, N, O] result)
            const uint32_t o0 = (uint32_t)pgt_row_off(sok ? gm : 0, g.eld0, g.e0_period, g.e0_hi);
            BX_STORE4(so, sok ? o0 * 4u + hcol4 : BX_OOR, rs_0);
            BX_STORE4(so, sok ? (uint32_t)gm * (uint32_t)(g.eld1 * 4) + hcol4 : BX_OOR, rs_1);
          }
        }
      }
    }
    BX_FENCE();
    e_issue_rows(b + nwg);                              // the next block's operands: behind every load of this iteration
    BX_TRACE(6);
  };
  // a partial-sum region is free again when its reader has added the previous block's sums (part_seen) and — it carries
  // the block's sums between the two barriers — every wavefront has taken its rows (tile_seen)
  auto wait_regions_free = [&]() {
    while (part_seen[cb] != n_iter) { BX_YIELD(); }
    if constexpr (ROWS) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        while (tile_seen[q] != n_iter) { BX_YIELD(); }
    }
  };

  if (producer) {
    // ---- element map of a 32-row block over the 256 producer threads: row = ptid / 8, pairs (ptid % 8) + 8 t
    const int ptid = tid - 256, erow = ptid >> 3, el = ptid & 7;
    const int half = g.seg_k >> 1, rpairs = g.n_seg * half;
    uint32_t goff[EPT];   // byte offset from the block base; past the row's last pair: outside the descriptor (reads 0)
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
      const int pi = el + 8 * t, seg = pi / half, pp = pi - seg * half;
      goff[t] = pi < rpairs ? (uint32_t)((seg * g.a_seg_stride + erow * g.lda + 2 * pp) * 4) : 0xfffffff0u;
    }
    const uint32_t lbase = (uint32_t)(erow * SROW + el * 4);
    // a block's rows are read through a buffer descriptor that ends with the last valid row of the last segment: rows
    // past M (ragged last block) and whole blocks past the end read as zero, without a branch
    const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
    auto block_rsrc = [&](int b) {
      const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
      const int64_t rows = rows_left < BM ? rows_left : BM;
      const int64_t bytes = rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0;
      return bx_make_rsrc(g.A + (int64_t)(rows_left > 0 ? b : 0) * BM * g.lda, bytes);
    };
    // Loads return in order and every conversion is followed by the reload of its register pair, so exactly EPT - 1
    // younger loads of A are in flight when element t of the previous round is due — plus, in the steady state, the NOP
    // gate-operand loads issued between the two rounds (at the end of the previous block's epilogue).
    bx_u32x2 raw[EPT];
    auto issue_load = [&](int t, const BxRsrc& r) {
      if (BX_LAB_SKIP(8)) return;
      BX_LOAD2(raw[t], goff[t], r);
    };
    auto convert_one = [&](int t, unsigned char* buf, auto NYOUNG) {
      uint32_t p1, p2, p3;
      BX_WAIT(decltype(NYOUNG)::value, raw[t]);
      bx_split2_fast(bx_as_float(raw[t][0]), bx_as_float(raw[t][1]), p1, p2, p3);
      unsigned char* d = buf + lbase + 32 * t;
      *reinterpret_cast<uint32_t*>(d) = p1;
      *reinterpret_cast<uint32_t*>(d + PLANE) = p2;
      *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = p3;
    };
    {
      const BxRsrc r0 = block_rsrc(rb), r1 = block_rsrc(rb + nwg);
#pragma unroll
      for (int t = 0; t < EPT; ++t) issue_load(t, r0);
#pragma unroll
      for (int t = 0; t < EPT; ++t) {
        convert_one(t, lds, BxInt<EPT - 1>{});
        issue_load(t, r1);
      }
      e_issue_rows(rb);
    }
    BX_SETPRIO(1);        // the younger half of the workgroup loses the VALU arbitration otherwise
    bx_barrier();
    int cur = 0;
    for (; rb < n_blocks; rb += nwg, ++n_iter) {
      BX_TRACE(0);
      unsigned char* bcur = lds + cur * BUF;
      unsigned char* bnxt = lds + (cur ^ 1) * BUF;
      const BxRsrc r2 = block_rsrc(rb + 2 * nwg);
      pgt_f32x16 am[WN], ac[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[j][r] = 0.f; ac[j][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < KPR; ++i) {
        {
          bx_u32x4 fa[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + (kbase + i) * 32);
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (BX_LAB_SKIP(1)) continue;
            am[j] = bx_mfma(fa[0], bf[i][j][0], am[j]);
            ac[j] = bx_mfma(fa[0], bf[i][j][1], ac[j]);
            ac[j] = bx_mfma(fa[1], bf[i][j][0], ac[j]);
            ac[j] = bx_mfma(fa[1], bf[i][j][1], ac[j]);
            ac[j] = bx_mfma(fa[0], bf[i][j][2], ac[j]);
            ac[j] = bx_mfma(fa[2], bf[i][j][0], ac[j]);
          }
        }
        // this k-step's share of the next block: fp32 (in registers since the previous iteration) -> bf16 planes in the
        // other buffer, and the load of the block after it into the freed registers
        {
#pragma unroll
          for (int t = i * EPT / KPR; t < (i + 1) * EPT / KPR; ++t) {
            convert_one(t, bnxt, BxInt<EPT - 1 + NOP>{});
            issue_load(t, r2);
          }
        }
      }
      BX_TRACE(1);
      wait_regions_free();
      BX_TRACE(2);
      {
        float4* d = reinterpret_cast<float4*>(stage_part + ((part - 1) * NCOL + cb) * PART);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            d[(j * 4 + r4) * 64 + lane] = make_float4(am[j][4 * r4] + ac[j][4 * r4], am[j][4 * r4 + 1] + ac[j][4 * r4 + 1],
                                                      am[j][4 * r4 + 2] + ac[j][4 * r4 + 2], am[j][4 * r4 + 3] + ac[j][4 * r4 + 3]);
      }
      BX_TRACE(3);
      bx_barrier();      // partial sums visible; everyone is done with this block's planes and the next block's are complete
      BX_TRACE(4);
      if constexpr (ROWS) {
        bx_barrier();    // the block's sums are in LDS
        row_epilogue(rb, BxInt<EPT>{});
      }
      cur ^= 1;
    }
    BX_DRAIN();
  } else if (Q4 && part != 0) {
    // ---- compute-only consumers (Q4): their part of K, then the partial sums, like a producer without a block to fetch
    e_issue_rows(rb);
    bx_barrier();
    int cur = 0;
    for (; rb < n_blocks; rb += nwg, ++n_iter) {
      unsigned char* bcur = lds + cur * BUF;
      pgt_f32x16 am[WN], ac[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[j][r] = 0.f; ac[j][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < KC1; ++i) {
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + (kbase + i) * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (BX_LAB_SKIP(1)) continue;
          am[j] = bx_mfma(fa[0], bf[i][j][0], am[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][0], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][2], ac[j]);
          ac[j] = bx_mfma(fa[2], bf[i][j][0], ac[j]);
        }
      }
      wait_regions_free();
      {
        float4* d = reinterpret_cast<float4*>(stage_part + ((part - 1) * NCOL + cb) * PART);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            d[(j * 4 + r4) * 64 + lane] = make_float4(am[j][4 * r4] + ac[j][4 * r4], am[j][4 * r4 + 1] + ac[j][4 * r4 + 1],
                                                      am[j][4 * r4 + 2] + ac[j][4 * r4 + 2], am[j][4 * r4 + 3] + ac[j][4 * r4 + 3]);
      }
      bx_barrier();
      if constexpr (ROWS) {
        bx_barrier();
        row_epilogue(rb, BxInt<0>{});
      }
      cur ^= 1;
    }
    BX_DRAIN();
  } else {
    const int lo = lane & 31, hi = lane >> 5;
    float bias_r[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gn = (cb * WN + j) * 32 + lo;
      bias_r[j] = (g.bias && gn < g.N) ? g.bias[gn] : 0.f;
    }
    const bool cols_live = cb * WN * 32 < g.N;           // N <= 96: the last column wavefronts only keep the barriers company
    e_issue_rows(rb);
    bx_barrier();
    int cur = 0;
    for (; rb < n_blocks; rb += nwg, ++n_iter) {
      BX_TRACE(0);
      unsigned char* bcur = lds + cur * BUF;
      pgt_f32x16 am[WN], ac[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[j][r] = 0.f; ac[j][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < KC0; ++i) {
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + i * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (BX_LAB_SKIP(1)) continue;
          am[j] = bx_mfma(fa[0], bf[i][j][0], am[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][0], ac[j]);
          ac[j] = bx_mfma(fa[1], bf[i][j][1], ac[j]);
          ac[j] = bx_mfma(fa[0], bf[i][j][2], ac[j]);
          ac[j] = bx_mfma(fa[2], bf[i][j][0], ac[j]);
        }
      }
      float acc[WN][16];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = am[j][r] + ac[j][r] + bias_r[j];
      BX_TRACE(1);
      bx_barrier();
      BX_TRACE(2);
      // ---- the other parts' partial sums join in registers (accumulator layout)
#pragma unroll
      for (int p = 1; p < NPART; ++p) {
        const float4* d = reinterpret_cast<const float4*>(stage_part + ((p - 1) * NCOL + cb) * PART);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = d[(j * 4 + r4) * 64 + lane];
            acc[j][4 * r4] += v.x; acc[j][4 * r4 + 1] += v.y; acc[j][4 * r4 + 2] += v.z; acc[j][4 * r4 + 3] += v.w;
          }
      }
      if (lane == 0) part_seen[cb] = n_iter + 1;     // after the reads above: a wavefront's LDS operations complete in order
      if (cols_live) {
        // a non-finite operand shows as nan sums: redo the tile in exact fp32 (bx_exact_tile; rare)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (bx_tile_has_nan(acc[j])) {
            BX_DRAIN();
            const int64_t rows_left = (int64_t)g.M - (int64_t)rb * BM, rows = rows_left < BM ? rows_left : BM;
            const BxRsrc ra = bx_make_rsrc(g.A + (int64_t)rb * BM * g.lda, (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4 +
                                           (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4);
            bx_exact_tile(g, ra, (cb * WN + j) * 32 + lo, hi, bias_r[j], acc[j]);
          }
      }
      if constexpr (ROWS) {
        // the block's sums -> LDS, [column block][row][32] in the region its own part-1 sums came through (this wavefront
        // has just read them); register r of lane (lo, hi) is row (r & 3) + 8 (r >> 2) + 4 hi, column lo
        float* tile = reinterpret_cast<float*>(stage_part + cb * PART);
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo] = acc[0][r];
        BX_TRACE(3);
        bx_barrier();
        BX_TRACE(4);
        row_epilogue(rb, BxInt<0>{});
      } else if (cols_live) {
        // two column blocks per wavefront (plain products with K <= 128): stored straight from the accumulator layout, a
        // register is one 128-byte row piece per half-wavefront
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int gn = (cb * WN + j) * 32 + lo;
          if (gn >= g.N) continue;
          const int js = gn / g.c_seg_n;
          float* cp = g.C + (int64_t)js * g.c_seg_stride + (gn - js * g.c_seg_n);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int gm = rb * BM + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (gm >= g.M) continue;
            cp[(int64_t)gm * g.ldc] = acc[j][r];
          }
        }
      }
      cur ^= 1;
    }
    BX_DRAIN();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[k, n] += sum_m A[m, k] G[m, n], db[n] += sum_m G[m, n] (K + 1 <= 352 rows of dW: row K is the bias
// gradient — A^T carries a row of ones there — N <= 32 NCB columns).  BOTH operands stream, 16 rows per stage; both are
// cut into three bf16 planes and stored TRANSPOSED in LDS ([column][row], 48-byte rows: `ds_read_b128` of eight
// consecutive rows of one column is conflict-free), which is the MFMA operand layout for a product that contracts over
// rows.  A lane converts (column, row pair) units: the two floats of a pair are exactly what v_cvt_pk_bf16_f32 packs into
// the dword LDS wants.  One persistent 512-thread workgroup per CU, all wavefronts alike: wavefront w owns the 32-column
// block w % NCB of G and every (8 / NCB)-th 32-row block of dW (six 32 x 32 accumulators at NCB = 4); double-buffered
// stages, one LDS-only barrier per stage, hand-issued loads / hand-counted waits as above (two loads per unit:
// vmcnt(2 (EPT - 1))).  A launch covers at most ~5 000 rows per workgroup (the host cuts taller operands into several
// launches): the rounding error of a running sum grows with its length times its size, and that is what the fp32 kernels'
// slabs hold (measured at M = 2.5 M rows against fp64: 6.5e-3 of a scale of 1 100 in one launch, 2 - 3e-3 in two — the
// fp32 kernels: 2e-3).  The sums leave through atomics, or as the deterministic mode's partial-sum slabs.  Measured (lab/gemm_bx_tn_lab.hip, M = 2 543 616, K = 330): N = 128: 1.50 ms against 2.27 ms for the
// fp32 MFMA kernel, N = 64: 1.10 against 1.47 ms.
template <int NCB, bool DET>
__global__ __launch_bounds__(512, 1) void gemm_bx_tn_kernel(PgtTnArgs g, int n_stages, int slab_base) {
  constexpr int RB = 11, ROWB = 48, APL = RB * 32 * ROWB, GPL = NCB * 32 * ROWB, BUF = 3 * (APL + GPL);
  constexpr int RSTEP = 8 / NCB, MAXB = (RB + RSTEP - 1) / RSTEP;
  constexpr int EPT_A = 6, EPT_G = NCB * 32 * 8 / 512, EPT = EPT_A + EPT_G;     // (column, row pair) units per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = BX_SGPR(tid >> 6);
  const int cb = wave % NCB, r0 = wave / NCB;
  const int nwg = gridDim.x;
  const int K = g.n_seg * g.seg_k;
  // ---- LDS: zeros, then the row of ones at column K of A^T (first plane; 1.0 = 0x3f80)
  for (int i = tid; i < 2 * BUF / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 16) reinterpret_cast<uint32_t*>(lds + (tid >> 3) * BUF + K * ROWB)[tid & 7] = 0x3f803f80u;
  // ---- unit map: unit = (column, row pair); lanes along the columns
  uint32_t goff[EPT];   // byte offset of the unit's first row from the stage base (A or G)
  uint32_t loff[EPT];   // byte offset of the unit's dword inside a buffer, first plane
#pragma unroll
  for (int t = 0; t < EPT_A; ++t) {
    const int u = tid + 512 * t;
    if (u < K * 8) {
      const int rp = u / K, c = u - rp * K, seg = c / g.seg_k, cc = c - seg * g.seg_k;
      goff[t] = (uint32_t)((seg * g.a_seg_stride + 2 * rp * g.lda + cc) * 4);
      loff[t] = (uint32_t)(c * ROWB + rp * 4);
    } else {
      goff[t] = 0xfffffff0u;                              // outside the descriptor: reads zero ...
      loff[t] = (uint32_t)((RB * 32 - 1) * ROWB + 32);    // ... and lands in the padding of the last row
    }
  }
#pragma unroll
  for (int t = 0; t < EPT_G; ++t) {
    const int u = tid + 512 * t, rp = u / (NCB * 32), c = u - rp * (NCB * 32);
    goff[EPT_A + t] = c < g.N ? (uint32_t)((2 * rp * g.ldg + c) * 4) : 0xfffffff0u;
    loff[EPT_A + t] = (uint32_t)(3 * APL + c * ROWB + rp * 4);
  }
  const uint32_t lda4 = (uint32_t)(g.lda * 4), ldg4 = (uint32_t)(g.ldg * 4);
  auto rsrc = [&](const float* p, int64_t bytes) {
    return bx_make_rsrc(p, bytes > 0 ? bytes : 0);
  };
  const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
  auto a_rsrc = [&](int st) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)st * 16;
    const int64_t rows = rows_left < 16 ? rows_left : 16;
    return rsrc(g.A + (int64_t)(rows_left > 0 ? st : 0) * 16 * g.lda,
                rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0);
  };
  auto g_rsrc = [&](int st) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)st * 16;
    const int64_t rows = rows_left < 16 ? rows_left : 16;
    return rsrc(g.G + (int64_t)(rows_left > 0 ? st : 0) * 16 * g.ldg, rows_left > 0 ? (rows - 1) * g.ldg * 4 + (int64_t)g.N * 4 : 0);
  };
  float raw0[EPT], raw1[EPT];
  auto issue = [&](int t, const BxRsrc& ra, const BxRsrc& rg) {
    if (t < EPT_A) {
      BX_LOAD1(raw0[t], goff[t], ra);
      BX_LOAD1S(raw1[t], goff[t], ra, lda4);
    } else {
      BX_LOAD1(raw0[t], goff[t], rg);
      BX_LOAD1S(raw1[t], goff[t], rg, ldg4);
    }
  };
  // rows_left: valid rows of the stage being converted.  A rows past M inside the earlier segments are other data, not
  // zeros (the descriptor only ends the LAST segment): masked here; G rows past M read zero through the descriptor.
  auto convert = [&](int t, unsigned char* buf, int rows_left) {
    BX_WAIT2(2 * (EPT - 1), raw0[t], raw1[t]);
    float x = raw0[t], y = raw1[t];
    if (rows_left < 16 && t < EPT_A) {
      const int m0 = 2 * (int)((loff[t] % ROWB) >> 2);
      x = m0 < rows_left ? x : 0.f;
      y = m0 + 1 < rows_left ? y : 0.f;
    }
    uint32_t p1, p2, p3;
    bx_split2_fast(x, y, p1, p2, p3);
    unsigned char* d = buf + loff[t];
    const int pl = t < EPT_A ? APL : GPL;
    *reinterpret_cast<uint32_t*>(d) = p1;
    *reinterpret_cast<uint32_t*>(d + pl) = p2;
    *reinterpret_cast<uint32_t*>(d + 2 * pl) = p3;
  };
  pgt_f32x16 acc[MAXB];
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int lo = lane & 31, hi = lane >> 5;
  const int n = cb * 32 + lo;
  // one 32 x 32 block of dW (rows rb * 32 ..) leaves through atomics or as this workgroup's deterministic slab
  auto flush_block = [&](int rb, const auto& v) {
    const int slab = slab_base + (int)blockIdx.x;
    float* const wbase = (DET ? g.part + (int64_t)slab * g.part_stride : g.dW) + n;
    float* const bbase = g.db == nullptr ? nullptr : (DET ? g.dbpart + (int64_t)slab * g.N : g.db) + n;
    const uint32_t ld = (uint32_t)g.lddw;
    if (rb >= RB || n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (k < K) {
        if constexpr (DET) wbase[(uint32_t)k * ld] = v[r];
        else atomicAdd(wbase + (uint32_t)k * ld, v[r]);
      } else if (k == K && bbase != nullptr) {
        if constexpr (DET) *bbase = v[r];
        else atomicAdd(bbase, v[r]);
      }
    }
  };
  int st = blockIdx.x;
  __syncthreads();
  {
    const BxRsrc ra0 = a_rsrc(st), rg0 = g_rsrc(st), ra1 = a_rsrc(st + nwg), rg1 = g_rsrc(st + nwg);
#pragma unroll
    for (int t = 0; t < EPT; ++t) issue(t, ra0, rg0);
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
      convert(t, lds, g.M - st * 16);
      issue(t, ra1, rg1);
    }
  }
  bx_barrier();
  int cur = 0;
  const int afrag = (lane & 31) * ROWB + 16 * (lane >> 5);
  for (; st < n_stages; st += nwg) {
    unsigned char* bcur = lds + cur * BUF;
    unsigned char* bnxt = lds + (cur ^ 1) * BUF;
    const BxRsrc ra2 = a_rsrc(st + 2 * nwg), rg2 = g_rsrc(st + 2 * nwg);
    const int rows_next = g.M - (st + nwg) * 16;
    bx_u32x4 fb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bx_u32x4*>(bcur + 3 * APL + q * GPL + cb * 32 * ROWB + afrag);
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      const int rb = r0 + RSTEP * b;
      if (rb < RB) {
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * APL + rb * 32 * ROWB + afrag);
        acc[b] = bx_mfma(fa[2], fb[0], acc[b]);      // small piece products first
        acc[b] = bx_mfma(fa[0], fb[2], acc[b]);
        acc[b] = bx_mfma(fa[1], fb[1], acc[b]);
        acc[b] = bx_mfma(fa[1], fb[0], acc[b]);
        acc[b] = bx_mfma(fa[0], fb[1], acc[b]);
        acc[b] = bx_mfma(fa[0], fb[0], acc[b]);
      }
#pragma unroll
      for (int t = b * EPT / MAXB; t < (b + 1) * EPT / MAXB; ++t) {
        convert(t, bnxt, rows_next);
        issue(t, ra2, rg2);
      }
    }
    bx_barrier();
    cur ^= 1;
  }
  BX_DRAIN();
  // ---- non-finite operands (see bx_exact_tile): a nan among this workgroup's sums -> its whole slab again as an fp32
  // fmaf chain over the original operands, so that the +-inf / nan of the reference's dW = A^T G land where they belong
  {
    bool bad = false;
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) bad |= acc[b][r] != acc[b][r];
    if (__ballot(bad) != 0) {
      // the fast sums are dropped here, so the redo needs no register beyond one block's
      for (int b = 0; b < MAXB; ++b) {
        const int rb = r0 + RSTEP * b;
        if (rb >= RB) continue;
        int koff[16];                                      // element offset of column k inside a row of A; -1: the ones; -2: none
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const int seg = k / g.seg_k;
          koff[r] = k > K ? -2 : k == K ? -1 : (int)(seg * g.a_seg_stride + (k - seg * g.seg_k));
          t[r] = 0.f;
        }
        for (int s2 = blockIdx.x; s2 < n_stages; s2 += nwg)
          for (int i = 0; i < 16; ++i) {
            const int64_t m = (int64_t)s2 * 16 + i;
            if (m >= g.M) break;
            const float gv = n < g.N ? g.G[m * g.ldg + n] : 0.f;
            const float* ar = g.A + m * g.lda;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = koff[r];
              const float a = o >= 0 ? ar[o] : (o == -1 ? 1.f : 0.f);
              t[r] = fmaf(a, gv, t[r]);
            }
          }
        flush_block(rb, t);
      }
      return;
    }
  }
#pragma unroll
  for (int b = 0; b < MAXB; ++b) flush_block(r0 + RSTEP * b, acc[b]);
}

// ---------------------------------------------------------------------------------------------------------------------
// The weight gradient with the wavefronts SPECIALISED (round 5; pgt_tune("gemm_bx_tn_pc", 1)).  In gemm_bx_tn_kernel all eight
// wavefronts are alike — matrix products, then conversion of the next stage, then a barrier — and a stage costs the SUM of those
// phases: matrix pipe 52 % busy, LDS 50 %, HBM 3.7 TB/s, nothing saturated (notebook 5.7).  Here
//   * wavefronts 0 .. 3 ("consumers", one per SIMD) do nothing but LDS fragment reads and MFMAs: wavefront w owns column block w
//     of G and ALL eleven 32-row blocks of dW (NCB = 4; 176 accumulator registers), or column block w & 1 and every second row
//     block (NCB = 2) — 66 back-to-back MFMAs per stage and SIMD, 2 112 cycles: the floor of the matrix pipe;
//   * wavefronts 4 .. 7 ("producers", one per SIMD) never issue an MFMA: they stream both operands (hand-issued 4-byte loads,
//     lanes along the columns, 32 in flight per thread), cut them into three bf16 planes and write the transposed LDS image of
//     the NEXT stage — their VALU / LDS / VMEM instructions issue in the shadow of the consumers' MFMAs.
// A producer's unit is (column, EIGHT rows): the four dwords of a plane leave as one ds_write_b128 at c * 48 + 16 q, which is
// conflict-free (sixteen consecutive columns x four dwords = the 64 banks once), where one-dword writes meet four ways — with the
// LDS pipe now shared by four writers and four readers at full rate that is the difference between 1 500 and 2 600 LDS cycles per
// stage.  Same planes, same six piece products per block in the same order as gemm_bx_tn_kernel: the same sums bit for bit.
// One LDS-only barrier per stage, double-buffered stages, the non-finite redo and the flush on the consumers.
// NCW consumer + NPW producer wavefronts.  4 + 4 is the form described above.  Twelve wavefronts (168 registers) put the extra four
// where a shape is short: N <= 64 has half the MFMAs per stage, so its stage is paced by ONE producer wavefront per SIMD walking
// its dependent convert chains — 4 + 8 (one unit per producer thread); N = 128 is paced by the matrix pipe with one consumer per
// SIMD waiting out its fragment reads — 8 + 4 (two consumers per SIMD, half the row blocks each).
template <int NCB, bool DET, int NCW, int NPW>
__global__ __launch_bounds__(64 * (NCW + NPW), 1) void gemm_bx_tn_pc_kernel(PgtTnArgs g, int n_stages, int slab_base) {
  constexpr int RB = 11, ROWB = 48, APL = RB * 32 * ROWB, GPL = NCB * 32 * ROWB, BUF = 3 * (APL + GPL);
  constexpr int NT = 64 * (NCW + NPW);
  constexpr int RSTEP = NCW / NCB, MAXB = (RB + RSTEP - 1) / RSTEP;      // row blocks of a consumer: every RSTEP-th
  constexpr int UPT = NPW == 4 ? 2 : 1;                                // (column pair, eight rows) units per producer thread and stage
  static_assert((NPW == 4 || NPW == 8) && NCW % NCB == 0, "wavefront roles");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = BX_SGPR(tid >> 6);
  const bool producer = wave >= NCW;
  const int nwg = gridDim.x;
  const int K = g.n_seg * g.seg_k;
  for (int i = tid; i < 2 * BUF / 16; i += NT) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 16) reinterpret_cast<uint32_t*>(lds + (tid >> 3) * BUF + K * ROWB)[tid & 7] = 0x3f803f80u;   // the row of ones (bias gradient)
  __syncthreads();
  int st = blockIdx.x;
  if (producer) {
    // ---- unit map (256 threads).  A unit = (column PAIR, eight rows): eight 8-byte loads (a wavefront's instruction reads 512
    // contiguous bytes of a row), two columns x four row pairs -> two quads per plane.  Unit 0 of thread p: A unit p; unit 1: A
    // unit 256 + p on producer wavefronts 0, 1, G unit p - 128 on wavefronts 2, 3 (uniform per wavefront: descriptor and row
    // pitch of a hand-issued load are scalar).  K and N even, operands 8-byte aligned (the host side checks).
    // (eight producer wavefronts: ONE unit per thread — A unit p on wavefronts 0 - 5, G unit p - 384 on wavefronts 6, 7)
    const int p = tid - 64 * NCW, pw = wave - NCW;
    const int KH = K >> 1;
    uint32_t ug[UPT], ul[UPT];          // byte offset of the unit's first element in its operand's stage; of its first quad in a buffer
    const bool g1 = NPW == 4 ? pw >= 2 : pw >= 6;      // the LAST unit of this wavefront is a G unit
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      if (j == UPT - 1 && g1) {
        const int w = NPW == 4 ? p - 128 : p - 384;
        const int q = w / (NCB * 16), c = 2 * (w - q * (NCB * 16));
        const bool unit = w < NCB * 32;
        ug[j] = unit && c < g.N ? (uint32_t)((8 * q * g.ldg + c) * 4) : 0xfffffff0u;
        ul[j] = unit ? (uint32_t)(3 * APL + c * ROWB + 16 * q) : (uint32_t)((RB * 32 - 2) * ROWB + 32);
      } else {
        const int v = 256 * j + p;
        const int q = v / KH, c = 2 * (v - q * KH), seg = c / g.seg_k, cc = c - seg * g.seg_k;
        const bool unit = v < K;
        ug[j] = unit ? (uint32_t)((seg * g.a_seg_stride + 8 * q * g.lda + cc) * 4) : 0xfffffff0u;
        ul[j] = unit ? (uint32_t)(c * ROWB + 16 * q) : (uint32_t)((RB * 32 - 2) * ROWB + 32);      // (no unit: the padding of the last two rows)
      }
    }
    const uint32_t lda4 = (uint32_t)(g.lda * 4), ldg4 = (uint32_t)(g.ldg * 4);
    const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
    auto a_rsrc = [&](int s) {
      const int64_t rows_left = (int64_t)g.M - (int64_t)s * 16;
      const int64_t rows = rows_left < 16 ? rows_left : 16;
      return bx_make_rsrc(g.A + (int64_t)(rows_left > 0 ? s : 0) * 16 * g.lda,
                          rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0);
    };
    auto g_rsrc = [&](int s) {
      const int64_t rows_left = (int64_t)g.M - (int64_t)s * 16;
      const int64_t rows = rows_left < 16 ? rows_left : 16;
      return bx_make_rsrc(g.G + (int64_t)(rows_left > 0 ? s : 0) * 16 * g.ldg, rows_left > 0 ? (rows - 1) * g.ldg * 4 + (int64_t)g.N * 4 : 0);
    };
    // TWO stages of loads in flight per thread (2 x 16 loads of 8 bytes = 64 KB per CU): with one (32 KB) the stream is bound by
    // latency x bytes in flight at ~4 TB/s whatever the wavefronts do meanwhile (the first form of this kernel: 1 353 us against
    // 1 433 for the kernel it replaces — and the reason that one never passed 3.7 TB/s).  The two register sets take turns: the
    // loop below is unrolled by two, so every register has one name.
    bx_u32x2 raw[2][UPT][8];
    auto issue_unit = [&](int set, int j, const BxRsrc& ra, const BxRsrc& rg) {
      const bool gu = j == UPT - 1 && g1;
      const BxRsrc ru = bx_select_rsrc(gu, rg, ra);
      const uint32_t ld = (uint32_t)BX_SGPR((int)(gu ? ldg4 : lda4));
      BX_LOAD2(raw[set][j][0], ug[j], ru);
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        const uint32_t so = (uint32_t)BX_SGPR((int)(r * ld));
        BX_LOAD2S(raw[set][j][r], ug[j], ru, so);
      }
    };
    // unit j of `set`, whose eight loads are the OLDEST in flight: 8 (2 UPT - 1) younger ones behind them
    auto convert_unit = [&](int set, int j, unsigned char* buf, int rows_left) {
      BX_WAIT8(8 * (2 * UPT - 1), raw[set][j][0], raw[set][j][1], raw[set][j][2], raw[set][j][3], raw[set][j][4], raw[set][j][5],
               raw[set][j][6], raw[set][j][7]);
      const int q8 = 8 * (int)((ul[j] % ROWB) >> 4);          // first row of the unit inside the stage
      const int pl = (j == UPT - 1 && g1) ? GPL : APL;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                           // the unit's two columns
        uint32_t hold[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = bx_as_float(raw[set][j][2 * i][h]), y = bx_as_float(raw[set][j][2 * i + 1][h]);
          if (rows_left < 16) {                                // (rows past M: other data in A's earlier segments)
            x = q8 + 2 * i < rows_left ? x : 0.f;
            y = q8 + 2 * i + 1 < rows_left ? y : 0.f;
          }
          bx_split2_fast(x, y, hold[0][i], hold[1][i], hold[2][i]);
        }
        unsigned char* d = buf + ul[j] + h * ROWB;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const bx_u32x4 v4 = {hold[q][0], hold[q][1], hold[q][2], hold[q][3]};
          *reinterpret_cast<bx_u32x4*>(d + q * pl) = v4;
        }
      }
    };
    // stage s of this workgroup's sequence: st + s * nwg
    {
      const BxRsrc ra0 = a_rsrc(st), rg0 = g_rsrc(st), ra1 = a_rsrc(st + nwg), rg1 = g_rsrc(st + nwg);
      const BxRsrc ra2 = a_rsrc(st + 2 * nwg), rg2 = g_rsrc(st + 2 * nwg);
#pragma unroll
      for (int j = 0; j < UPT; ++j) issue_unit(0, j, ra0, rg0);
#pragma unroll
      for (int j = 0; j < UPT; ++j) issue_unit(1, j, ra1, rg1);
#pragma unroll
      for (int j = 0; j < UPT; ++j) {
        convert_unit(0, j, lds, g.M - st * 16);
        issue_unit(0, j, ra2, rg2);
      }
    }
    bx_barrier();
    // iteration i consumes stage i out of buffer i & 1; the producers meanwhile convert stage i + 1 (register set (i + 1) & 1) into
    // the other buffer and request stage i + 3 into the same set
    // S stages for this workgroup (>= 1: the grid never has more workgroups than stages), i.e. S more barriers: whole rounds of
    // two, then the odd one BEHIND the loop — the loop has one exit and one back edge, so the two halves' in-flight orders never
    // meet at its header (scripts/bx_isa_audit.py follows the control-flow graph, not the values of the exit tests).
    const int S = (n_stages - st + nwg - 1) / nwg;
    auto half = [&](int set, unsigned char* buf, int stage_conv, int stage_load) {
      const BxRsrc ran = a_rsrc(stage_load), rgn = g_rsrc(stage_load);
#pragma unroll
      for (int j = 0; j < UPT; ++j) {
        convert_unit(set, j, buf, g.M - stage_conv * 16);
        issue_unit(set, j, ran, rgn);
      }
      bx_barrier();
    };
    for (int r = 0; r < (S >> 1); ++r) {
      half(1, lds + BUF, st + nwg, st + 3 * nwg);
      half(0, lds, st + 2 * nwg, st + 4 * nwg);
      st += 2 * nwg;
    }
    if (S & 1) half(1, lds + BUF, st + nwg, st + 3 * nwg);
    BX_DRAIN();
    return;
  }
  // ---- consumers
  const int cb = wave % NCB, r0 = wave / NCB;
  pgt_f32x16 acc[MAXB];
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int lo = lane & 31, hi = lane >> 5;
  const int n = cb * 32 + lo;
  auto flush_block = [&](int rb, const auto& v) {
    const int slab = slab_base + (int)blockIdx.x;
    float* const wbase = (DET ? g.part + (int64_t)slab * g.part_stride : g.dW) + n;
    float* const bbase = g.db == nullptr ? nullptr : (DET ? g.dbpart + (int64_t)slab * g.N : g.db) + n;
    const uint32_t ld = (uint32_t)g.lddw;
    if (rb >= RB || n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (k < K) {
        if constexpr (DET) wbase[(uint32_t)k * ld] = v[r];
        else atomicAdd(wbase + (uint32_t)k * ld, v[r]);
      } else if (k == K && bbase != nullptr) {
        if constexpr (DET) *bbase = v[r];
        else atomicAdd(bbase, v[r]);
      }
    }
  };
  bx_barrier();                                           // stage 0 is in buffer 0
  int cur = 0;
  const int afrag = (lane & 31) * ROWB + 16 * (lane >> 5);
  for (; st < n_stages; st += nwg) {
    unsigned char* bcur = lds + cur * BUF;
    bx_u32x4 fb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bx_u32x4*>(bcur + 3 * APL + q * GPL + cb * 32 * ROWB + afrag);
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      const int rb = r0 + RSTEP * b;
      if (rb < RB) {
        bx_u32x4 fa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * APL + rb * 32 * ROWB + afrag);
        acc[b] = bx_mfma(fa[2], fb[0], acc[b]);      // small piece products first (the order of gemm_bx_tn_kernel)
        acc[b] = bx_mfma(fa[0], fb[2], acc[b]);
        acc[b] = bx_mfma(fa[1], fb[1], acc[b]);
        acc[b] = bx_mfma(fa[1], fb[0], acc[b]);
        acc[b] = bx_mfma(fa[0], fb[1], acc[b]);
        acc[b] = bx_mfma(fa[0], fb[0], acc[b]);
      }
    }
    bx_barrier();
    cur ^= 1;
  }
  // ---- non-finite operands: as in gemm_bx_tn_kernel (a nan among this wavefront's sums -> its blocks again as an fp32 fmaf chain)
  {
    bool bad = false;
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) bad |= acc[b][r] != acc[b][r];
    if (__ballot(bad) != 0) {
      for (int b = 0; b < MAXB; ++b) {
        const int rb = r0 + RSTEP * b;
        if (rb >= RB) continue;
        int koff[16];
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const int seg = k / g.seg_k;
          koff[r] = k > K ? -2 : k == K ? -1 : (int)(seg * g.a_seg_stride + (k - seg * g.seg_k));
          t[r] = 0.f;
        }
        for (int s2 = blockIdx.x; s2 < n_stages; s2 += nwg)
          for (int i = 0; i < 16; ++i) {
            const int64_t m = (int64_t)s2 * 16 + i;
            if (m >= g.M) break;
            const float gv = n < g.N ? g.G[m * g.ldg + n] : 0.f;
            const float* ar = g.A + m * g.lda;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = koff[r];
              const float a = o >= 0 ? ar[o] : (o == -1 ? 1.f : 0.f);
              t[r] = fmaf(a, gv, t[r]);
            }
          }
        flush_block(rb, t);
      }
      return;
    }
  }
#pragma unroll
  for (int b = 0; b < MAXB; ++b) flush_block(r0 + RSTEP * b, acc[b]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Short K (<= 128) into up to 320 columns — the feature-gradient products dP W^T of the training step: nothing to split
// along K, so all eight wavefronts are alike: wavefront w owns the 32-column block w for the whole K (B slice: KSTEPS x 3
// fragments in registers), every thread converts its share of the next 32-row block of A, no partial sums, no flag,
// one LDS-only barrier per block.  Loads AND stores are hand-issued buffer instructions (the descriptors end at the
// last valid row: ragged blocks need no branch), so that the number of younger instructions at every wait is known:
// EPT - 1 loads + the 16 stores of the previous block.  Column blocks 8 and 9 (N = 320 = five 64-wide stack segments:
// one product reads dP once instead of a 256-column product plus a 64-column remainder) go to wavefronts 0 and 1 as a
// second block whose B fragments wait in LDS in operand order.
template <int KSTEPS>
__global__ __launch_bounds__(512, 1) void gemm_bx_sym_kernel(PgtGemmArgs g, int n_blocks) {
  constexpr int BM = 32, KP = KSTEPS * 16, SROW = KP * 2 + 16, PLANE = BM * SROW, BUF = 3 * PLANE;
  constexpr int EPT = (KP / 2) / 16;                      // float pairs per thread and block (16 threads per row)
  constexpr int B2 = KSTEPS * 3 * 64 * 16;                // one extra column block's B fragments, operand order
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF + 2 * B2];
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int wave = BX_SGPR(tid >> 6);
  const int nwg = gridDim.x;
  const int Ktot = g.n_seg * g.seg_k;
  const int col = wave * 32 + lo;
  const bool live = wave * 32 < g.N;
  const bool two = (wave + 8) * 32 < g.N;                  // this wavefront owns a second column block (wave + 8)
  const int col2 = (wave + 8) * 32 + lo;
  bx_u32x4* const b2 = reinterpret_cast<bx_u32x4*>(lds + 2 * BUF + (wave & 1) * B2);
  bx_u32x4 bf[KSTEPS][3];
#pragma unroll
  for (int i = 0; i < KSTEPS; ++i) {
    const int k0 = i * 16 + 8 * hi;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = k0 + t;
      v[t] = (k < Ktot && col < g.N) ? g.Bw[(int64_t)k * g.sbk + (int64_t)col * g.sbn] : 0.f;
    }
    uint32_t p[3][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bx_split2(v[2 * t], v[2 * t + 1], p[0][t], p[1][t], p[2][t]);
#pragma unroll
    for (int q = 0; q < 3; ++q) { bx_u32x4 f = {p[q][0], p[q][1], p[q][2], p[q][3]}; bf[i][q] = f; }
  }
  for (int i = tid; i < 2 * BUF / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  if (two) {
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      const int k0 = i * 16 + 8 * hi;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = k0 + t;
        v[t] = (k < Ktot && col2 < g.N) ? g.Bw[(int64_t)k * g.sbk + (int64_t)col2 * g.sbn] : 0.f;
      }
      uint32_t p[3][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bx_split2(v[2 * t], v[2 * t + 1], p[0][t], p[1][t], p[2][t]);
#pragma unroll
      for (int q = 0; q < 3; ++q) { bx_u32x4 f = {p[q][0], p[q][1], p[q][2], p[q][3]}; b2[(i * 3 + q) * 64 + lane] = f; }
    }
  }
  int rb = blockIdx.x;
  if (rb >= n_blocks) return;
  __syncthreads();
  // ---- element map: row = tid / 16, pairs (tid % 16) + 16 t
  const int erow = tid >> 4, el = tid & 15;
  const int half = g.seg_k >> 1, rpairs = g.n_seg * half;
  uint32_t goff[EPT];
#pragma unroll
  for (int t = 0; t < EPT; ++t) {
    const int pi = el + 16 * t, seg = pi / half, pp = pi - seg * half;
    goff[t] = pi < rpairs ? (uint32_t)((seg * g.a_seg_stride + erow * g.lda + 2 * pp) * 4) : 0xfffffff0u;
  }
  const uint32_t lbase = (uint32_t)(erow * SROW + el * 4);
  auto rsrc = [&](const float* p, int64_t bytes) {
    return bx_make_rsrc(p, bytes > 0 ? bytes : 0);
  };
  const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
  auto a_rsrc = [&](int b) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
    const int64_t rows = rows_left < BM ? rows_left : BM;
    return rsrc(g.A + (int64_t)(rows_left > 0 ? b : 0) * BM * g.lda, rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0);
  };
  // the wavefront's output block: column segment js of C (c_seg_n is a multiple of 32), rows of block b
  const int js = (wave * 32) / g.c_seg_n;
  const float* cbase = g.C + (int64_t)js * g.c_seg_stride + (wave * 32 - js * g.c_seg_n);
  const uint32_t cvoff = col < g.N ? (uint32_t)((4 * hi * g.ldc + lo) * 4) : 0xfffffff0u;
  const int js2 = ((wave + 8) * 32) / g.c_seg_n;
  const float* cbase2 = g.C + (int64_t)js2 * g.c_seg_stride + ((wave + 8) * 32 - js2 * g.c_seg_n);
  const uint32_t cvoff2 = col2 < g.N ? (uint32_t)((4 * hi * g.ldc + lo) * 4) : 0xfffffff0u;
  auto c_rsrc = [&](const float* cb_, int b) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
    const int64_t rows = rows_left < BM ? rows_left : BM;
    return rsrc(cb_ + (int64_t)b * BM * g.ldc, ((rows - 1) * g.ldc + 32) * 4);
  };
  bx_u32x2 raw[EPT];
  auto issue_load = [&](int t, const BxRsrc& r) {
    BX_LOAD2(raw[t], goff[t], r);
  };
  // The wait for element t may only count the YOUNGER LOADS (the other EPT - 1): loads retire in order among themselves,
  // but a younger store can be acknowledged before an older load lands (vmcnt is one counter for both kinds, unordered
  // against each other), so the previous block's stores must not be added to the count.  (Round 2 did add them — 16 / 32
  // per block — and a launch at M = 211 968, K = 64 -> 320 columns then converted stale registers in about four rows of
  // 200 000: scripts/bx_sym_race_probe.py.)  The price: the wait also covers the acknowledgement of those stores.
  auto convert_one = [&](int t, unsigned char* buf) {
    BX_WAIT(EPT - 1, raw[t]);
    uint32_t p1, p2, p3;
    bx_split2_fast(bx_as_float(raw[t][0]), bx_as_float(raw[t][1]), p1, p2, p3);
    unsigned char* d = buf + lbase + 64 * t;
    *reinterpret_cast<uint32_t*>(d) = p1;
    *reinterpret_cast<uint32_t*>(d + PLANE) = p2;
    *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = p3;
  };
  {
    const BxRsrc r0 = a_rsrc(rb), r1 = a_rsrc(rb + nwg);
#pragma unroll
    for (int t = 0; t < EPT; ++t) issue_load(t, r0);
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
      convert_one(t, lds);
      issue_load(t, r1);
    }
  }
  const float bias_r = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  const float bias_r2 = (g.bias && col2 < g.N) ? g.bias[col2] : 0.f;
  BX_WAIT_PLAIN(EPT);      // the B / bias loads above are older than the EPT block loads
  bx_barrier();
  int cur = 0;
  const int arow = lo * SROW + 16 * hi;
  const uint32_t ldc4 = (uint32_t)(g.ldc * 4);
  for (; rb < n_blocks; rb += nwg) {
    unsigned char* bcur = lds + cur * BUF;
    unsigned char* bnxt = lds + (cur ^ 1) * BUF;
    const BxRsrc r2 = a_rsrc(rb + 2 * nwg);
    pgt_f32x16 am, ac, am2, ac2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; am2[r] = 0.f; ac2[r] = 0.f; }
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      bx_u32x4 fa[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + i * 32);
      am = bx_mfma(fa[0], bf[i][0], am);
      ac = bx_mfma(fa[0], bf[i][1], ac);
      ac = bx_mfma(fa[1], bf[i][0], ac);
      ac = bx_mfma(fa[1], bf[i][1], ac);
      ac = bx_mfma(fa[0], bf[i][2], ac);
      ac = bx_mfma(fa[2], bf[i][0], ac);
      if (two) {
        bx_u32x4 fb[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fb[q] = b2[(i * 3 + q) * 64 + lane];
        am2 = bx_mfma(fa[0], fb[0], am2);
        ac2 = bx_mfma(fa[0], fb[1], ac2);
        ac2 = bx_mfma(fa[1], fb[0], ac2);
        ac2 = bx_mfma(fa[1], fb[1], ac2);
        ac2 = bx_mfma(fa[0], fb[2], ac2);
        ac2 = bx_mfma(fa[2], fb[0], ac2);
      }
#pragma unroll
      for (int t = i * EPT / KSTEPS; t < (i + 1) * EPT / KSTEPS; ++t) {
        convert_one(t, bnxt);
        issue_load(t, r2);
      }
    }
    if (live) {
      const BxRsrc rc = c_rsrc(cbase, rb);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = am[r] + ac[r] + bias_r;
      if (bx_tile_has_nan(v)) {                              // non-finite operand: exact fp32 tile (rare)
        BX_DRAIN();
        bx_exact_tile(g, a_rsrc(rb), col, hi, bias_r, v);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t soff = (uint32_t)BX_SGPR((int)(((r & 3) + 8 * (r >> 2)) * ldc4));
        BX_STORE1S(v[r], cvoff, rc, soff);
      }
      if (two) {
        const BxRsrc rc2 = c_rsrc(cbase2, rb);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = am2[r] + ac2[r] + bias_r2;
        if (bx_tile_has_nan(v)) {
          BX_DRAIN();
          bx_exact_tile(g, a_rsrc(rb), col2, hi, bias_r2, v);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t soff = (uint32_t)BX_SGPR((int)(((r & 3) + 8 * (r >> 2)) * ldc4));
          BX_STORE1S(v[r], cvoff2, rc2, soff);
        }
      }
    }
    bx_barrier();
    cur ^= 1;
  }
  BX_DRAIN();
}

// ---------------------------------------------------------------------------------------------------------------------
// The short-K product with the wavefronts SPECIALISED (round 5; pgt_tune("gemm_bx_sym_pc", 1)): twelve wavefronts,
//   * ten "consumers" — wavefront w owns the 32-column block w for the whole K (B fragments in registers), reads the A planes of
//     the current block out of LDS, issues its MFMAs and STORES its 32 x 32 tile; it never loads from global memory, so it never
//     waits on vmcnt: the stores of one block drain while the next block's MFMAs run;
//   * two "producers" — they issue no store and no MFMA: 8-byte loads of the next blocks' rows of A (TWO blocks in flight),
//     three bf16 planes into the other LDS buffer.
// In gemm_bx_sym_kernel every wavefront does all of that, and because vmcnt is ONE counter for loads and stores (unordered
// against each other) its wait for a load of the next block is also a wait for the acknowledgement of the previous block's 16 - 32
// stores — every block pays a store round trip.  Here loads and stores are counted by different wavefronts.  Same planes, same
// products in the same order per accumulator (the big term and the corrections apart): the same bits.
// One LDS-only barrier per block; registers: 96 (B) + 32 (accumulators) + fragments — three wavefronts per SIMD.
template <int KSTEPS>
__global__ __launch_bounds__(768, 1) void gemm_bx_sym_pc_kernel(PgtGemmArgs g, int n_blocks) {
  constexpr int BM = 32, KP = KSTEPS * 16, SROW = KP * 2 + 16, PLANE = BM * SROW, BUF = 3 * PLANE;
  constexpr int NCONS = 10, NPROD = 2, NPT = NPROD * 64;
  constexpr int EPT = (BM * KP / 2) / NPT;                 // float pairs per producer thread and block
  static_assert(2 * EPT - 1 < 64, "vmcnt is six bits");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int wave = BX_SGPR(tid >> 6);
  const bool producer = wave >= NCONS;
  const int nwg = gridDim.x;
  const int Ktot = g.n_seg * g.seg_k;
  for (int i = tid; i < 2 * BUF / 16; i += 768) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  int rb = blockIdx.x;
  if (rb >= n_blocks) return;
  __syncthreads();
  const int64_t span_last = (int64_t)(g.n_seg - 1) * g.a_seg_stride * 4;
  auto a_rsrc = [&](int b) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
    const int64_t rows = rows_left < BM ? rows_left : BM;
    return bx_make_rsrc(g.A + (int64_t)(rows_left > 0 ? b : 0) * BM * g.lda,
                        rows_left > 0 ? span_last + (rows - 1) * g.lda * 4 + (int64_t)g.seg_k * 4 : 0);
  };
  // blocks of this workgroup: rb, rb + nwg, ...  S of them (>= 1)
  const int S = (n_blocks - rb + nwg - 1) / nwg;
  if (producer) {
    // ---- element map: 16 threads per row (pairs el + 16 t of a row), 8 rows per pass of the 128 producer threads
    const int ptid = tid - NCONS * 64;
    const int prow = ptid >> 4, el = ptid & 15;
    constexpr int PPR = KP / 32;                           // pairs per thread and row (16 threads per row)
    constexpr int PASSES = BM / (NPT / 16);
    static_assert(PPR * PASSES == EPT, "element map");
    const int half = g.seg_k >> 1, rpairs = g.n_seg * half;
    uint32_t goff[EPT], loff[EPT];
#pragma unroll
    for (int u = 0; u < PASSES; ++u)
#pragma unroll
      for (int t = 0; t < PPR; ++t) {
        const int row = prow + (NPT / 16) * u, pi = el + 16 * t, seg = pi / half, pp = pi - seg * half;
        goff[u * PPR + t] = pi < rpairs ? (uint32_t)((seg * g.a_seg_stride + row * g.lda + 2 * pp) * 4) : 0xfffffff0u;
        loff[u * PPR + t] = (uint32_t)(row * SROW + el * 4 + 64 * t);
      }
    bx_u32x2 raw[2][EPT];
    auto issue_set = [&](int set, const BxRsrc& r) {
#pragma unroll
      for (int t = 0; t < EPT; ++t) BX_LOAD2(raw[set][t], goff[t], r);
    };
    // the set's EPT loads are the oldest in flight, the other set's EPT the younger ones: element t has EPT - 1 - t of its own
    // set and EPT of the other behind it
    auto convert_one = [&](int set, int t, unsigned char* buf, auto NY) {
      BX_WAIT(decltype(NY)::value, raw[set][t]);
      uint32_t p1, p2, p3;
      bx_split2_fast(bx_as_float(raw[set][t][0]), bx_as_float(raw[set][t][1]), p1, p2, p3);
      unsigned char* d = buf + loff[t];
      *reinterpret_cast<uint32_t*>(d) = p1;
      *reinterpret_cast<uint32_t*>(d + PLANE) = p2;
      *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = p3;
    };
    // convert element t of `set`, then request the same element of the block two ahead: the count behind element t stays 2 EPT - 1
    auto half_step = [&](int set, unsigned char* buf, int block_load) {
      const BxRsrc rn = a_rsrc(block_load);
#pragma unroll
      for (int t = 0; t < EPT; ++t) {
        convert_one(set, t, buf, BxInt<2 * EPT - 1>());
        BX_LOAD2(raw[set][t], goff[t], rn);
      }
      bx_barrier();
    };
    issue_set(0, a_rsrc(rb));
    issue_set(1, a_rsrc(rb + nwg));
    {
      const BxRsrc rn = a_rsrc(rb + 2 * nwg);
#pragma unroll
      for (int t = 0; t < EPT; ++t) {
        convert_one(0, t, lds, BxInt<2 * EPT - 1>());
        BX_LOAD2(raw[0][t], goff[t], rn);
      }
    }
    bx_barrier();                                          // block 0 is in buffer 0
    for (int r = 0; r < (S >> 1); ++r) {
      half_step(1, lds + BUF, rb + 3 * nwg);
      half_step(0, lds, rb + 4 * nwg);
      rb += 2 * nwg;
    }
    if (S & 1) half_step(1, lds + BUF, rb + 3 * nwg);
    BX_DRAIN();
    return;
  }
  // ---- consumers
  const int col = wave * 32 + lo;
  const bool live = wave * 32 < g.N;
  bx_u32x4 bf[KSTEPS][3];
#pragma unroll
  for (int i = 0; i < KSTEPS; ++i) {
    const int k0 = i * 16 + 8 * hi;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = k0 + t;
      v[t] = (k < Ktot && col < g.N) ? g.Bw[(int64_t)k * g.sbk + (int64_t)col * g.sbn] : 0.f;
    }
    uint32_t p[3][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bx_split2(v[2 * t], v[2 * t + 1], p[0][t], p[1][t], p[2][t]);
#pragma unroll
    for (int q = 0; q < 3; ++q) { bx_u32x4 f = {p[q][0], p[q][1], p[q][2], p[q][3]}; bf[i][q] = f; }
  }
  const int js = (wave * 32) / g.c_seg_n;
  const float* cbase = g.C + (int64_t)js * g.c_seg_stride + (wave * 32 - js * g.c_seg_n);
  const uint32_t cvoff = col < g.N ? (uint32_t)((4 * hi * g.ldc + lo) * 4) : 0xfffffff0u;
  auto c_rsrc = [&](int b) {
    const int64_t rows_left = (int64_t)g.M - (int64_t)b * BM;
    const int64_t rows = rows_left < BM ? rows_left : BM;
    return bx_make_rsrc(cbase + (int64_t)b * BM * g.ldc, ((rows - 1) * g.ldc + 32) * 4);
  };
  const float bias_r = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  BX_DRAIN();                                              // the B / bias loads (compiler-issued) have landed; from here on only stores
  bx_barrier();                                            // block 0 is in buffer 0
  int cur = 0;
  const int arow = lo * SROW + 16 * hi;
  const uint32_t ldc4 = (uint32_t)(g.ldc * 4);
  for (int it = 0; it < S; ++it, rb += nwg) {
    unsigned char* bcur = lds + cur * BUF;
    pgt_f32x16 am, ac;
#pragma unroll
    for (int r = 0; r < 16; ++r) { am[r] = 0.f; ac[r] = 0.f; }
    if (live) {
      // the fragments of k-step i + 1 are requested BEFORE the six products of k-step i are issued (pinned: left alone the
      // compiler reuses one set of fragment registers and every k-step waits out an LDS round trip in front of its MFMAs)
      bx_u32x4 fa[3], fn[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow);
#pragma unroll
      for (int i = 0; i < KSTEPS; ++i) {
        if (i + 1 < KSTEPS) {
#pragma unroll
          for (int q = 0; q < 3; ++q) fn[q] = *reinterpret_cast<const bx_u32x4*>(bcur + q * PLANE + arow + (i + 1) * 32);
        }
        PGT_SCHED_FENCE();
        am = bx_mfma(fa[0], bf[i][0], am);
        ac = bx_mfma(fa[0], bf[i][1], ac);
        ac = bx_mfma(fa[1], bf[i][0], ac);
        ac = bx_mfma(fa[1], bf[i][1], ac);
        ac = bx_mfma(fa[0], bf[i][2], ac);
        ac = bx_mfma(fa[2], bf[i][0], ac);
        PGT_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = fn[q];
      }
      const BxRsrc rc = c_rsrc(rb);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = am[r] + ac[r] + bias_r;
      if (bx_tile_has_nan(v)) {                              // non-finite operand: exact fp32 tile (rare)
        BX_DRAIN();
        bx_exact_tile(g, a_rsrc(rb), col, hi, bias_r, v);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t soff = (uint32_t)BX_SGPR((int)(((r & 3) + 8 * (r >> 2)) * ldc4));
        BX_STORE1S(v[r], cvoff, rc, soff);
      }
    }
    bx_barrier();
    cur ^= 1;
  }
  BX_DRAIN();
}

int bx_device_cus() {
#ifdef PGT_EMU
  return 4;
#else
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else cus = 256;
  }
  return cus;
#endif
}

}  // namespace

void pgt_gemm_bx_set(int v) { g_bx = v; }
void pgt_gemm_bx_sym_set(int v) { g_bx_sym = v; }
void pgt_gemm_bx_tn_pc_set(int v) { g_bx_tn_pc = v; }
void pgt_gemm_bx_sym_pc_set(int v) { g_bx_sym_pc = v; }

int pgt_gemm_bx_launch(const PgtGemmArgs& g, pgt_stream_t stream) {
  if (!g_bx) return 0;
  const int64_t K = (int64_t)g.n_seg * g.seg_k;
  // covered: K <= 336 in even segments read with 8-byte loads, N <= 128 (256 for K <= 128), no accumulation into C
  if (g.accumulate || K < 16 || K > 336 || g.seg_k % 2 || g.lda % 2 || g.a_seg_stride % 2 || !pgt_aligned(g.A, 8)) return 0;
  if (g.lda < 0 || g.a_seg_stride < 0 || g.c_seg_n <= 0) return 0;
  if (g.M < 8192 && g_bx != 2) return 0;
  // 32-bit byte offsets inside a block's buffer descriptor
  if (((int64_t)g.n_seg * g.a_seg_stride + 33 * g.lda + g.seg_k) * 4 >= (int64_t)0xfff00000) return 0;
  const int wn = g.N <= 128 ? 1 : 2;
  if (g.N > 320 || (wn == 2 && K > 128)) return 0;
  // one column block per wavefront: the row-sliced epilogue moves quads (16 bytes) at absolute 32-bit byte offsets
  if (wn == 1) {
    auto fits = [&](int64_t ld, int64_t extra) { return ld >= 0 && ((int64_t)g.M * ld + extra + 4) * 4 < (int64_t)0x7ffffff0; };
    if (g.N % 4 || g.c_seg_n % 4 || g.c_seg_stride < 0 ||
        !fits(g.ldc, (int64_t)((g.N - 1) / g.c_seg_n) * g.c_seg_stride + g.c_seg_n)) return 0;
  }
  // Where it does not pay (measured inside the training step, M = 211 968; g_bx = 2 runs them anyway for the tests):
  //  * short K into <= 128 columns: little arithmetic per row block, the fp32 tile kernels are as fast (48 vs 50 us);
  //  * K <= 64 into 256 columns: 85 vs 85 us.
  // The candidate-gate product (64 columns, 550 MB moved with its three outputs) lost with half the column wavefronts idle
  // (180 vs 158 us) and wins with K cut four ways and the hardware exp / rcp in its tanh (131 - 144 vs 158 - 164 us).
  if (g_bx != 2 && ((K <= 128 && wn == 1) || (K <= 64 && !g_bx_sym))) return 0;
  if (g.epi) {
    // the gate epilogues: whole 32-column blocks on either side of the z | r boundary, hidden width = N (h) or N / 2 (zr),
    // one output segment, K in the 21-step bucket (the short-K kernels have no epilogue variants)
    if (wn != 1 || K <= 128 || g.c_seg_n != g.N) return 0;
    if (g.epi == 1 && (g.eO % 32 || g.N != 2 * g.eO || (g.N > 64 && g.eO != 64))) return 0;   // reset half = the last 64-column group
    // candidate gate: N = hidden <= 64 (the 128-wide instantiation would spill registers: scripts/bx_isa_audit.py)
    if (g.epi == 2 && (g.N != g.eO || g.N > 64)) return 0;
    if (g.epi != 1 && g.epi != 2) return 0;
    auto fits = [&](int64_t ld, int64_t extra) { return ld >= 0 && ((int64_t)g.M * ld + extra + 4) * 4 < (int64_t)0x7ffffff0; };
    if (!fits(g.eldh, g.eO)) return 0;
    if (g.epi == 1 && !fits(g.eldx, g.efin + g.eO)) return 0;
    if (g.epi == 2) {
      if (!fits(2 * (int64_t)g.eO, 0) || (g.eO1 && !fits(g.eld1, g.N))) return 0;
      const int64_t last = g.e0_period > 0 ? ((int64_t)(g.M - 1) / g.e0_period) * g.e0_hi + (g.e0_period - 1) * g.eld0
                                           : (int64_t)(g.M - 1) * g.eld0;
      if (g.eld0 < 0 || g.e0_hi < 0 || (last + g.N + 4) * 4 >= (int64_t)0x7ffffff0) return 0;
    }
  }
  const int n_blocks = (int)pgt_cdiv(g.M, 32);
  // short K: the symmetric kernel (one column block per wavefront) when the output layout allows its buffer stores
  const bool sym_ok = K <= 128 && g.N > 128 && !g.epi && g.c_seg_n % 32 == 0 && g.ldc >= 0 && g.c_seg_stride >= 0 &&
                      (33 * g.ldc + 32) * 4 < (int64_t)0xfff00000 && g_bx_sym;
  if (g.N > 256 && !sym_ok) return 0;                   // only the symmetric kernel reaches past 256 columns
  int wgs = bx_device_cus();
  if (g_bx == 2 && wgs > 3) wgs = 3;                    // tests: several blocks per workgroup at small sizes
  if (wgs > n_blocks) wgs = n_blocks;
  dim3 grid((unsigned)wgs), block(512);
#define PGT_BX_GO(KS_, WN_, EPI_, Q4_) PGT_LAUNCH((gemm_bx_kernel<KS_, WN_, EPI_, Q4_>), grid, block, stream, g, n_blocks)
  // K <= 64: specialised wavefronts (83 -> 73 - 76 us at the step's 64 -> 320 product).  At K = 128 that form measures the SAME
  // as the all-alike kernel (123 vs 122 us), and balancing the ten column blocks over the SIMDs measured 3 % slower: that launch is
  // matrix-pipe bound at the clock its power draw allows (docs/LAB_NOTEBOOK.md 5.8), so the longer K stays where it was.
  if (sym_ok && g_bx_sym_pc && K <= 64 && g.N <= 320 && g.seg_k % 2 == 0) {
    PGT_LAUNCH((gemm_bx_sym_pc_kernel<4>), grid, dim3(768), stream, g, n_blocks);
  } else if (sym_ok) {
    if (K > 64) PGT_LAUNCH((gemm_bx_sym_kernel<8>), grid, block, stream, g, n_blocks);
    else PGT_LAUNCH((gemm_bx_sym_kernel<4>), grid, block, stream, g, n_blocks);
  } else if (K > 128 && g.N <= 64) {                         // two column blocks: K cut four ways
    if (g.epi == 1) PGT_BX_GO(21, 1, 1, true);
    else if (g.epi == 2) PGT_BX_GO(21, 1, 2, true);
    else PGT_BX_GO(21, 1, 0, true);
  } else if (K > 128) {
    if (g.epi == 1) PGT_BX_GO(21, 1, 1, false);
    else PGT_BX_GO(21, 1, 0, false);
  } else if (K > 64) {
    if (wn == 1) PGT_BX_GO(8, 1, 0, false); else PGT_BX_GO(8, 2, 0, false);
  } else {
    if (wn == 1) PGT_BX_GO(4, 1, 0, false); else PGT_BX_GO(4, 2, 0, false);
  }
#undef PGT_BX_GO
  const int rc = pgt_check_launch("pgt_gemm_f32 (split-bf16)");
  return rc ? rc : 1;
}

// A launch = `wgs` workgroups over a chunk of rows, at most ~5 000 rows (320 stages) per workgroup: what the fp32
// kernels' slabs hold; at most four launches (the deterministic scratch is sized for 1024 slabs)
static void bx_tn_schedule(const PgtTnArgs& t, int* wgs, int64_t* chunk_rows, int* n_chunks) {
  const int64_t n_stages = pgt_cdiv(t.M, 16);
  *wgs = bx_device_cus();
  if (g_bx == 2 && *wgs > 3) *wgs = 3;
  if (*wgs > n_stages) *wgs = (int)n_stages;
  const int64_t per_wg = g_bx == 2 ? 5 : 320;
  int64_t chunks = pgt_cdiv(n_stages, (int64_t)*wgs * per_wg);
  if (chunks > 4) chunks = 4;
  *chunk_rows = pgt_cdiv(pgt_cdiv(n_stages, chunks), *wgs) * *wgs * 16;      // whole rounds of the workgroups
  *n_chunks = (int)pgt_cdiv(t.M, *chunk_rows);
}

int pgt_gemm_bx_tn_plan(const PgtTnArgs& t, int64_t* nslab) {
  if (!g_bx) return 0;
  const int64_t K = (int64_t)t.n_seg * t.seg_k;
  // covered: 128 < K <= 351 (below that the fp32 whole-K kernel is HBM-bound already), N <= 128, tall operands
  if (K <= 128 || K > 351 || t.N > 128 || t.N < 1) return 0;
  if (t.M < 16384 && g_bx != 2) return 0;
  if (t.lda < 0 || t.a_seg_stride < 0 || t.ldg < 0 || !pgt_aligned(t.A, 4) || !pgt_aligned(t.G, 4)) return 0;
  // 32-bit byte offsets inside a stage's buffer descriptors
  if (((int64_t)t.n_seg * t.a_seg_stride + 17 * t.lda + t.seg_k) * 4 >= (int64_t)0xfff00000 ||
      (17 * t.ldg + t.N) * 4 >= (int64_t)0xfff00000) return 0;
  int wgs, n_chunks;
  int64_t chunk_rows;
  bx_tn_schedule(t, &wgs, &chunk_rows, &n_chunks);
  *nslab = (int64_t)wgs * n_chunks;
  return 1;
}

int pgt_gemm_bx_tn_launch(const PgtTnArgs& t, pgt_stream_t stream) {
  int wgs, n_chunks;
  int64_t chunk_rows;
  bx_tn_schedule(t, &wgs, &chunk_rows, &n_chunks);
  for (int c = 0; c < n_chunks; ++c) {
    PgtTnArgs u = t;
    const int64_t row0 = (int64_t)c * chunk_rows;
    u.A = t.A + row0 * t.lda;
    u.G = t.G + row0 * t.ldg;
    u.M = (int)((t.M - row0) < chunk_rows ? (t.M - row0) : chunk_rows);
    const int n_stages = (int)pgt_cdiv(u.M, 16);
    dim3 grid((unsigned)wgs), block(512);                  // every workgroup writes its slab, also when it has no stage
    const int slab_base = c * wgs;
    // the specialised kernel reads column PAIRS with 8-byte loads: even widths and pitches, 8-byte aligned operands
    const bool pc_ok = g_bx_tn_pc && t.seg_k % 2 == 0 && t.lda % 2 == 0 && t.a_seg_stride % 2 == 0 && t.N % 2 == 0 && t.ldg % 2 == 0 &&
                       pgt_aligned(u.A, 8) && pgt_aligned(u.G, 8);
#define PGT_BX_TN_GO(NCB_, DET_, CW_, PW_)                                                                             \
    do {                                                                                                                \
      if (pc_ok && (g_bx_tn_pc == 2 || NCB_ == 2))                                                                      \
        PGT_LAUNCH((gemm_bx_tn_pc_kernel<NCB_, DET_, CW_, PW_>), grid, dim3(64 * (CW_ + PW_)), stream, u, n_stages, slab_base); \
      else if (pc_ok) PGT_LAUNCH((gemm_bx_tn_pc_kernel<NCB_, DET_, 4, 4>), grid, block, stream, u, n_stages, slab_base); \
      else PGT_LAUNCH((gemm_bx_tn_kernel<NCB_, DET_>), grid, block, stream, u, n_stages, slab_base);                    \
    } while (0)
    // N <= 64: twelve wavefronts, four consumers + eight producers (885 us per call at the step's shape against 933 for four +
    // four and 980 all alike); N = 128: four + four (1 246 us; eight consumers + four producers measure the same, 1 230 - 1 240:
    // pgt_tune("gemm_bx_tn_pc", 2) launches that form) — profiles/r05v_tn_twelve_wavefronts_ab.jsonl
    if (t.part != nullptr) {
      // (the atomics-free epilogue on twelve wavefronts spills 80 registers — 603 against 454 us per launch at the step's shape: four + four)
      if (t.N > 64) PGT_BX_TN_GO(4, true, 8, 4); else PGT_BX_TN_GO(2, true, 4, 4);
    } else {
      if (t.N > 64) PGT_BX_TN_GO(4, false, 8, 4); else PGT_BX_TN_GO(2, false, 4, 8);
    }
#undef PGT_BX_TN_GO
  }
  return pgt_check_launch("pgt_gemm_tn_acc_f32 (split-bf16)");
}

