// Whole DCRNN sequences at hidden width 64, one workgroup per sample: BatchedDCRNN.forward (nn/recurrent/dcrnn.py:429-475; cell
// :194-219, gates :172-192, diffusion convolution :85-106) for graphs of up to ~220 nodes (METR-LA: 207) — the benchmarked
// model BatchedDCRNN(2, 64, K = 3).
//
// The general path runs four launches per cell step (two LDS-resident diffusion stacks, two gate-fused split-bf16 products): at
// the reference's default batch (64 windows, pems_ddp.py:31) every one of them is a latency-bound chain of a few 32-row blocks per
// CU, and at B = 1024 every diffusion term crosses HBM twice (the stack kernel writes it, the product reads it back).  Here a
// 1024-thread workgroup owns a sample for the WHOLE T-step sequence:
//   * the sample's [N, 64 + Fin] block lives in LDS (two blocks, 272-byte rows: sixteen hidden quads + one quad of input columns)
//     next to both CSR operators (16-bit sources, fp32 coefficients); the hops are gathered out of LDS exactly as in
//     csrc/dconv_slab.hip (one ds_read_b128 per slot, fmaf chain in slot order: the diffusion terms are bit-identical to that path);
//   * every diffusion term is multiplied by its weight block WHILE it sits in LDS: split-bf16 products (three bf16 pieces per fp32
//     operand, the six largest piece products, fp32 accumulation: csrc/gemm_bx.hip) on v_mfma_f32_16x16x32_bf16 — wavefront w owns
//     rows 16 w .. 16 w + 15 and ALL output columns, so its A fragment (its own rows, converted in registers) is built once per 32
//     columns of a term; the weights arrive pre-split in fragment order (pgt_dcrnn_seq64_pack_f32, once per optimizer step) and
//     stream through a two-slot LDS ring (12 KB chunks) filled by the last two wavefronts, one LDS-only barrier per chunk;
//   * the two input columns of a term (+ the bias) are a rank-2 update on the VALU; the gate chains run on the accumulators:
//     Z stays in registers from the update / reset epilogue to the blend, H_{t-1} in the accumulator layout across steps, H * R and
//     the new state go straight back into the LDS block as the next T_0;
//   * what the adjoint needs (both stacks, Z | R, the candidate, the states) is stored once, in the layout of the general path
//     (ops.DCRNNSeqFunction), so the general backward and the weight-gradient product run unchanged on it.
// Two blocks are enough for K = 3 because T_1^i = P_i T_0 is gathered together with T_1^o and waits in registers:
//   A = T0 | B = T1o, regs = T1i | MFMA T0, T1o | A = T2o = 2 P_o B - T0 | MFMA T2o | B = T1i | A = T2i = 2 P_i B - T0 | MFMA T1i, T2i.
#include "pgt_common.h"

namespace {

// ---- platform layer (tests/hipemu/pgt_sq_platform_emu.h spells the same operations in plain C++ for the CPU test double)
#ifdef PGT_EMU
#include "pgt_sq_platform_emu.h"
constexpr int SQ_CUS = 4;
#else
constexpr int SQ_CUS = 256;
typedef __bf16 sq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sq_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));
typedef float sq_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t sq_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float sq_as_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t sq_as_uint(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint32_t sq_pack(float x, float y) {     // v_cvt_pk_bf16_f32: round to nearest even
  sq_f32x2 v = {x, y};
  sq_bf16x2 r = __builtin_convertvector(v, sq_bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t sq_perm_hi16(uint32_t y, uint32_t x) { return __builtin_amdgcn_perm(y, x, 0x07060302u); }
__device__ __forceinline__ float sq_rcp(float x) { return __frcp_rn(x); }
__device__ __forceinline__ float sq_exp(float x) { return __expf(x); }
__device__ __forceinline__ void sq_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ sq_f32x4 sq_mfma16(sq_u32x4 a, sq_u32x4 b, sq_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sq_bf16x8, a), __builtin_bit_cast(sq_bf16x8, b), c, 0, 0, 0);
}
// the loaders' weight stream: loads the compiler does not track (its counter insertion waits for vmcnt(0) before every ring write and
// at every loop back-edge, which made each chunk cost a full L2 round trip) and hand-counted waits tied to the registers they
// release.  Loads return in order, so "at most n younger LOADS outstanding" is safe whatever stores are in flight beside them.
__device__ __forceinline__ uint64_t sq_uniform64(uint64_t x) {      // a wave-uniform value, provably so: both halves through v_readfirstlane
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
// (uniform 64-bit base in scalar registers + 32-bit lane offset + immediate: no 64-bit address per load for the compiler to hoist and spill)
#define SQ_GLOAD4(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) : "memory")
#define SQ_VMWAIT4(n, r0, r1, r2, r3) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(n))
#define SQ_VMWAIT6(n, r) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(n))
// one discarded dword: brings its cache line towards this CU ahead of the load proper
// (`sink` must stay allocated until the data has landed — the compiler cannot see the pending write to it: SQ_TOUCH_DONE(sink) stands
// where the loads proper have been waited for, which are younger and return in order)
#define SQ_TOUCH(sink, ptr) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(ptr) : "memory")
#define SQ_TOUCH_DONE(sink) asm volatile("" :: "v"(sink))
// exact fp32 products (v_mfma_f32_16x16x4_f32): the rank-Fin update of the input columns
__device__ __forceinline__ sq_f32x4 sq_mfma4(float a, float b, sq_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif

// (x, y) -> three packed bf16 pairs (low half = x's piece), every piece rounded to nearest: the packed weights
__device__ __forceinline__ void sq_split2(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = sq_pack(x, y);
  float rx = x - sq_as_float(p1 << 16), ry = y - sq_as_float(p1 & 0xffff0000u);
  rx = (fabsf(rx) <= 3.0e38f) ? rx : 0.f;      // inf / nan: the first piece carries it, the others are zero
  ry = (fabsf(ry) <= 3.0e38f) ? ry : 0.f;
  p2 = sq_pack(rx, ry);
  rx -= sq_as_float(p2 << 16);
  ry -= sq_as_float(p2 & 0xffff0000u);
  p3 = sq_pack(rx, ry);
}
// the streaming operand: first piece rounded to nearest, the other two cut off (csrc/gemm_bx.hip bx_split2_fast)
__device__ __forceinline__ void sq_split2_fast(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = sq_pack(x, y);
  float rx = x - sq_as_float(p1 << 16), ry = y - sq_as_float(p1 & 0xffff0000u);
  p2 = sq_perm_hi16(sq_as_uint(ry), sq_as_uint(rx));
  rx -= sq_as_float(p2 << 16);
  ry -= sq_as_float(p2 & 0xffff0000u);
  p3 = sq_perm_hi16(sq_as_uint(ry), sq_as_uint(rx));
}
__device__ __forceinline__ float sq_sigmoidf(float x) { return sq_rcp(1.f + sq_exp(-x)); }
__device__ __forceinline__ float sq_tanhf(float x) {             // csrc/gemm_bx.hip bx_tanhf
  const float x2 = x * x;
  const float small = x * fmaf(x2, fmaf(x2, 0.13333334f, -0.33333334f), 1.f);
  const float big = 1.f - 2.f * sq_rcp(1.f + sq_exp(2.f * x));
  return fabsf(x) < 0.04f ? small : big;
}
__device__ __forceinline__ pgt_f4 sq_fma4(float w, pgt_f4 x, pgt_f4 acc) {
  return pgt_mk4(fmaf(w, x.x, acc.x), fmaf(w, x.y, acc.y), fmaf(w, x.z, acc.z), fmaf(w, x.w, acc.w));
}
// 2 g - t with the rounding of csrc/dconv_slab.hip's axpby4 (2.0f * g + (-1.0f) * t)
__device__ __forceinline__ pgt_f4 sq_two_minus(pgt_f4 g, pgt_f4 t) {
  return pgt_mk4(2.0f * g.x + -1.0f * t.x, 2.0f * g.y + -1.0f * t.y, 2.0f * g.z + -1.0f * t.z, 2.0f * g.w + -1.0f * t.w);
}

// lab/seq64_lab.hip defines this to take the kernels apart (1: no MFMAs, 2: the B fragments of a chunk read once, 4: no ring
// writes, 8: no gathers, 16: no stores of the saved stacks / of dP, 32: no T_0 read back in the second hops / no parked slice in the
// adjoint, 64: no T_1^i read back, 128: no operand loads of the gate adjoints); compile-time constants in the library
#ifndef SQ_LAB_SKIP
#define SQ_LAB_SKIP(bit) false
#endif

constexpr int SQ_THREADS = 1024;
constexpr int SQ_O = 64;             // hidden width
constexpr int SQ_PITCH = 68;         // floats per LDS row: 16 hidden quads + the quad of input columns
constexpr int SQ_MAXT = 4;           // gather tasks (row, quad) per thread: 17 N <= 4 * 896
constexpr int SQ_CHUNK_DW = 3072;    // one ring slot: 4 column tiles x 3 planes x 64 lanes x 4 dwords = 12 KB
constexpr int SQ_LOADER0 = 14;       // the last two wavefronts fill the ring (128 lanes x 6 x 16 bytes = one chunk) and do nothing else
constexpr int SQ_GTHREADS = 896;     // threads that take gather tasks: every wavefront but the loaders
constexpr int SQ_LDS = 160 * 1024;

struct Seq64Args {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;
  int N, Fin, K, T, B, nnz_o, nnz_i;
  const float* X; int64_t xs_b, xs_t;     // X[b, t] = X + b xs_b + t xs_t: [N, Fin] rows
  const float* H0;                         // [B, N, 64] | null (zeros)
  const uint32_t* Wp;                      // the packed weights: chunks in consumption order (pgt_dcrnn_seq64_pack_f32)
  const float* Wzr; const float* Wh;       // the stacked fp32 operands [(2K-1)(Fin+64), 128 | 64]: the input-column rows are read in place
  const float* bzr; const float* bh;       // [128] | null, [64] | null
  float* out; int64_t os_b, os_t;          // out[b, t] = out + b os_b + t os_t: [N, 64] rows
  float* TSzr; float* TSh;                 // saved stacks: segment s, step t, row m = b N + n at s seg_stride + t t_stride + m C
  int64_t seg_stride, t_stride;
  float* ZR; float* HT;                    // [T, B N, 128], [T, B N, 64]
};

__host__ __device__ inline int sq_nseg(int K) { return 2 * K - 1; }
__host__ __device__ inline int sq_nchunks(int K) { return sq_nseg(K) * 6; }   // per cell step: 4 per segment (z | r) + 2 (candidate)
// position in the consumption order -> stack segment ([T0 | T1o T1i | T2o T2i]): K = 3: T0 T1o T2o T1i T2i; K = 2: T0 T1o T1i
__host__ __device__ inline int sq_seg_at(int K, int pos) {
  if (K >= 3) { const int order[5] = {0, 1, 3, 2, 4}; return order[pos]; }
  return pos;
}

// slots of an operator in LDS, padding included (rows start at even positions; an even count keeps the next array 4-byte aligned)
__host__ __device__ inline size_t sq_slot_cap(int64_t nnz, int64_t N) { return (size_t)((nnz + N + 2) & ~(int64_t)1); }
// two blocks, the ring, fp32 coefficients + 16-bit sources per slot, one 32-bit row descriptor per operator and task row, the
// 16-bit row of a task row
static size_t sq_lds_bytes(int64_t N, int64_t nnz_o, int64_t nnz_i) {
  return 2 * (size_t)N * SQ_PITCH * 4 + 2 * (size_t)SQ_CHUNK_DW * 4 + (sq_slot_cap(nnz_o, N) + sq_slot_cap(nnz_i, N)) * 6 +
         2 * (size_t)N * 4 + (size_t)(N + 2) * 2;
}

struct SqLds {
  float* bufA; float* bufB;
  uint32_t* ring;
  const float* val_o; const float* val_i;
  const uint16_t* col_o; const uint16_t* col_i;
  const uint32_t* desc_o; const uint32_t* desc_i;    // task row p -> first slot (even) | slot count << 16 of row prow[p]
  const uint16_t* prow;                              // task row p -> row: by falling slot count (both operators together)
};

// Slot lists in LDS: 16-bit sources and fp32 coefficients in two arrays, row r's slots from the EVEN position
// sq_row_start(rp[r], r) on (room for one padding slot per row), so that two slots are one 4-byte + one 8-byte read — the same two
// LDS instructions per slot pair + two quad reads as the packed (col, val) slots of csrc/dconv_slab.hip, at 6 instead of 8 bytes per slot.
__host__ __device__ inline int sq_row_start(int rp_r, int r) { return (rp_r + r + 1) & ~1; }
// The gather tasks walk the rows in the order of FALLING slot count (sq_setup: `prow`, one order for both operators and both
// directions — the counts of a row in P_o and P_i go together on road graphs: 0.93 correlation on the benchmark's), so that the four
// rows a wavefront gathers at a time are equally long give or take a slot: in row order the longest of four has 10.3 slots against
// a mean of 7.3 on the METR-LA-shaped graph, and the wavefront walks the longest (the adjoint's gather phases 3.5 -> 3.1 us, its
// launch - 3.4 %; the forward's, which also carry the stores of the saved stacks, measure the same either way).
// Row sum over the slots of one row, `desc` = its first slot | slot count << 16 (csrc/dconv_slab.hip gather_q: the same fmaf chain
// in slot order), four slots in flight
__device__ __forceinline__ pgt_f4 sq_gather(uint32_t desc, const uint16_t* __restrict__ col, const float* __restrict__ val,
                                            const float* __restrict__ blk, int qoff) {
  pgt_f4 acc = pgt_mk4(0.f, 0.f, 0.f, 0.f);
  int q = (int)(desc & 0xffffu);
  const int e = q + (int)(desc >> 16);
  if (SQ_LAB_SKIP(8)) return *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)col[q] * SQ_PITCH);
  for (; q + 4 <= e; q += 4) {
    const uint32_t c01 = *reinterpret_cast<const uint32_t*>(col + q), c23 = *reinterpret_cast<const uint32_t*>(col + q + 2);
    const float2 v01 = *reinterpret_cast<const float2*>(val + q), v23 = *reinterpret_cast<const float2*>(val + q + 2);
    const pgt_f4 x0 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c01 & 0xffffu) * SQ_PITCH);
    const pgt_f4 x1 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c01 >> 16) * SQ_PITCH);
    const pgt_f4 x2 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c23 & 0xffffu) * SQ_PITCH);
    const pgt_f4 x3 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c23 >> 16) * SQ_PITCH);
    acc = sq_fma4(v01.x, x0, acc);
    acc = sq_fma4(v01.y, x1, acc);
    acc = sq_fma4(v23.x, x2, acc);
    acc = sq_fma4(v23.y, x3, acc);
  }
  if (q + 2 <= e) {
    const uint32_t c01 = *reinterpret_cast<const uint32_t*>(col + q);
    const float2 v01 = *reinterpret_cast<const float2*>(val + q);
    const pgt_f4 x0 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c01 & 0xffffu) * SQ_PITCH);
    const pgt_f4 x1 = *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)(c01 >> 16) * SQ_PITCH);
    acc = sq_fma4(v01.x, x0, acc);
    acc = sq_fma4(v01.y, x1, acc);
    q += 2;
  }
  if (q < e) acc = sq_fma4(val[q], *reinterpret_cast<const pgt_f4*>(blk + qoff + (int)col[q] * SQ_PITCH), acc);
  return acc;
}

// LDS carve-up (two blocks, the ring, both operators) and the staging of the operators: every thread of the workgroup calls it
__device__ __forceinline__ SqLds sq_setup(char* smem, int N, const int32_t* rp_o, const int32_t* col_o, const float* val_o, int nnz_o,
                                          const int32_t* rp_i, const int32_t* col_i, const float* val_i, int nnz_i) {
  const int tid = threadIdx.x;
  SqLds s;
  char* p = smem;
  s.bufA = reinterpret_cast<float*>(p); p += (size_t)N * SQ_PITCH * 4;
  s.bufB = reinterpret_cast<float*>(p); p += (size_t)N * SQ_PITCH * 4;
  s.ring = reinterpret_cast<uint32_t*>(p); p += 2 * (size_t)SQ_CHUNK_DW * 4;
  const size_t cap_o = sq_slot_cap(nnz_o, N), cap_i = sq_slot_cap(nnz_i, N);
  float* vo = reinterpret_cast<float*>(p); p += cap_o * 4;
  float* vi = reinterpret_cast<float*>(p); p += cap_i * 4;
  uint16_t* co = reinterpret_cast<uint16_t*>(p); p += cap_o * 2;
  uint16_t* ci = reinterpret_cast<uint16_t*>(p); p += cap_i * 2;
  uint32_t* dso = reinterpret_cast<uint32_t*>(p); p += (size_t)N * 4;
  uint32_t* dsi = reinterpret_cast<uint32_t*>(p); p += (size_t)N * 4;
  uint16_t* pr = reinterpret_cast<uint16_t*>(p);
  // task order: rows by falling slot count (P_o's + P_i's), ties in row order — every row counts the rows ahead of it (block A is
  // free at this point: the keys wait there)
  uint32_t* key = reinterpret_cast<uint32_t*>(s.bufA);
  for (int r = tid; r < N; r += SQ_THREADS) key[r] = (uint32_t)((rp_o[r + 1] - rp_o[r]) + (rp_i[r + 1] - rp_i[r]));
  __syncthreads();
  for (int r = tid; r < N; r += SQ_THREADS) {
    const uint32_t k = key[r];
    int ahead = 0;
    for (int o = 0; o < N; ++o) ahead += (key[o] > k || (key[o] == k && o < r)) ? 1 : 0;
    pr[ahead] = (uint16_t)r;
    dso[ahead] = (uint32_t)sq_row_start(rp_o[r], r) | ((uint32_t)(rp_o[r + 1] - rp_o[r]) << 16);
    dsi[ahead] = (uint32_t)sq_row_start(rp_i[r], r) | ((uint32_t)(rp_i[r + 1] - rp_i[r]) << 16);
  }
  for (int r = tid; r < 2 * N; r += SQ_THREADS) {           // one thread per (operator, row): its slots to the row's even start
    const bool second = r >= N;
    const int row = second ? r - N : r;
    const int32_t* rp = second ? rp_i : rp_o;
    const int32_t* gcol = second ? col_i : col_o;
    const float* gval = second ? val_i : val_o;
    uint16_t* dc = second ? ci : co;
    float* dv = second ? vi : vo;
    const int b0 = rp[row], e0 = rp[row + 1], st = sq_row_start(b0, row);
    for (int q = b0; q < e0; ++q) { dc[st + q - b0] = (uint16_t)gcol[q]; dv[st + q - b0] = gval[q]; }
  }
  s.val_o = vo; s.val_i = vi; s.col_o = co; s.col_i = ci; s.desc_o = dso; s.desc_i = dsi; s.prow = pr;
  return s;
}

// ---- packed weights.  Chunk c of a cell step (consumption order): c < 4 S: the update / reset product, segment sq_seg_at(c / 4),
// hidden columns 32 kk .. + 31 (kk = (c / 2) % 2), output columns 64 half .. + 63 (half = c % 2); then 2 S chunks of the candidate
// product (segment, kk).  Inside a chunk: [column tile ct (4)][plane (3)][lane (64)][4 dwords] — lane l of v_mfma_f32_16x16x32_bf16
// holds B[k = 8 (l / 16) + j][n = l % 16], j = 0 .. 7, two bf16 per dword.
__global__ __launch_bounds__(256) void seq64_pack_kernel(const float* __restrict__ Wzr, const float* __restrict__ Wh, int Fin, int K,
                                                         uint32_t* __restrict__ Wp) {
  const int S = sq_nseg(K), C = Fin + SQ_O;
  const int total = sq_nchunks(K) * 4 * 64 * 4;
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (idx >= total) return;
  const int d = idx & 3, l = (idx >> 2) & 63, ct = (idx >> 8) & 3, c = idx >> 10;
  const float* W;
  int ld, seg, kk, n;
  if (c < 4 * S) {
    W = Wzr; ld = 2 * SQ_O; seg = sq_seg_at(K, c >> 2); kk = (c >> 1) & 1; n = 64 * (c & 1) + 16 * ct + (l & 15);
  } else {
    const int ch = c - 4 * S;
    W = Wh; ld = SQ_O; seg = sq_seg_at(K, ch >> 1); kk = ch & 1; n = 16 * ct + (l & 15);
  }
  const int h0 = 32 * kk + 8 * (l >> 4) + 2 * d;
  const float x = W[(int64_t)(seg * C + Fin + h0) * ld + n], y = W[(int64_t)(seg * C + Fin + h0 + 1) * ld + n];
  uint32_t p1, p2, p3;
  sq_split2(x, y, p1, p2, p3);
  uint32_t* dst = Wp + (int64_t)c * SQ_CHUNK_DW + ((ct * 3) * 64 + l) * 4 + d;
  dst[0] = p1;
  dst[256] = p2;
  dst[512] = p3;
}

// one 16-byte store / load of a hidden quad at 8-byte alignment (a row of 66 floats starts every 264 bytes; csrc/dconv_slab.hip
// stQ / ldQ), a float2 for the quad of input columns
__device__ __forceinline__ void sq_store_quad(float* p, bool hidden, pgt_f4 v) {
  if (SQ_LAB_SKIP(16)) return;
  if (hidden) __builtin_memcpy(__builtin_assume_aligned(p, 8), &v, 16);
  else *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
}
__device__ __forceinline__ pgt_f4 sq_load_quad(const float* p, bool hidden) {
  pgt_f4 v;
  if (hidden) {
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 8), 16);
  } else {
    const float2 a = *reinterpret_cast<const float2*>(p);
    v = pgt_mk4(a.x, a.y, 0.f, 0.f);
  }
  return v;
}
__device__ __forceinline__ int sq_min(int a, int b) { return a < b ? a : b; }

// gather task j of a thread (the loader wavefronts take none): idx = tid + 896 j; idx < 16 N: (task row idx / 16, hidden quad idx % 16);
// then the N quads of input columns.  Task row p stands for row prow[p] (sq_setup: rows by falling slot count)
struct SqTask {
  int pos, row, quad;
  bool live;
  __device__ __forceinline__ bool hidden() const { return quad < 16; }
  __device__ __forceinline__ int loff() const { return row * SQ_PITCH + 4 * quad; }                 // inside an LDS block
  __device__ __forceinline__ int goff(int C, int Fin) const { return row * C + (quad < 16 ? Fin + 4 * quad : 0); }   // inside [N, C]
};
__device__ __forceinline__ SqTask sq_task(int tid, int j, int N, const uint16_t* __restrict__ prow) {
  const int idx = tid + j * SQ_GTHREADS;
  SqTask k;
  k.live = idx < 17 * N;
  const int ic = k.live ? idx : 0;
  k.pos = ic < 16 * N ? (ic >> 4) : ic - 16 * N;
  k.quad = ic < 16 * N ? (ic & 15) : 16;
  k.row = (int)prow[k.pos];
  return k;
}

// lab/seq64_lab.hip defines this to record the phase timeline of workgroup 0 (wall_clock64 ticks); a no-op in the library
#ifndef SQ_MARK
#define SQ_MARK(t, G, slot) do { } while (0)
#endif

// a value the compiler must treat as unknown: address arithmetic that depends on it is recomputed where it is used instead of being
// hoisted out of the time loop and spilled (the first build of this kernel carried 444 spilled registers of loop-invariant addresses)
#ifdef PGT_EMU
#define SQ_OPAQUE(x) ((void)0)
#else
#define SQ_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
template <int V> struct SqInt { static constexpr int value = V; };

// ---- the loaders' side of the ring.  The last TWO wavefronts stream the packed weights and do nothing else: they issue no store
// and no other load, so their vmcnt counts exactly their own chunk loads, which return in order (wavefronts that take part in the
// gathers carry the stores of the saved stacks on the same counter: every wait for a chunk then also waited for write
// acknowledgements, 1 - 2.5 us per product phase).  Chunk c of the cell step (0 .. NCH - 1, NCH even) sits in ring slot c & 1 and
// travels through register set c & 1 (6 x 16 bytes per lane: 24 registers a set — with twelve pieces per lane in ONE wavefront
// the compiler spilled a set right behind its loads, i.e. before the data had landed).  Turn C (the MFMA wavefronts work on
// chunk C): chunk C + 1 has landed once at most the six loads of chunk C + 2 are outstanding; it goes to slot (C + 1) & 1 and
// its set is requested again for chunk C + 3 — two chunks of look-ahead, kept up across the gather phases.
struct SqLoader {
  sq_u32x4 pre0[6], pre1[6];
  const uint32_t* Wp;
  uint32_t* ring;
  int ll;                           // loader lane 0 .. 127
  uint32_t voff[3];                 // byte offsets of this lane's pieces 0 - 1, 2 - 3, 4 - 5 of a chunk (2 KB apart within a pair)
  __device__ __forceinline__ void init(const uint32_t* w, uint32_t* r, int l) {
    Wp = w; ring = r; ll = l;
    voff[0] = 16u * (uint32_t)l; voff[1] = voff[0] + 4096u; voff[2] = voff[0] + 8192u;
  }
  template <int CHUNK>
  __device__ __forceinline__ void issue() {
    const uint64_t base = sq_uniform64(reinterpret_cast<uint64_t>(Wp + CHUNK * SQ_CHUNK_DW));
    sq_u32x4(&st)[6] = (CHUNK & 1) ? pre1 : pre0;
    SQ_GLOAD4(st[0], voff[0], base, 0); SQ_GLOAD4(st[1], voff[0], base, 2048);
    SQ_GLOAD4(st[2], voff[1], base, 0); SQ_GLOAD4(st[3], voff[1], base, 2048);
    SQ_GLOAD4(st[4], voff[2], base, 0); SQ_GLOAD4(st[5], voff[2], base, 2048);
  }
  template <int CHUNK, int YOUNGER>
  __device__ __forceinline__ void write() {
    sq_u32x4(&st)[6] = (CHUNK & 1) ? pre1 : pre0;
    SQ_VMWAIT6(YOUNGER, st);
    sq_u32x4* dst = reinterpret_cast<sq_u32x4*>(ring + (CHUNK & 1) * SQ_CHUNK_DW) + ll;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (!SQ_LAB_SKIP(4) || i == 0) dst[128 * i] = st[i];
  }
  __device__ __forceinline__ void start() {       // kernel start: chunk 0 into slot 0; chunks 1 and 2 requested
    issue<0>();
    write<0, 0>();
    issue<1>();
    issue<2>();
  }
  template <int C, int NCH>
  __device__ __forceinline__ void turn() {
    write<(C + 1) % NCH, 6>();
    issue<(C + 3) % NCH>();
  }
};

// The whole kernel for one kind of wavefront.  LOADER = the three wavefronts that stream the packed weights into the ring: they run
// the same loop nest and meet the same barriers as everybody else (and take their share of the gathers), but their product phases
// move chunks instead of issuing MFMAs — as a separate instantiation, so that the chunk registers are not live in the MFMA
// wavefronts' code and the accumulators not in theirs (one body for both carried 530 spilled registers).
template <bool LOADER, int K>
__device__ __forceinline__ void sq_fwd_body(const Seq64Args& a, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N, Fin = a.Fin, C = Fin + SQ_O;
  constexpr int S = 2 * K - 1, NCH = 6 * S;
  // ---- LDS carve-up: two blocks, the ring, both operators
  const SqLds s = sq_setup(smem, N, a.rp_o, a.col_o, a.val_o, a.nnz_o, a.rp_i, a.col_i, a.val_i, a.nnz_i);
  // ---- roles
  const int NRT = (N + 15) >> 4;                 // row tiles = MFMA wavefronts
  const bool consumer = !LOADER && wave < NRT;
  constexpr bool loader = LOADER;
  // MFMA lane map: A row (clamped: the sums of rows past N are never used), D rows 4 (lane / 16) + i, D column lane % 16
  const int aoff = sq_min(16 * wave + (lane & 15), N - 1) * SQ_PITCH + 8 * (lane >> 4);
  const int dcol = lane & 15;
  const int drow0 = 16 * wave + 4 * (lane >> 4);
  SqLoader ld;
  if constexpr (LOADER) {
    ld.init(a.Wp, s.ring, tid - SQ_LOADER0 * 64);
    ld.start();
  }
  float hprev[4][4];                 // H_{t-1} in the accumulator layout: [column tile][row i]
  sq_f32x4 acc[8];

  for (int b = (int)blockIdx.x; b < a.B; b += (int)gridDim.x) {
    const int64_t m0 = (int64_t)b * N;
    // ---- H_0 and X_0 into block A
    sq_barrier();                    // (every lane is done with the blocks of the previous sample; first pass: operators staged)
    if constexpr (!LOADER)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = drow0 + i;
        float h = 0.f;
        if (consumer && r < N) {
          if (a.H0) h = a.H0[(m0 + r) * SQ_O + 16 * ct + dcol];
          s.bufA[r * SQ_PITCH + 16 * ct + dcol] = h;
        }
        hprev[ct][i] = h;
      }
    float2 xt = make_float2(0.f, 0.f), xn = make_float2(0.f, 0.f);
    if (!LOADER && tid < N) {
      const float* xp = a.X + (int64_t)b * a.xs_b + (int64_t)tid * Fin;
      xt.x = xp[0];
      if (Fin > 1) xt.y = xp[1];
      *reinterpret_cast<pgt_f4*>(s.bufA + tid * SQ_PITCH + 64) = pgt_mk4(xt.x, xt.y, 0.f, 0.f);
    }
#pragma unroll 1
    for (int t = 0; t < a.T; ++t) {
      if (!LOADER && tid < N && t + 1 < a.T) {   // next step's input columns: requested a whole step ahead
        const float* xp = a.X + (int64_t)b * a.xs_b + (int64_t)(t + 1) * a.xs_t + (int64_t)tid * Fin;
        xn.x = xp[0];
        if (Fin > 1) xn.y = xp[1];
      }
      // rows of this (sample, step) in the saved tensors: wave-uniform bases, 32-bit lane offsets
      float* const zr_rows = a.ZR + ((int64_t)t * a.B * N + m0) * (2 * SQ_O);
      float* const ht_rows = a.HT + ((int64_t)t * a.B * N + m0) * SQ_O;
      float* const out_rows = a.out + (int64_t)b * a.os_b + (int64_t)t * a.os_t;

      // ================================================================ one diffusion convolution + its gate chain
      auto gate = [&](auto gtag) {
        constexpr int G = decltype(gtag)::value;          // 0: update | reset gates (128 columns), 1: candidate (64 columns)
        constexpr int NOUT = G == 0 ? 2 * SQ_O : SQ_O, NCT = NOUT / 16;
        const float* const Wg = G == 0 ? a.Wzr : a.Wh;
        const float* const bg = G == 0 ? a.bzr : a.bh;
        float* const ts0 = (G == 0 ? a.TSzr : a.TSh) + (int64_t)t * a.t_stride + m0 * C;
        sq_barrier();                // block A = T_0 of this convolution, complete
        SQ_MARK(t, G, 0);
        if (consumer) {              // the sums start from the bias (requested here, needed after the first hop)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const float bias = bg ? bg[16 * ct + dcol] : 0.f;
            acc[ct] = sq_f32x4{bias, bias, bias, bias};
          }
        }

        // ---- products of one stack segment sitting in `buf` with its weight block (position POS of the consumption order): NSEG
        // chunks, (32 hidden columns kk, 64 output columns half) each; chunk index of the step C0 + j
        auto mfma_seg = [&](const float* buf, auto postag) {
          constexpr int POS = decltype(postag)::value;
          constexpr int NSEG = NCT / 2;
          constexpr int C0 = (G == 0 ? 0 : 4 * S) + POS * NSEG;
          // the segment's input columns: a rank-Fin update as ONE exact-fp32 MFMA per column tile — lane l supplies B[k = l / 16]
          // [n = l % 16] = the weight row of input column k (zero for k >= Fin), requested here and used behind the first chunk
          float xw[NCT];
          int ol = lane;
          SQ_OPAQUE(ol);             // (the lane's part of these addresses is recomputed here, not kept per segment across the time loop)
          if (consumer) {
            const float* const wx = Wg + (int64_t)(sq_seg_at(K, POS) * C) * NOUT;
            const int xo = sq_min(ol >> 4, Fin - 1) * NOUT + (ol & 15);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) xw[ct] = (ol >> 4) < Fin ? wx[xo + 16 * ct] : 0.f;
          }
          sq_u32x4 a1, a2, a3;
          sq_u32x4 bq[2][3];
          auto chunk = [&](auto jtag) {
            constexpr int J = decltype(jtag)::value;
            constexpr int kk = NSEG == 4 ? J / 2 : J, half = NSEG == 4 ? J % 2 : 0, CH = C0 + J;
            if (consumer) {
              if (half == 0) {
                const pgt_f4 f0 = *reinterpret_cast<const pgt_f4*>(buf + aoff + 32 * kk);
                const pgt_f4 f1 = *reinterpret_cast<const pgt_f4*>(buf + aoff + 32 * kk + 4);
                uint32_t p1, p2, p3;
                sq_split2_fast(f0.x, f0.y, p1, p2, p3); a1[0] = p1; a2[0] = p2; a3[0] = p3;
                sq_split2_fast(f0.z, f0.w, p1, p2, p3); a1[1] = p1; a2[1] = p2; a3[1] = p3;
                sq_split2_fast(f1.x, f1.y, p1, p2, p3); a1[2] = p1; a2[2] = p2; a3[2] = p3;
                sq_split2_fast(f1.z, f1.w, p1, p2, p3); a1[3] = p1; a2[3] = p2; a3[3] = p3;
                if (kk == 1) {
                  // (before the segment's last barrier, so that whoever rewrites the block afterwards cannot race with this read)
                  const float xa = (lane >> 4) < Fin ? buf[(aoff - 8 * (lane >> 4)) + 64 + (lane >> 4)] : 0.f;   // A[row][k = lane / 16]
#pragma unroll
                  for (int ct = 0; ct < NCT; ++ct) acc[ct] = sq_mfma4(xa, xw[ct], acc[ct]);
                }
              }
              // B fragments of column tile ct + 1 are read while the six products of tile ct issue (two sets, not all four).  The
              // chunk's barrier stands BEFORE the products of its last tile — every read of this slot is behind it, the loaders have
              // written the next chunk into the other slot — and the first fragments of the next chunk are requested right behind
              // it: their LDS latency and the barrier's skew hide under six MFMAs instead of idling the matrix pipe at every chunk
              const sq_u32x4* slot = reinterpret_cast<const sq_u32x4*>(s.ring + (CH & 1) * SQ_CHUNK_DW) + lane;
              if (J == 0) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[0][pl] = slot[pl * 64];
              }
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                if (ct < 3 && !SQ_LAB_SKIP(2)) {
#pragma unroll
                  for (int pl = 0; pl < 3; ++pl) bq[(ct + 1) & 1][pl] = slot[((ct + 1) * 3 + pl) * 64];
                }
                if (ct == 3) {
                  sq_barrier();
                  if (J < NSEG - 1) {
                    const sq_u32x4* nslot = reinterpret_cast<const sq_u32x4*>(s.ring + ((CH + 1) & 1) * SQ_CHUNK_DW) + lane;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[0][pl] = nslot[pl * 64];
                  }
                }
                const sq_u32x4 b1 = bq[SQ_LAB_SKIP(2) ? 0 : ct & 1][0], b2 = bq[SQ_LAB_SKIP(2) ? 0 : ct & 1][1], b3 = bq[SQ_LAB_SKIP(2) ? 0 : ct & 1][2];
                sq_f32x4 c = acc[4 * half + ct];
                if (!SQ_LAB_SKIP(1)) {
                  // (one chain of six on the running sum: two chains of three, added at the end, measured 12 % SLOWER per phase)
                  c = sq_mfma16(a3, b1, c);
                  c = sq_mfma16(a1, b3, c);
                  c = sq_mfma16(a2, b2, c);
                  c = sq_mfma16(a2, b1, c);
                  c = sq_mfma16(a1, b2, c);
                  c = sq_mfma16(a1, b1, c);
                } else {
                  c[0] += sq_as_float(a1[0] ^ b1[0] ^ a2[1] ^ b2[1] ^ a3[2] ^ b3[2]);
                }
                acc[4 * half + ct] = c;
                PGT_SCHED_FENCE();
              }
            } else {
              if constexpr (LOADER) ld.template turn<CH, NCH>();
              sq_barrier();
            }
          };
          chunk(SqInt<0>{});
          chunk(SqInt<1>{});
          if constexpr (NSEG == 4) {
            chunk(SqInt<2>{});
            chunk(SqInt<3>{});
          }
        };

        // ---- hop 1: T1o = P_o T0 -> block B; T1i = P_i T0 and T0 itself wait in the saved stack (this thread's own stores,
        // read back by the same thread) while block A is still being read by the products
        int ot = tid;
        SQ_OPAQUE(ot);
if constexpr (!LOADER)
#pragma unroll
        for (int j = 0; j < SQ_MAXT; ++j) {
          const SqTask k = sq_task(ot, j, N, s.prow);
          if (k.live) {
            const int go = k.goff(C, Fin);
            sq_store_quad(ts0 + go, k.hidden(), *reinterpret_cast<const pgt_f4*>(s.bufA + k.loff()));
            const pgt_f4 o1 = sq_gather(s.desc_o[k.pos], s.col_o, s.val_o, s.bufA, 4 * k.quad);
            const pgt_f4 i1 = sq_gather(s.desc_i[k.pos], s.col_i, s.val_i, s.bufA, 4 * k.quad);
            *reinterpret_cast<pgt_f4*>(s.bufB + k.loff()) = o1;
            sq_store_quad(ts0 + a.seg_stride + go, k.hidden(), o1);
            sq_store_quad(ts0 + 2 * a.seg_stride + go, k.hidden(), i1);
          }
        }
        sq_barrier();
        SQ_MARK(t, G, 1);
        mfma_seg(s.bufA, SqInt<0>{});
        mfma_seg(s.bufB, SqInt<1>{});
        SQ_MARK(t, G, 2);
        if constexpr (K >= 3) {
          // ---- hop 2, first direction: T2o = 2 P_o T1o - T0 -> block A (T_0 is dead in LDS)
          ot = tid;
          SQ_OPAQUE(ot);
if constexpr (!LOADER)
#pragma unroll
          for (int j = 0; j < SQ_MAXT; ++j) {
            const SqTask k = sq_task(ot, j, N, s.prow);
            if (k.live) {
              const int go = k.goff(C, Fin);
              const pgt_f4 t0 = SQ_LAB_SKIP(32) ? pgt_mk4(0.f, 0.f, 0.f, 0.f) : sq_load_quad(ts0 + go, k.hidden());
              const pgt_f4 o2 = sq_two_minus(sq_gather(s.desc_o[k.pos], s.col_o, s.val_o, s.bufB, 4 * k.quad), t0);
              *reinterpret_cast<pgt_f4*>(s.bufA + k.loff()) = o2;
              sq_store_quad(ts0 + 3 * a.seg_stride + go, k.hidden(), o2);
            }
          }
          // T1i comes back from the saved stack: requested here, a product phase ahead of its use
          pgt_f4 i1[SQ_MAXT];
if constexpr (!LOADER)
#pragma unroll
          for (int j = 0; j < SQ_MAXT; ++j) {
            const SqTask k = sq_task(ot, j, N, s.prow);
            i1[j] = pgt_mk4(0.f, 0.f, 0.f, 0.f);
            if (k.live && !SQ_LAB_SKIP(64)) i1[j] = sq_load_quad(ts0 + 2 * a.seg_stride + k.goff(C, Fin), k.hidden());
          }
          sq_barrier();
          SQ_MARK(t, G, 3);
          mfma_seg(s.bufA, SqInt<2>{});
          SQ_MARK(t, G, 4);
          ot = tid;
          SQ_OPAQUE(ot);
if constexpr (!LOADER)
#pragma unroll
          for (int j = 0; j < SQ_MAXT; ++j) {
            const SqTask k = sq_task(ot, j, N, s.prow);
            if (k.live) *reinterpret_cast<pgt_f4*>(s.bufB + k.loff()) = i1[j];
          }
          sq_barrier();
          SQ_MARK(t, G, 5);
          // ---- hop 2, second direction: T2i = 2 P_i T1i - T0 -> block A
          ot = tid;
          SQ_OPAQUE(ot);
if constexpr (!LOADER)
#pragma unroll
          for (int j = 0; j < SQ_MAXT; ++j) {
            const SqTask k = sq_task(ot, j, N, s.prow);
            if (k.live) {
              const int go = k.goff(C, Fin);
              const pgt_f4 t0 = SQ_LAB_SKIP(32) ? pgt_mk4(0.f, 0.f, 0.f, 0.f) : sq_load_quad(ts0 + go, k.hidden());
              const pgt_f4 i2 = sq_two_minus(sq_gather(s.desc_i[k.pos], s.col_i, s.val_i, s.bufB, 4 * k.quad), t0);
              *reinterpret_cast<pgt_f4*>(s.bufA + k.loff()) = i2;
              sq_store_quad(ts0 + 4 * a.seg_stride + go, k.hidden(), i2);
            }
          }
          sq_barrier();
          SQ_MARK(t, G, 6);
          mfma_seg(s.bufB, SqInt<3>{});
          mfma_seg(s.bufA, SqInt<4>{});
        } else {
          ot = tid;
          SQ_OPAQUE(ot);
if constexpr (!LOADER)
#pragma unroll
          for (int j = 0; j < SQ_MAXT; ++j) {
            const SqTask k = sq_task(ot, j, N, s.prow);
            if (k.live) *reinterpret_cast<pgt_f4*>(s.bufA + k.loff()) = sq_load_quad(ts0 + 2 * a.seg_stride + k.goff(C, Fin), k.hidden());
          }
          sq_barrier();
          mfma_seg(s.bufA, SqInt<2>{});
        }
        SQ_MARK(t, G, 7);
        // ---- gate chain on the accumulators (every product of this convolution is behind a barrier: both blocks are free)
        int orow = drow0;
        SQ_OPAQUE(orow);
        if (G == 0) {
          if (consumer) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int r = orow + i;
                const float g = sq_sigmoidf(acc[ct][i]);
                if (r < N) {
                  if (ct >= 4) s.bufA[r * SQ_PITCH + 16 * (ct & 3) + dcol] = hprev[ct & 3][i] * g;     // H * R: the candidate's T_0
                  zr_rows[r * (2 * SQ_O) + 16 * ct + dcol] = g;
                }
              }
            }
          }
          if (!LOADER && tid < N) *reinterpret_cast<pgt_f4*>(s.bufA + tid * SQ_PITCH + 64) = pgt_mk4(xt.x, xt.y, 0.f, 0.f);
        } else {
          if (consumer) {
            float z[4][4];           // Z of this step: stored by this very lane in the update / reset epilogue, read back instead of held
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
              for (int i = 0; i < 4; ++i) z[ct][i] = zr_rows[sq_min(orow + i, N - 1) * (2 * SQ_O) + 16 * ct + dcol];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int r = orow + i;
                const float ht = sq_tanhf(acc[ct][i]);
                const float hn = pgt_gru_blend(z[ct][i], hprev[ct][i], ht);
                hprev[ct][i] = hn;
                if (r < N) {
                  s.bufA[r * SQ_PITCH + 16 * ct + dcol] = hn;                                 // the next step's T_0
                  out_rows[r * SQ_O + 16 * ct + dcol] = hn;
                  ht_rows[r * SQ_O + 16 * ct + dcol] = ht;
                }
              }
            }
          }
          xt = xn;
          if (!LOADER && tid < N) *reinterpret_cast<pgt_f4*>(s.bufA + tid * SQ_PITCH + 64) = pgt_mk4(xt.x, xt.y, 0.f, 0.f);
        }
        SQ_MARK(t, G, 8);
      };
      gate(SqInt<0>{});
      gate(SqInt<1>{});
    }
  }
}

// (lab/seq64_lab.hip instantiates the loader wavefront's path as a kernel of its own: its register report must show NO scratch — a
// register spilled while a hand-issued load is pending on it would be saved before the data lands)
template <int K>
__global__ __launch_bounds__(SQ_THREADS) void dcrnn_seq64_fwd_kernel(Seq64Args a) {
  __shared__ __attribute__((aligned(16))) char smem[SQ_LDS];
  if ((int)(threadIdx.x >> 6) >= SQ_LOADER0) sq_fwd_body<true, K>(a, smem);
  else sq_fwd_body<false, K>(a, smem);
}


// ================================================================================================================ adjoint
// Hand-written BPTT of the same sequences (ops.DCRNNSeqFunction.backward: gate adjoints pgt_gru_h_bwd_f32 / pgt_gru_zr_bwd_f32,
// feature-gradient products dP W^T, stack adjoints pgt_dconv_stack_slab_bwd_f32 — six launches per cell step) as ONE launch for
// all T steps of every sample, the input being data (no d/dX): only the 64 hidden columns of a stack gradient exist.
//   * wavefront w owns rows 16 w .. 16 w + 15 in the MFMA A-operand layout (lane = row l % 16, columns 8 (l / 16) .. + 7 of every
//     32): the gate adjoints are computed in that layout, so dP IS the A operand of the feature-gradient products and the running
//     d/dH_t never leaves the registers; dPh / dP(z | r) are stored once (the weight-gradient product over all T steps reads them);
//   * G_s = dP W_s^T per stack segment (split-bf16, weights transposed + pre-split by pgt_dcrnn_seq64_pack_bwd_f32, the "- T_0" of
//     the second hop folded into segment 0: ops.fold_backward_weight) leaves the accumulators straight into an LDS block;
//   * the stack adjoint runs on the two blocks with the TRANSPOSED operators:
//       A = G2o, B = G1o | B += 2 P_o^T A | park = P_o^T B | A = G2i, B = G1i | B += 2 P_i^T A | A = G0' | A += P_i^T B + park
//     (`park`: one [N, 64] slice of global scratch per workgroup, written and read back by the same threads);
//   * d/dT_0 comes back out of block A in the A-operand layout for the next gate adjoint.
struct Seq64BwdArgs {
  const int32_t* rp_o; const int32_t* col_o; const float* val_o;    // P_o^T
  const int32_t* rp_i; const int32_t* col_i; const float* val_i;    // P_i^T
  int N, Fin, K, T, B, nnz_o, nnz_i;
  const float* dOut; int64_t gs_b, gs_t;   // incoming gradient of out[b, t]: [N, 64] rows
  const float* out; int64_t os_b, os_t;    // the states
  const float* H0;                         // [B, N, 64] | null (zeros)
  const float* ZR; const float* HT;        // saved by the forward launch
  const uint32_t* Wp;                      // pgt_dcrnn_seq64_pack_bwd_f32
  float* dPzr; float* dPh;                 // [T, B N, 128], [T, B N, 64]
  float* dH0;                              // [B, N, 64] | null
  float* park;                             // [gridDim.x][N][64]
};

// position in the adjoint's consumption order -> stack segment: K = 3: G2o G1o G2i G1i G0; K = 2: G1o G1i G0
__host__ __device__ inline int sq_bwd_seg_at(int K, int pos) {
  if (K >= 3) { const int order[5] = {3, 1, 4, 2, 0}; return order[pos]; }
  const int order2[3] = {1, 2, 0};
  return order2[pos];
}

// Packed weights of the adjoint.  Chunk c of a cell step: c < 2 S: the candidate's product (segment sq_bwd_seg_at(c / 2), columns
// 32 kk .. + 31 of dPh, kk = c % 2); then 4 S chunks of the z | r product (segment, kk = c % 4 over the 128 columns of dP(z | r)).
// Inside a chunk [column tile ct][plane][lane][4 dwords]: B[k = 8 (l / 16) + j][n = 16 ct + l % 16] = W'[(seg C + Fin + n), 32 kk + k]
// with W' = the stacked operand, segment 0 folded for K = 3 (W_0 - W_3 - W_4: ops.fold_backward_weight).
__global__ __launch_bounds__(256) void seq64_pack_bwd_kernel(const float* __restrict__ Wzr, const float* __restrict__ Wh, int Fin, int K,
                                                             uint32_t* __restrict__ Wp) {
  const int S = sq_nseg(K), C = Fin + SQ_O;
  const int total = sq_nchunks(K) * 4 * 64 * 4;
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (idx >= total) return;
  const int d = idx & 3, l = (idx >> 2) & 63, ct = (idx >> 8) & 3, c = idx >> 10;
  const float* W;
  int ld, seg, kk;
  if (c < 2 * S) {
    W = Wh; ld = SQ_O; seg = sq_bwd_seg_at(K, c >> 1); kk = c & 1;
  } else {
    const int ch = c - 2 * S;
    W = Wzr; ld = 2 * SQ_O; seg = sq_bwd_seg_at(K, ch >> 2); kk = ch & 3;
  }
  const int n = 16 * ct + (l & 15), k0 = 32 * kk + 8 * (l >> 4) + 2 * d;
  const float* row = W + (int64_t)(seg * C + Fin + n) * ld;
  float x = row[k0], y = row[k0 + 1];
  if (seg == 0 && K == 3) {
    const float* r3 = W + (int64_t)(3 * C + Fin + n) * ld;
    const float* r4 = W + (int64_t)(4 * C + Fin + n) * ld;
    x = (x - r3[k0]) - r4[k0];                       // the order of ops.fold_backward_weight (k = 2: d = 0, then d = 1)
    y = (y - r3[k0 + 1]) - r4[k0 + 1];
  }
  uint32_t p1, p2, p3;
  sq_split2(x, y, p1, p2, p3);
  uint32_t* dst = Wp + (int64_t)c * SQ_CHUNK_DW + ((ct * 3) * 64 + l) * 4 + d;
  dst[0] = p1;
  dst[256] = p2;
  dst[512] = p3;
}

template <bool LOADER, int K>
__device__ __forceinline__ void sq_bwd_body(const Seq64BwdArgs& a, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N;
  constexpr int S = 2 * K - 1, NCH = 6 * S;
  const SqLds s = sq_setup(smem, N, a.rp_o, a.col_o, a.val_o, a.nnz_o, a.rp_i, a.col_i, a.val_i, a.nnz_i);
  const int NRT = (N + 15) >> 4;
  const bool consumer = !LOADER && wave < NRT;
  // A-operand layout: this lane's row (clamped; `rvalid`: it exists) and its first column of every 32
  const int arow_raw = 16 * wave + (lane & 15);
  const bool rvalid = consumer && arow_raw < N;
  const int arow = sq_min(arow_raw, N - 1);
  const int acol = 8 * (lane >> 4);
  const int dcol = lane & 15;
  const int drow0 = 16 * wave + 4 * (lane >> 4);
  SqLoader ld;
  if constexpr (LOADER) {
    ld.init(a.Wp, s.ring, tid - SQ_LOADER0 * 64);
    ld.start();
  }
  float* const park = a.park + (int64_t)blockIdx.x * N * SQ_O;

  // G = dP W_seg^T for one stack segment (position POS of the adjoint's consumption order): KK chunks (32 columns of dP each), the
  // 64 columns of G leave the accumulators into `dst` (accumulator layout: rows 4 (l / 16) + i, column 16 ct + l % 16) before the
  // segment's last barrier.  Chunk index of the step: the candidate's 2 S chunks come first, then the 4 S of z | r
  auto mfma_to = [&](float* dst, const float* dp, auto kktag, auto postag) {
    constexpr int KK = decltype(kktag)::value, POS = decltype(postag)::value;
    constexpr int C0 = (KK == 2 ? 0 : 2 * S) + POS * KK;
    sq_f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = sq_f32x4{0.f, 0.f, 0.f, 0.f};
    sq_u32x4 bq[2][3];
    auto chunk = [&](auto jtag) {
      constexpr int kk = decltype(jtag)::value, CH = C0 + kk;
      constexpr bool LAST = kk == KK - 1;       // the segment's last chunk keeps its barrier at the end: G must be in `dst` behind it
      if (consumer) {
        sq_u32x4 a1, a2, a3;
        uint32_t p1, p2, p3;
        sq_split2_fast(dp[8 * kk + 0], dp[8 * kk + 1], p1, p2, p3); a1[0] = p1; a2[0] = p2; a3[0] = p3;
        sq_split2_fast(dp[8 * kk + 2], dp[8 * kk + 3], p1, p2, p3); a1[1] = p1; a2[1] = p2; a3[1] = p3;
        sq_split2_fast(dp[8 * kk + 4], dp[8 * kk + 5], p1, p2, p3); a1[2] = p1; a2[2] = p2; a3[2] = p3;
        sq_split2_fast(dp[8 * kk + 6], dp[8 * kk + 7], p1, p2, p3); a1[3] = p1; a2[3] = p2; a3[3] = p3;
        const sq_u32x4* slot = reinterpret_cast<const sq_u32x4*>(s.ring + (CH & 1) * SQ_CHUNK_DW) + lane;
        if (kk == 0) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bq[0][pl] = slot[pl * 64];
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          if (ct < 3) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[(ct + 1) & 1][pl] = slot[((ct + 1) * 3 + pl) * 64];
          }
          if (ct == 3 && !LAST) {              // (see the forward kernel: the barrier before the last tile's products, the next chunk's
            sq_barrier();                      // first fragments requested right behind it)
            const sq_u32x4* nslot = reinterpret_cast<const sq_u32x4*>(s.ring + ((CH + 1) & 1) * SQ_CHUNK_DW) + lane;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[0][pl] = nslot[pl * 64];
          }
          const sq_u32x4 b1 = bq[ct & 1][0], b2 = bq[ct & 1][1], b3 = bq[ct & 1][2];
          sq_f32x4 c = acc[ct];
          if (!SQ_LAB_SKIP(1)) {
            c = sq_mfma16(a3, b1, c);
            c = sq_mfma16(a1, b3, c);
            c = sq_mfma16(a2, b2, c);
            c = sq_mfma16(a2, b1, c);
            c = sq_mfma16(a1, b2, c);
            c = sq_mfma16(a1, b1, c);
          } else {
            c[0] += sq_as_float(a1[0] ^ b1[0] ^ a2[1] ^ b2[1] ^ a3[2] ^ b3[2]);
          }
          acc[ct] = c;
          PGT_SCHED_FENCE();
        }
        if (LAST) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (drow0 + i < N) dst[(drow0 + i) * SQ_PITCH + 16 * ct + dcol] = acc[ct][i];
          sq_barrier();
        }
      } else {
        if constexpr (LOADER) ld.template turn<CH, NCH>();
        sq_barrier();
      }
    };
    chunk(SqInt<0>{});
    chunk(SqInt<1>{});
    if constexpr (KK == 4) {
      chunk(SqInt<2>{});
      chunk(SqInt<3>{});
    }
  };

  // adjoint of one diffusion convolution: dp = this lane's KK * 8 columns of dP; on exit block A holds d/dT_0 (hidden columns),
  // NOT yet behind a barrier
  int mark_t = 0;                     // (lab timeline only)
  // `after_products`: the caller's cache-line touches for the next gate adjoint, run two gather phases before the convolution ends
  auto conv_adjoint = [&](const float* dp, auto kktag, auto&& after_products) {
    constexpr int MG = decltype(kktag)::value == 2 ? 0 : 1;
    // hidden quad j of this thread: tasks tid + 896 j < 16 N, task row `pos` = row prow[pos] (sq_setup), `off` inside an LDS block
    auto own = [&](int j, int& pos, int& off, bool& live) {
      int ot = tid;
      SQ_OPAQUE(ot);
      const int idx = ot + j * SQ_GTHREADS;
      live = idx < 16 * N;
      pos = live ? idx >> 4 : 0;
      off = (int)s.prow[pos] * SQ_PITCH + 4 * (idx & 15);
    };
    auto addq = [](pgt_f4 x, pgt_f4 y) { return pgt_mk4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); };
    if constexpr (K >= 3) {
      mfma_to(s.bufA, dp, kktag, SqInt<0>{});          // G2o
      mfma_to(s.bufB, dp, kktag, SqInt<1>{});          // G1o
      SQ_MARK(mark_t, MG, 1);
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {  // B += 2 P_o^T A
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const pgt_f4 g = sq_gather(s.desc_o[pos], s.col_o, s.val_o, s.bufA, off % SQ_PITCH);
          const pgt_f4 t = *reinterpret_cast<const pgt_f4*>(s.bufB + off);
          *reinterpret_cast<pgt_f4*>(s.bufB + off) = pgt_mk4(2.0f * g.x + 1.0f * t.x, 2.0f * g.y + 1.0f * t.y, 2.0f * g.z + 1.0f * t.z, 2.0f * g.w + 1.0f * t.w);
        }
      }
      sq_barrier();
      SQ_MARK(mark_t, MG, 2);
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {  // park = P_o^T B
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const int r = off / SQ_PITCH, q4 = off % SQ_PITCH;
          const pgt_f4 pk = sq_gather(s.desc_o[pos], s.col_o, s.val_o, s.bufB, q4);
          if (!SQ_LAB_SKIP(32)) *reinterpret_cast<pgt_f4*>(park + r * SQ_O + q4) = pk;
        }
      }
      sq_barrier();
      SQ_MARK(mark_t, MG, 3);
      mfma_to(s.bufA, dp, kktag, SqInt<2>{});          // G2i
      mfma_to(s.bufB, dp, kktag, SqInt<3>{});          // G1i
      SQ_MARK(mark_t, MG, 4);
      after_products();                    // (ahead of a gather phase without global loads: nothing queues behind what it requests)
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {  // B += 2 P_i^T A
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const pgt_f4 g = sq_gather(s.desc_i[pos], s.col_i, s.val_i, s.bufA, off % SQ_PITCH);
          const pgt_f4 t = *reinterpret_cast<const pgt_f4*>(s.bufB + off);
          *reinterpret_cast<pgt_f4*>(s.bufB + off) = pgt_mk4(2.0f * g.x + 1.0f * t.x, 2.0f * g.y + 1.0f * t.y, 2.0f * g.z + 1.0f * t.z, 2.0f * g.w + 1.0f * t.w);
        }
      }
      sq_barrier();
      SQ_MARK(mark_t, MG, 5);
      mfma_to(s.bufA, dp, kktag, SqInt<4>{});          // G0 (the "- T_0" of the second hop folded in)
      SQ_MARK(mark_t, MG, 6);
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {  // A += park + P_i^T B
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const int r = off / SQ_PITCH, q4 = off % SQ_PITCH;
          const pgt_f4 pk = SQ_LAB_SKIP(32) ? pgt_mk4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const pgt_f4*>(park + r * SQ_O + q4);
          const pgt_f4 g = sq_gather(s.desc_i[pos], s.col_i, s.val_i, s.bufB, q4);
          const pgt_f4 t = *reinterpret_cast<const pgt_f4*>(s.bufA + off);
          *reinterpret_cast<pgt_f4*>(s.bufA + off) = addq(addq(t, pk), g);
        }
      }
    } else {
      mfma_to(s.bufA, dp, kktag, SqInt<0>{});          // G1o
      mfma_to(s.bufB, dp, kktag, SqInt<1>{});          // G1i
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {  // park = P_o^T A + P_i^T B
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const int r = off / SQ_PITCH, q4 = off % SQ_PITCH;
          const pgt_f4 go = sq_gather(s.desc_o[pos], s.col_o, s.val_o, s.bufA, q4);
          const pgt_f4 gi = sq_gather(s.desc_i[pos], s.col_i, s.val_i, s.bufB, q4);
          *reinterpret_cast<pgt_f4*>(park + r * SQ_O + q4) = addq(go, gi);
        }
      }
      sq_barrier();
      mfma_to(s.bufA, dp, kktag, SqInt<2>{});          // G0
      after_products();
if constexpr (!LOADER)
#pragma unroll
      for (int j = 0; j < SQ_MAXT; ++j) {
        int pos, off; bool live;
        own(j, pos, off, live);
        if (live) {
          const int r = off / SQ_PITCH, q4 = off % SQ_PITCH;
          const pgt_f4 t = *reinterpret_cast<const pgt_f4*>(s.bufA + off);
          *reinterpret_cast<pgt_f4*>(s.bufA + off) = addq(t, *reinterpret_cast<const pgt_f4*>(park + r * SQ_O + q4));
        }
      }
    }
  };

  // 8 floats of this lane's row at columns c0 + acol .. + 7 of a [rows, ld] global array
  auto ld8 = [&](const float* rows, int ld, int c0, float* v) {
    if (SQ_LAB_SKIP(128)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.25f;
      return;
    }
    const pgt_f4 x = *reinterpret_cast<const pgt_f4*>(rows + arow * ld + c0 + acol);
    const pgt_f4 y = *reinterpret_cast<const pgt_f4*>(rows + arow * ld + c0 + acol + 4);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  };
  auto st8 = [&](float* rows, int ld, int c0, const float* v) {
    if (!rvalid || SQ_LAB_SKIP(16)) return;
    *reinterpret_cast<pgt_f4*>(rows + arow * ld + c0 + acol) = pgt_mk4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<pgt_f4*>(rows + arow * ld + c0 + acol + 4) = pgt_mk4(v[4], v[5], v[6], v[7]);
  };

  for (int b = (int)blockIdx.x; b < a.B; b += (int)gridDim.x) {
    const int64_t m0 = (int64_t)b * N;
    float dh[16];                      // running d/dH_t, A-operand layout: [kk][8]
#pragma unroll
    for (int i = 0; i < 16; ++i) dh[i] = 0.f;
    // The gate adjoints read four [N, 64] operands per step from HBM in this lane's own layout; requested right before use, their
    // latency is 17 of a step's 75 us.  (Requested a gather phase ahead into 48 - 64 registers they were spilled on the spot:
    // 75 -> 95 us per step.)  Instead their cache lines are TOUCHED a gather phase ahead — one discarded dword per 32-byte piece —
    // so that the loads proper find them in L2.
    uint32_t sink = 0;
    auto rows_of = [&](int t, const float*& zr_rows, const float*& ht_rows, const float*& g_rows, const float*& hp_rows) {
      zr_rows = a.ZR + ((int64_t)t * a.B * N + m0) * (2 * SQ_O);
      ht_rows = a.HT + ((int64_t)t * a.B * N + m0) * SQ_O;
      g_rows = a.dOut + (int64_t)b * a.gs_b + (int64_t)t * a.gs_t;
      hp_rows = t > 0 ? a.out + (int64_t)b * a.os_b + (int64_t)(t - 1) * a.os_t : (a.H0 ? a.H0 + m0 * SQ_O : nullptr);
    };
    auto touch_blend_operands = [&](int t) {
      const float *zr_rows, *ht_rows, *g_rows, *hp_rows;
      rows_of(t, zr_rows, ht_rows, g_rows, hp_rows);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        SQ_TOUCH(sink, g_rows + arow * SQ_O + 32 * kk + acol);
        SQ_TOUCH(sink, zr_rows + arow * (2 * SQ_O) + 32 * kk + acol);
        SQ_TOUCH(sink, ht_rows + arow * SQ_O + 32 * kk + acol);
        if (hp_rows) SQ_TOUCH(sink, hp_rows + arow * SQ_O + 32 * kk + acol);
      }
    };
    if (consumer) touch_blend_operands(a.T - 1);
    sq_barrier();
#pragma unroll 1
    for (int t = a.T - 1; t >= 0; --t) {
      const float *zr_rows, *ht_rows, *g_rows, *hp_rows;
      rows_of(t, zr_rows, ht_rows, g_rows, hp_rows);
      float* const dpzr_rows = a.dPzr + ((int64_t)t * a.B * N + m0) * (2 * SQ_O);
      float* const dph_rows = a.dPh + ((int64_t)t * a.B * N + m0) * SQ_O;
      // ---- adjoint of the blend and of the candidate's tanh (pgt_gru_h_bwd_f32): dPh, dP(z), d/dH_{t-1} (the Z H part)
      float dp[32];
      if (consumer) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float g[8], z[8], h[8], c[8], dpz[8];
          ld8(g_rows, SQ_O, 32 * kk, g);
          ld8(zr_rows, 2 * SQ_O, 32 * kk, z);
          ld8(ht_rows, SQ_O, 32 * kk, c);
          if (hp_rows) ld8(hp_rows, SQ_O, 32 * kk, h);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (!hp_rows) h[i] = 0.f;
            const float gi = g[i] + dh[8 * kk + i];
            dp[8 * kk + i] = gi * (1.f - z[i]) * (1.f - c[i] * c[i]);
            dpz[i] = gi * (h[i] - c[i]) * z[i] * (1.f - z[i]);
            dh[8 * kk + i] = gi * z[i];
          }
          st8(dph_rows, SQ_O, 32 * kk, dp + 8 * kk);
          st8(dpzr_rows, 2 * SQ_O, 32 * kk, dpz);
        }
        SQ_TOUCH_DONE(sink);
      }
      mark_t = t;
      SQ_MARK(t, 0, 0);
      conv_adjoint(dp, SqInt<2>{}, [&]() {
        if (consumer) {                // the reset-gate adjoint's R (H_{t-1} and this lane's own dP(z) were just here)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) SQ_TOUCH(sink, zr_rows + arow * (2 * SQ_O) + SQ_O + 32 * kk + acol);
        }
      });
      sq_barrier();
      SQ_MARK(t, 0, 7);
      // ---- adjoint of H * R and of the reset gate (pgt_gru_zr_bwd_f32); dP(z) comes back from this lane's own store
      if (consumer) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float r[8], h[8];
          ld8(zr_rows, 2 * SQ_O, SQ_O + 32 * kk, r);
          if (hp_rows) ld8(hp_rows, SQ_O, 32 * kk, h);
          ld8(dpzr_rows, 2 * SQ_O, 32 * kk, dp + 8 * kk);
          const pgt_f4 x = *reinterpret_cast<const pgt_f4*>(s.bufA + arow * SQ_PITCH + 32 * kk + acol);
          const pgt_f4 y = *reinterpret_cast<const pgt_f4*>(s.bufA + arow * SQ_PITCH + 32 * kk + acol + 4);
          const float d0[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (!hp_rows) h[i] = 0.f;
            dp[16 + 8 * kk + i] = d0[i] * h[i] * r[i] * (1.f - r[i]);
            dh[8 * kk + i] += d0[i] * r[i];
          }
          st8(dpzr_rows, 2 * SQ_O, SQ_O + 32 * kk, dp + 16 + 8 * kk);
        }
        SQ_TOUCH_DONE(sink);
      }
      sq_barrier();                    // block A has been read: the next products may overwrite it
      SQ_MARK(t, 1, 0);
      conv_adjoint(dp, SqInt<4>{}, [&]() {
        if (consumer && t > 0) touch_blend_operands(t - 1);
      });
      sq_barrier();
      SQ_MARK(t, 1, 7);
      if (consumer) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const pgt_f4 x = *reinterpret_cast<const pgt_f4*>(s.bufA + arow * SQ_PITCH + 32 * kk + acol);
          const pgt_f4 y = *reinterpret_cast<const pgt_f4*>(s.bufA + arow * SQ_PITCH + 32 * kk + acol + 4);
          dh[8 * kk + 0] += x.x; dh[8 * kk + 1] += x.y; dh[8 * kk + 2] += x.z; dh[8 * kk + 3] += x.w;
          dh[8 * kk + 4] += y.x; dh[8 * kk + 5] += y.y; dh[8 * kk + 6] += y.z; dh[8 * kk + 7] += y.w;
        }
      }
      sq_barrier();                    // (block A read before the next step's products write it)
      SQ_MARK(t, 1, 8);
    }
    if (a.dH0 && consumer) {
      float* rows = a.dH0 + m0 * SQ_O;
      st8(rows, SQ_O, 0, dh);
      st8(rows, SQ_O, 32, dh + 8);
    }
  }
}

template <int K>
__global__ __launch_bounds__(SQ_THREADS) void dcrnn_seq64_bwd_kernel(Seq64BwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[SQ_LDS];
  if ((int)(threadIdx.x >> 6) >= SQ_LOADER0) sq_bwd_body<true, K>(a, smem);
  else sq_bwd_body<false, K>(a, smem);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- C ABI
extern "C" int pgt_dcrnn_seq64_fits(int64_t N, int64_t E_o, int64_t E_i, int64_t Fin, int64_t O, int64_t K) {
  if (O != SQ_O || Fin != 2 || (K != 2 && K != 3) || N < 1 || E_o < 0 || E_i < 0) return 0;
  if ((N + 15) / 16 > SQ_LOADER0 || 17 * N > SQ_MAXT * SQ_GTHREADS) return 0;
  if (E_o > 65535 || E_i > 65535) return 0;
  return sq_lds_bytes(N, E_o, E_i) <= (size_t)SQ_LDS ? 1 : 0;
}

extern "C" int64_t pgt_dcrnn_seq64_pack_floats(int64_t K) { return (int64_t)sq_nchunks((int)K) * SQ_CHUNK_DW; }

extern "C" int pgt_dcrnn_seq64_pack_f32(const float* Wzr, const float* Wh, int64_t Fin, int64_t K, float* Wp, pgt_stream_t stream) {
  PGT_REQUIRE(Wzr && Wh && Wp, "pgt_dcrnn_seq64_pack_f32: null pointer");
  PGT_REQUIRE(Fin == 2 && (K == 2 || K == 3), "pgt_dcrnn_seq64_pack_f32: Fin = 2 and K = 2 | 3 only");
  const int total = sq_nchunks((int)K) * 4 * 64 * 4;
  PGT_LAUNCH(seq64_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, Wzr, Wh, (int)Fin, (int)K,
             reinterpret_cast<uint32_t*>(Wp));
  return pgt_check_launch("pgt_dcrnn_seq64_pack_f32");
}

extern "C" int pgt_dcrnn_seq64_f32(const pgt_csr* op_o, const pgt_csr* op_i, int64_t E_o, int64_t E_i, int64_t N, const float* X,
                                   int64_t x_stride_b, int64_t x_stride_t, const float* H0, const float* Wp, const float* Wzr,
                                   const float* bzr, const float* Wh, const float* bh, int64_t B, int64_t T, int64_t Fin, int64_t K,
                                   float* out, int64_t out_stride_b, int64_t out_stride_t, float* TSzr, float* TSh,
                                   int64_t seg_stride, int64_t t_stride, float* ZR, float* HT, pgt_stream_t stream) {
  PGT_REQUIRE(op_o && op_i && X && Wp && Wzr && Wh && out, "pgt_dcrnn_seq64_f32: null pointer");
  PGT_REQUIRE(pgt_dcrnn_seq64_fits(N, E_o, E_i, Fin, SQ_O, K), "pgt_dcrnn_seq64_f32: shape not covered (pgt_dcrnn_seq64_fits)");
  PGT_REQUIRE(B >= 0 && T >= 0, "pgt_dcrnn_seq64_f32: negative extent");
  PGT_REQUIRE(TSzr && TSh && ZR && HT, "pgt_dcrnn_seq64_f32: the saved tensors double as the kernel's own parking space (T_0, T_1^i, Z)");
  PGT_REQUIRE(pgt_aligned(Wp, 16), "pgt_dcrnn_seq64_f32: the packed weights must be 16-byte aligned");
  PGT_REQUIRE(pgt_aligned(TSzr, 8) && pgt_aligned(TSh, 8) && seg_stride % 2 == 0 && t_stride % 2 == 0,
              "pgt_dcrnn_seq64_f32: the saved stacks must be 8-byte aligned with even strides");
  if (B == 0 || T == 0) return PGT_OK;
  Seq64Args a;
  a.rp_o = op_o->rowptr; a.col_o = op_o->col; a.val_o = op_o->val;
  a.rp_i = op_i->rowptr; a.col_i = op_i->col; a.val_i = op_i->val;
  a.N = (int)N; a.Fin = (int)Fin; a.K = (int)K; a.T = (int)T; a.B = (int)B; a.nnz_o = (int)E_o; a.nnz_i = (int)E_i;
  a.X = X; a.xs_b = x_stride_b; a.xs_t = x_stride_t; a.H0 = H0;
  a.Wp = reinterpret_cast<const uint32_t*>(Wp); a.Wzr = Wzr; a.Wh = Wh; a.bzr = bzr; a.bh = bh;
  a.out = out; a.os_b = out_stride_b; a.os_t = out_stride_t;
  a.TSzr = TSzr; a.TSh = TSh; a.seg_stride = seg_stride; a.t_stride = t_stride; a.ZR = ZR; a.HT = HT;
  const int nblk = (int)(B < SQ_CUS ? B : SQ_CUS);
  if (K == 3) PGT_LAUNCH(dcrnn_seq64_fwd_kernel<3>, dim3((unsigned)nblk), dim3(SQ_THREADS), stream, a);
  else PGT_LAUNCH(dcrnn_seq64_fwd_kernel<2>, dim3((unsigned)nblk), dim3(SQ_THREADS), stream, a);
  return pgt_check_launch("pgt_dcrnn_seq64_f32");
}

extern "C" int pgt_dcrnn_seq64_pack_bwd_f32(const float* Wzr, const float* Wh, int64_t Fin, int64_t K, float* Wp, pgt_stream_t stream) {
  PGT_REQUIRE(Wzr && Wh && Wp, "pgt_dcrnn_seq64_pack_bwd_f32: null pointer");
  PGT_REQUIRE(Fin == 2 && (K == 2 || K == 3), "pgt_dcrnn_seq64_pack_bwd_f32: Fin = 2 and K = 2 | 3 only");
  const int total = sq_nchunks((int)K) * 4 * 64 * 4;
  PGT_LAUNCH(seq64_pack_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, Wzr, Wh, (int)Fin, (int)K,
             reinterpret_cast<uint32_t*>(Wp));
  return pgt_check_launch("pgt_dcrnn_seq64_pack_bwd_f32");
}

extern "C" int64_t pgt_dcrnn_seq64_bwd_ws_floats(int64_t N, int64_t B) { return (B < SQ_CUS ? B : SQ_CUS) * N * SQ_O; }

extern "C" int pgt_dcrnn_seq64_bwd_f32(const pgt_csr* tp_o, const pgt_csr* tp_i, int64_t E_o, int64_t E_i, int64_t N, const float* dOut,
                                       int64_t g_stride_b, int64_t g_stride_t, const float* out, int64_t out_stride_b,
                                       int64_t out_stride_t, const float* H0, const float* ZR, const float* HT, const float* Wp,
                                       int64_t B, int64_t T, int64_t Fin, int64_t K, float* dPzr, float* dPh, float* dH0, float* ws,
                                       int64_t ws_floats, pgt_stream_t stream) {
  PGT_REQUIRE(tp_o && tp_i && dOut && out && ZR && HT && Wp && dPzr && dPh && ws, "pgt_dcrnn_seq64_bwd_f32: null pointer");
  PGT_REQUIRE(pgt_dcrnn_seq64_fits(N, E_o, E_i, Fin, SQ_O, K), "pgt_dcrnn_seq64_bwd_f32: shape not covered (pgt_dcrnn_seq64_fits)");
  PGT_REQUIRE(B >= 0 && T >= 0, "pgt_dcrnn_seq64_bwd_f32: negative extent");
  PGT_REQUIRE(ws_floats >= pgt_dcrnn_seq64_bwd_ws_floats(N, B), "pgt_dcrnn_seq64_bwd_f32: scratch too small (pgt_dcrnn_seq64_bwd_ws_floats)");
  PGT_REQUIRE(pgt_aligned(Wp, 16) && pgt_aligned(dOut, 16) && pgt_aligned(out, 16) && pgt_aligned(ZR, 16) && pgt_aligned(HT, 16) &&
                  pgt_aligned(dPzr, 16) && pgt_aligned(dPh, 16) && pgt_aligned(ws, 16) && (!H0 || pgt_aligned(H0, 16)) &&
                  (!dH0 || pgt_aligned(dH0, 16)) && g_stride_b % 4 == 0 && g_stride_t % 4 == 0 && out_stride_b % 4 == 0 &&
                  out_stride_t % 4 == 0,
              "pgt_dcrnn_seq64_bwd_f32: operands must be 16-byte addressable");
  if (B == 0 || T == 0) return PGT_OK;
  Seq64BwdArgs a;
  a.rp_o = tp_o->rowptr; a.col_o = tp_o->col; a.val_o = tp_o->val;
  a.rp_i = tp_i->rowptr; a.col_i = tp_i->col; a.val_i = tp_i->val;
  a.N = (int)N; a.Fin = (int)Fin; a.K = (int)K; a.T = (int)T; a.B = (int)B; a.nnz_o = (int)E_o; a.nnz_i = (int)E_i;
  a.dOut = dOut; a.gs_b = g_stride_b; a.gs_t = g_stride_t; a.out = out; a.os_b = out_stride_b; a.os_t = out_stride_t;
  a.H0 = H0; a.ZR = ZR; a.HT = HT; a.Wp = reinterpret_cast<const uint32_t*>(Wp);
  a.dPzr = dPzr; a.dPh = dPh; a.dH0 = dH0; a.park = ws;
  const int nblk = (int)(B < SQ_CUS ? B : SQ_CUS);
  if (K == 3) PGT_LAUNCH(dcrnn_seq64_bwd_kernel<3>, dim3((unsigned)nblk), dim3(SQ_THREADS), stream, a);
  else PGT_LAUNCH(dcrnn_seq64_bwd_kernel<2>, dim3((unsigned)nblk), dim3(SQ_THREADS), stream, a);
  return pgt_check_launch("pgt_dcrnn_seq64_bwd_f32");
}
