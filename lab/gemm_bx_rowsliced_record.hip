// LAB RECORD (not compiled, not shipped): the K-split split-bf16 kernel of csrc/gemm_bx.hip with a ROW-SLICED epilogue — built,
// parity-green on the CPU test double and the GPU, measured and retired in round 3.
//
// Why it was tried: the timeline of the shipped kernel (lab/gemm_bx_trace_lab.hip) shows the part-0 wavefronts spending
// 2.8 - 3.3 us of a 6.3 - 7.4 us iteration issuing the block's stores while the other wavefronts wait at the barrier.  Here
// the part-0 wavefronts leave the block's sums in LDS, a second barrier follows, and wavefront w owns rows 4w .. 4w + 3: a
// lane is a column, every load / store is one whole 256-byte row piece through a per-row buffer descriptor (scalar row
// addressing, masking by the descriptor's range), the gate operands are hand-issued one block ahead and the producers'
// hand-counted waits count them (vmcnt(EPT - 1 + NOP)).
// What it measured (same box, inside the training step, scripts/tree_ab.sh): z | r gates 147 - 149 -> 157 - 158 us,
// candidate gate 159 - 161 -> 154 - 155 us: a wash.  The stores were not slow because four wavefronts issued them, they
// are slow because the memory system accepts them slowly: with the epilogue spread over eight wavefronts every wavefront's
// four rows still take 1.5 - 2.8 us, and the k-loops stretch instead.  The take-apart of the shipped kernel
// (lab/gemm_bx_trace_lab_<mask>) says why: its phases ADD (skeleton 57 us + MFMAs 52 + A stream 27 +
// epilogue traffic 32 ~ the whole 190 us), the A stream alone runs at 3.3 TB/s — 42 KB in flight per CU, the producers'
// registers — and one workgroup per CU with a barrier per block cannot put one phase under another.
//
// The text below replaces `gemm_bx_kernel` in csrc/gemm_bx.hip (plus `template <int V> struct BxInt { static constexpr int
// value = V; };` and a 64-byte LDS tail for `tile_seen`).

#if 0
template <int KSTEPS, int WN, int EPI, bool Q4>
__global__ __launch_bounds__(512, 1) void gemm_bx_kernel(PgtGemmArgs g, int n_blocks) {
  constexpr int BM = 32, KP = KSTEPS * 16, SROW = KP * 2 + 16, PLANE = BM * SROW, BUF = 3 * PLANE;
  constexpr int EPT = (KP / 2) / 8;                       // float pairs per producer thread and block (8 threads per row)
  constexpr int PART = 64 * 16 * WN * 4;                  // a column's partial sums, accumulator layout
  constexpr int NPART = Q4 ? 4 : 2, NCOL = Q4 ? 2 : 4;     // parts of K x column blocks = the eight wavefronts
  constexpr int KQ = KSTEPS / NPART, KMAX = KSTEPS - (NPART - 1) * KQ;   // k-steps of a part / of the last part
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
  constexpr int NOP = !ROWS || EPI == 0 ? 0 : (EPI == 1 ? 4 : 8);   // gate-operand loads per wavefront and block
  static_assert((KP / 2) % 8 == 0 && KQ >= 1 && (!Q4 || WN == 1) && (EPI == 0 || WN == 1), "shape");
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
  const int kbase = part * KQ, ksteps = part == NPART - 1 ? KMAX : KQ;
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

  // ---- the row-sliced epilogue (ROWS).  Column data of a lane: output byte offset inside a row of C (segments folded in),
  // byte offset of its gate operand inside a row of H (zr gate: the reset half only); 0xfffffff0 = past every
  // descriptor: reads zero / is dropped.
  uint32_t c_off[CPL], h_off[CPL];
#pragma unroll
  for (int s = 0; s < CPL; ++s) {
    const int gn = lane + 64 * s;
    const bool live = ROWS && gn < g.N;
    const int js = live ? gn / g.c_seg_n : 0;
    c_off[s] = live ? (uint32_t)(((int64_t)js * g.c_seg_stride + (gn - js * g.c_seg_n)) * 4) : 0xfffffff0u;
    if (EPI == 1) h_off[s] = (live && gn >= g.eO) ? (uint32_t)((gn - g.eO) * 4) : 0xfffffff0u;
    else h_off[s] = live ? (uint32_t)(gn * 4) : 0xfffffff0u;
  }
  const int zero_s = BX_SGPR(0);
  float eh[4], ez[4];
  // one row of an [M, .] array as a buffer: `bytes` from its first element, nothing for rows past M
  auto row_rsrc = [&](const float* p, int64_t ld, int gm, int64_t bytes) {
    const bool ok = gm < g.M && p != nullptr;
    return bx_make_rsrc(p + (ok ? (int64_t)gm * ld : 0), ok ? bytes : 0);
  };
  // gate operands of this wavefront's four rows of block `b`: always NOP load instructions (the hand-counted waits of the
  // producers count them), rows / columns that do not exist read zero through the descriptor
  auto e_issue_rows = [&](int b) {
    if (BX_LAB_SKIP(16)) return;
    if constexpr (NOP > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gm = BX_SGPR(b * BM + 4 * wave + j);
        const BxRsrc rh = row_rsrc(g.eH, g.eldh, gm, (int64_t)g.eO * 4);
        BX_LOAD1S(eh[j], h_off[OPG], rh, zero_s);
        if constexpr (EPI == 2) {
          const BxRsrc rz = row_rsrc(g.eZ, 2 * (int64_t)g.eO, gm, (int64_t)g.eO * 4);
          BX_LOAD1S(ez[j], h_off[OPG], rz, zero_s);
        }
      }
    }
  };
  // NY: loads of this wavefront that are younger than its gate operands when they are due (a producer: the EPT loads of the
  // block after next; everyone else: none)
  auto row_epilogue = [&](int b, auto NY) {
    constexpr int ny = decltype(NY)::value;
    float v[4][CPL];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < CPL; ++s) {
        const int gn = lane + 64 * s;
        v[j][s] = *reinterpret_cast<const float*>(stage_part + (gn >> 5) * PART + ((4 * wave + j) * 32 + (gn & 31)) * 4);
      }
    if (lane == 0) tile_seen[wave] = n_iter + 1;       // after the reads above: a wavefront's LDS operations complete in order
    BX_TRACE(5);
    if constexpr (NOP > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        BX_WAIT(ny, eh[i]);
        if constexpr (EPI == 2) BX_WAIT(ny, ez[i]);
      }
    }
    // out0 of the candidate gate in a two-level row layout (pgt_rowmap: H_t straight into the [B, T, N, O] result): one
    // uniform division per block, a compare per row (a block of 32 rows crosses at most one period boundary when
    // period >= 32)
    int64_t o0_q = 0;
    int o0_rem = 0;
    if constexpr (EPI == 2) {
      if (g.e0_period > 0) {
        const uint32_t q = (uint32_t)BX_SGPR((int)((uint32_t)(b * BM) / (uint32_t)g.e0_period));
        o0_q = (int64_t)q * g.e0_hi;
        o0_rem = b * BM - (int)(q * (uint32_t)g.e0_period);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * wave + j;
      const int gm = BX_SGPR(b * BM + row);
      const BxRsrc rc = row_rsrc(g.C, g.ldc, gm, 0x7fffff00);
      if (BX_LAB_SKIP(4)) continue;
      if constexpr (EPI == 0) {
#pragma unroll
        for (int s = 0; s < CPL; ++s) BX_STORE1S(v[j][s], c_off[s], rc, zero_s);
      } else if constexpr (EPI == 1) {
        const BxRsrc rx = row_rsrc(g.eX ? g.eX + g.efin : nullptr, g.eldx, gm, (int64_t)g.eO * 4);
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
          const float x = bx_sigmoidf(v[j][s]);
          BX_STORE1S(x, c_off[s], rc, zero_s);
          if (s == OPG) BX_STORE1S(eh[j] * x, h_off[s], rx, zero_s);   // the update half's lanes lie past the descriptor: dropped
        }
      } else {
        int64_t o0;
        if (g.e0_period >= BM) {
          const int rr = o0_rem + row;
          const bool wrap = rr >= (int)g.e0_period;
          o0 = o0_q + (wrap ? g.e0_hi : 0) + (int64_t)(rr - (wrap ? (int)g.e0_period : 0)) * g.eld0;
        } else {
          o0 = pgt_row_off(gm < g.M ? gm : 0, g.eld0, g.e0_period, g.e0_hi);
        }
        const bool ok = gm < g.M;
        const BxRsrc r0 = bx_make_rsrc(g.eO0 + (ok ? o0 : 0), ok ? (int64_t)g.N * 4 : 0);
        const BxRsrc r1 = row_rsrc(g.eO1, g.eld1, gm, (int64_t)g.N * 4);
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
          const float x = bx_tanhf(v[j][s]);
          const float side = pgt_gru_blend(ez[j], eh[j], x);
          BX_STORE1S(x, c_off[s], rc, zero_s);
          BX_STORE1S(side, h_off[s], r0, zero_s);
          BX_STORE1S(side, h_off[s], r1, zero_s);
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
      for (int i = 0; i < KMAX; ++i) {
        if (i < ksteps) {
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
        if (i < KQ) {
#pragma unroll
          for (int t = i * EPT / KQ; t < (i + 1) * EPT / KQ; ++t) {
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
      for (int i = 0; i < KQ; ++i) {
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
      for (int i = 0; i < KQ; ++i) {
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

#endif
