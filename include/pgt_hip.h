/*
 * pgt_hip.h — C ABI of libpgt_hip.so, the MI355X (gfx950 / CDNA4) native
 * message-passing hot path behind torch_geometric_temporal.nn.
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  No torch types,
 * no exceptions across the boundary.  All pointers are DEVICE pointers unless
 * a parameter says "host".  Every call is stream-ordered on `stream`
 * (a hipStream_t passed as void*), allocates nothing (the caller owns every
 * buffer, including the scratch `ws`), and is safe to capture into a hipGraph.
 * Return value: PGT_OK (0) or a negative PGT_ERR_* code; pgt_last_error()
 * returns a thread-local message for the last failing call.
 *
 * Reference interfaces replaced (paths relative to the reference checkout,
 * benedekrozemberczki/pytorch_geometric_temporal @ 2025-10-03):
 *
 *   pgt_dconv_prep        torch_geometric_temporal/nn/recurrent/dcrnn.py:59-77  (DConv graph prep:
 *                         to_dense_adj, degrees, reciprocals, norm gathers, dense_to_sparse(adj^T))
 *                         and dcrnn.py:277-290 (BatchedDConv scatter_add_/argsort form)
 *   pgt_gcn_prep          PyG gcn_norm as called from GCNConv: nn/recurrent/temporalgcn.py:38-70,
 *                         nn/recurrent/evolvegcno.py:88-90
 *   pgt_cheb_prep         PyG ChebConv.__norm__ / nn/attention/astgcn.py:82-110 (ChebConvAttention.__norm__)
 *   pgt_cheb_prep_graphs  the same with one lambda_max per graph of a disjoint batch: astgcn.py:97-98
 *                         (`lambda_max = lambda_max[batch[edge_index[0]]]`), exercised by test/attention_test.py:205-217
 *   pgt_spmm_csr_f32      MessagePassing.propagate(aggr="add") + message():
 *                         dcrnn.py:39-40,86-87,95-100,300-313; astgcn.py:169-175,185-190; evolvegcno.py:95-101
 *                         (index_select -> norm*x_j -> scatter_add, fused, with the 2*P*T - T0 epilogue of dcrnn.py:96,100)
 *   pgt_spmm_ellw_f32     the same propagate call sites on a locality-ordered graph (ELLW layout, LDS window)
 *   pgt_dconv_stack_slab* the K-hop recursion dcrnn.py:85-106 (all propagate calls of one DConv) for small graphs
 *   pgt_gemm_f32          the dense feature transforms: dcrnn.py:81-83,88-92,101-105 (torch.matmul on weight[d][k]);
 *                         PyG Linear in GCNConv/ChebConv; temporalgcn.py:84,90,96 (linear_{z,r,h})
 *   pgt_gemm_tn_acc_f32   autograd of the above w.r.t. the weights (torch autograd in the reference)
 *   pgt_att_*             SpatialAttention / TemporalAttention of ASTGCN: nn/attention/astgcn.py:226-262, :291-328
 *   pgt_window_gather_f32 signal/index_dataset.py:32-57 (the index-batch windows of a resident series)
 *   pgt_relu_linear_*     the per-node read-out of the reference's models, `self.linear(F.relu(h))` with a torch.nn.Linear(hidden,
 *                         1 .. 4): examples/indexBatching/tgcn/metr_la_main.py:43-45, examples/recurrent/dcrnn_example.py:27-31
 *   pgt_gru_*             the GRU gate chains: dcrnn.py:172-192,406-427; temporalgcn.py:82-102
 *   pgt_lstm_gates*       the LSTM gate chains: gconv_lstm.py:138-172 (peepholes), gc_lstm.py:138-169
 */
#ifndef PGT_HIP_H_
#define PGT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGT_OK 0
#define PGT_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, misaligned, unsupported) */
#define PGT_ERR_LAUNCH (-2)    /* HIP reported an error at launch */
#define PGT_ERR_WORKSPACE (-3) /* scratch buffer too small */

#define PGT_ABI_VERSION 17

typedef void* pgt_stream_t; /* hipStream_t */

/* CSR operator by destination row: y[i,:] = sum_{q in [rowptr[i], rowptr[i+1])} val[q] * x[col[q], :] */
typedef struct pgt_csr {
  int32_t* rowptr; /* [n_rows + 1] */
  int32_t* col;    /* [nnz]  source node of each slot */
  float* val;      /* [nnz]  per-slot coefficient */
} pgt_csr;

/* Everything DConv / BatchedDConv derive from (edge_index, edge_weight); all buffers caller-allocated.
 * fwd_o/fwd_i are the out-/in-direction diffusion operators used by propagate (dcrnn.py:86-87);
 * bwd_o/bwd_i are their transposes (used for the gradient w.r.t. the node features). */
typedef struct pgt_dconv_graph {
  pgt_csr fwd_o;    /* nnz = E */
  pgt_csr fwd_i;    /* nnz = E */
  pgt_csr bwd_o;    /* nnz = E */
  pgt_csr bwd_i;    /* nnz = E */
  float* deg_out;   /* [N]  scatter_add(edge_weight, row)  (dcrnn.py:61-64 / :279) */
  float* deg_in;    /* [N]  scatter_add(edge_weight, col)  (dcrnn.py:65-68 / :280) */
  int32_t* info;    /* [4]  info[0] = #duplicate (row,col) pairs, info[1] = #zero weights,
                            info[2] = #edge endpoints outside [0,N), info[3] = #slots of P_o / P_i with a
                            non-finite coefficient (1 / deg of a node without out- / in-edges) */
} pgt_dconv_graph;

/* GCN / Chebyshev operators: one CSR for propagate, one transposed CSR for the feature gradient.
 * Slot capacity is E + 2N; the number of live slots is rowptr[N] (dropped self-loops sort past the end). */
typedef struct pgt_sym_graph {
  pgt_csr fwd;      /* capacity E + 2N */
  pgt_csr bwd;      /* capacity E + 2N */
  float* deg;       /* [N] */
  int32_t* info;    /* [4] info[2] = #edge endpoints outside [0,N), info[3] = #batch labels outside [0,n_graphs) (pgt_cheb_prep_graphs) */
} pgt_sym_graph;

/* Two-level row layout of an [M, W] operand: row m starts at base + (m / period) * stride_hi + (m % period) * ld floats,
 * with ld the operand's ordinary row-stride argument (period = 0, or a NULL map: a plain matrix, row m at base + m * ld).
 * It lets one time step of the DCRNN sequence read and write the reference's [B, T, N, O] tensors in place
 * (BatchedDCRNN.forward returns torch.stack(outputs, dim=1), dcrnn.py:463-475, and autograd hands the gradient back in
 * that layout): batch-major rows m = b * N + n -> period = N, stride_hi = T * N * O, ld = O, base = out + t * N * O;
 * node-major rows m = n * B + b -> period = B, stride_hi = O, ld = T * N * O.  period, M < 2^31. */
typedef struct pgt_rowmap {
  int64_t period;
  int64_t stride_hi;
} pgt_rowmap;

int pgt_abi_version(void);
const char* pgt_last_error(void);
/* "gfx950" for the product library; "emu" for the CPU test double built under tests/. */
const char* pgt_build_target(void);

/* Schedule switches for A/B measurements and for forcing a kernel onto small test problems; every key selects
 * between kernels that compute the same sums (results differ at most in fp32 summation order), the defaults are the
 * measured-best configuration.  GEMM: "gemm_db" (pipelined tile kernel: 1 where it applies, 2 always, 0 never),
 * "gemm_db64", "gemm_dbp" (persistent deferred-store tiles: 1 / 2 = on three workgroups / 0), "gemm_skinny"
 * (streaming kernels for an extent <= 4: 1 from 1024 rows / 2 always / 0), "gemm_small_tiles", "gemm_tn_pipe",
 * "gemm_tn_fullk", "gemm_small_fill", "gemm_bx" (split-bf16 kernel on the bf16 matrix pipe — fp32 operands as three
 * bf16 pieces, six piece products, fp32 accumulation; at least as close to the exact product as the fp32 kernels:
 * 1 where it wins / 2 at any size / 0 never), "gemm_bx_sym" (0: short-K products on its K-split variant), "gemm_bx_tn_pc" / "gemm_bx_sym_pc" (0: the
 * weight-gradient / K <= 64 short products on the kernels whose wavefronts are all alike instead of producers and consumers; the
 * same bits).  Diffusion
 * stack: "slab_pairs", "slab_split", "slab_wpc", "slab_threads" (pgt_dconv_stack_slab_plan), "slab_quad" (0: 64 / 66-column blocks on the pair-layout kernels), "slab_gu" (2 | 4 LDS reads in flight in their backward gathers).  Aggregation: "spmm_tile_rows", "spmm_unroll", "spmm_tile_xcd",
 * "spmm_tile_nt" (streaming stores: 1 = for outputs >= 32 MiB / 2 always / 0), "spmm_ellw" (0: pgt_spmm_ellw_f32 runs
 * the CSR kernels), "spmm_ellw_rows" / "spmm_ellw_cus" / "spmm_ellw_cfg" (test hooks of pgt_ellw_plan), "tgcn_rows" (0: the column-per-lane T-GCN cell kernels of round 4 for every shape), "tgcn_wgs" (n > 0: at most n workgroups per T-GCN cell launch).  Returns PGT_ERR_INVALID for an unknown key — and for "tgcn_probe" (forward-kernel variants with parts switched off, WRONG results by design) unless the library was built with -DPGT_LAB_PROBES.  Not thread-safe: call between launches. */
int pgt_tune(const char* key, int value);

/* ---------------------------------------------------------------- graph preparation */

/* Scratch bytes needed by the *_prep calls for a graph with E edges and N nodes. */
size_t pgt_prep_workspace_bytes(int64_t E, int64_t N);

/* edge_index: int64 [2,E] row-major (row = edge_index[0] = source, col = edge_index[1] = target).
 * edge_weight: float [E] or NULL (unit weights, dcrnn.py:59 via to_dense_adj(edge_attr=None)). */
int pgt_dconv_prep(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N,
                   const pgt_dconv_graph* out, void* ws, size_t ws_bytes, pgt_stream_t stream);

/* gcn_norm: add_remaining_self_loops(fill = improved && edge_weight ? 2 : 1: with NULL weights PyG adds the loops before it
 * creates the unit weights, so they get 1) when add_self_loops != 0, deg = scatter_add(w, col),
 * w' = deg^-1/2[row] * w * deg^-1/2[col] (inf -> 0).  Live slots: (#non-loop edges) + N when add_self_loops, else E. */
int pgt_gcn_prep(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N,
                 int improved, int add_self_loops, const pgt_sym_graph* out, void* ws, size_t ws_bytes,
                 pgt_stream_t stream);

/* ChebConv.__norm__: normalization 0 = None, 1 = "sym", 2 = "rw"; scaled Laplacian 2L/lambda_max - I.
 * lambda_max: > 0 as given; NaN = "not passed" (2.0 for "sym"; 2*max(L) computed on the device otherwise).
 * variant 0 = PyG ChebConv (STConv, stgcn.py:115-121): out[col] += norm*x[row], "-1" folded into the diagonal slot;
 * variant 1 = in-tree ChebConvAttention.__norm__ (astgcn.py:82-110,166-175): a second set of N "-1" diagonal
 *             slots is appended and propagate runs on the transposed edge list (out[row] += norm*x[col]).
 * Self-loops of the input are removed (get_laplacian / remove_self_loops). */
int pgt_cheb_prep(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N,
                  int normalization, float lambda_max, int variant, const pgt_sym_graph* out, void* ws,
                  size_t ws_bytes, pgt_stream_t stream);

/* pgt_cheb_prep for a disjoint batch of graphs with one lambda_max each (astgcn.py:97-98, PyG ChebConv.__norm__):
 * batch [N] int64 = graph label of every node, lambda_max [n_graphs] fp32 (device); every entry of the Laplacian
 * (edges and the appended diagonal) is scaled by 2 / lambda_max[batch[row]], row = edge_index[0] of that entry.
 * Labels outside [0, n_graphs) are counted in out->info[3] and their entries become NaN. */
int pgt_cheb_prep_graphs(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N,
                         int normalization, const int64_t* batch, const float* lambda_max, int64_t n_graphs,
                         int variant, const pgt_sym_graph* out, void* ws, size_t ws_bytes, pgt_stream_t stream);

/* ---------------------------------------------------------------- aggregation (the graded kernel) */

/* Y[i, 0:F] = alpha * sum_q val[q] * X[col[q], 0:F]  +  beta * T[i, 0:F]      (T may be NULL => beta term dropped)
 * X, Y, T are row-major with row strides ldx/ldy/ldt (in floats).  Y may alias T; Y must not alias X.
 * A batch of B graphs sharing one topology is laid out node-major ([N][B][C], F = B*C) so one launch covers it.
 * Deterministic: per-row sequential accumulation in slot order, no atomics. */
int pgt_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                     const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt,
                     float alpha, float beta, int64_t F, pgt_stream_t stream);

/* ELLW: the layout of a locality-ordered operator for F = 64 (spmm_ellw64_kernel, csrc/spmm.hip) — what replaces
 * `propagate` on a bandwidth-reduced node numbering (road-sensor graphs; the north-star shape N = 200 000, in-degree 8:
 * 21 us = 0.69 of 8 TB/s against 33 us for the CSR row tiles).  Rows are cut into tiles of `tile_rows` rows; every row
 * has `width` slots (the longest row rounded up to a multiple of 8; padding slots point at a zero row); slot j of row
 * i is CSR slot rowptr[i] + j, stored as the 16-bit offset of its source row inside the tile's window
 * [t * tile_rows - halo, (t + 1) * tile_rows + halo), or 0xFFFF when the source lies outside it (then read through
 * the CSR arrays: any operator is represented exactly).  Coefficients: `vals` (one per slot), or — when
 * val[q] == scale[col[q]] for every slot, as for DConv's P_o = A D_out^-1 (dcrnn.py:70-73) — the per-source table
 * `scale` alone (the coefficient stream is dropped and the products are rounded once, as norm * x_j is). */
typedef struct pgt_ellw {
  const uint16_t* slots;  /* [n_tiles * tile_rows * width] */
  const float* vals;      /* [n_tiles * tile_rows * width], or NULL in source-scale mode */
  const float* scale;     /* [n_rows] coefficient per SOURCE row, or NULL; exactly one of vals / scale is set */
  int32_t tile_rows, halo, width;
  int32_t config;         /* launch shape, from pgt_ellw_plan: 1 = one 1024-thread workgroup per CU (456 window rows),
                             2 = two 512-thread workgroups per CU (240 window rows) */
  int64_t n_tiles;
  const int32_t* far_col; /* [n_tiles * far_rows] the DISTINCT sources outside a tile's window that get an LDS row of their
                             own (a hash set: entry k holds the source whose slots read window_rows + 1 + k; -1 = unused
                             entry), or NULL: every out-of-window slot is 0xFFFF and comes through the CSR.  This is what
                             carries numberings whose tiles are compact without being a band: a 2-D mesh numbered along a
                             space-filling curve names ~85 outside rows per 392-row tile (its ring) */
  int32_t far_rows;       /* from pgt_ellw_plan (depends on config and mode) */
  const int32_t* order;   /* [n_rows] or NULL.  Set: the layout lives in a RENUMBERED row space (pgt_tile_order_host) — layout
                             row p is row order[p] of X / Y / T; slots, scale, far_col and the CSR arrays passed along are all
                             in layout numbering; config must be 3 (halo 0).  The product is still Y = A X in the CALLER's
                             numbering: the kernel reads and writes whole 256-byte rows through `order`, no permutation pass */
  /* Hubs (all NULL / 0: none).  Rows longer than `width` are left out of the layout (pgt_ellw_build info[2]).  With these tables
   * pgt_spmm_ellw_f32 produces them itself at F = 64: the slots of hub h = hub_rows[h] are cut into hub_split pieces of at most
   * P = 2 * (threads of the launch shape / 16) = 128 (config 1, 3) / 64 (config 2) slots, piece s of hub h rides with tile
   * h * hub_split + s (entry j of that tile: hub_col / hub_val [(h * hub_split + s) * P + j]; hub_col -1 = unused entry), every
   * wavefront of the tile leaves one partial row in hub_partial and a second, tiny launch adds them in a fixed order.  At other
   * widths the tables are ignored and the caller produces the hub rows with pgt_spmm_csr_rows_f32, as it does without tables.
   * hub_col and hub_rows name rows of X / Y: the CALLER's numbering, also on a renumbered layout (`order`). */
  const int32_t* hub_col;  /* [n_hub * hub_split * P] */
  const float* hub_val;    /* [n_hub * hub_split * P] */
  const int32_t* hub_rows; /* [n_hub] */
  float* hub_partial;      /* [n_hub * hub_split * (threads / 64) * 64] workspace written by every launch: launches that share a
                              layout must be ordered (one stream / one hipGraph) */
  int32_t n_hub, hub_split; /* n_hub * hub_split <= n_tiles */
  const int32_t* far_src; /* [n_tiles * far_rows] or NULL; a renumbered layout only: far_src[e] = order[far_col[e]] (-1 where
                             far_col is) — the outside rows' X rows by the CALLER's numbering, so that the kernel reaches them in
                             two dependent loads like its window rows instead of three (far_col -> order -> X) */
} pgt_ellw;

/* Host-only: tile height / slot width / tile count for an operator with `n_rows` rows whose longest row has
 * `max_row_len` slots and whose sources lie (mostly) within `halo` rows of their destination; the tile height fills
 * whole rounds of the resident workgroups (config 1: rows of <= 8 slots, <= 16 in source-scale mode; config 2: wider rows) of the current device.
 * `far_rows` = entries per tile of the out-of-window table (what the LDS budget of the launch shape leaves: more in
 * source-scale mode, which has no coefficient block).
 * halo 0 plans the layout of a renumbered operator (config 3: a 400-row window without halo, 208 / 160 table entries).
 * PGT_ERR_INVALID when the layout does not apply (rows longer than 32 slots, halo > 116). */
int pgt_ellw_plan(int64_t n_rows, int32_t halo, int32_t max_row_len, int32_t source_scaled, int32_t* tile_rows,
                  int32_t* width, int32_t* config, int64_t* n_tiles, int32_t* far_rows);

/* Fill `slots` / `vals` (each n_tiles * tile_rows * width entries; `vals` may be NULL) from the CSR operator, with the
 * geometry in `op` (its pointers are ignored).  `scale` (float [n_rows], may be NULL) receives the candidate
 * per-source table scale[col[q]] = val[q]; it also states the mode the layout is built for (NULL: per-slot mode,
 * op->far_rows must be the per-slot plan's).  `far_col` (int32 [n_tiles * op->far_rows]) / `far_cnt` (int32 [n_tiles],
 * scratch: the number of distinct outside sources that got a row) receive the out-of-window table; both NULL: no table.  info (int32 [4], device): [0] = slots outside their
 * window, [1] = slots whose val differs bitwise from scale[col] (0 = the source-scale mode applies; otherwise rebuild
 * with scale = NULL and the per-slot plan), [2] = rows longer than `width`: LEFT OUT of the layout (pgt_spmm_ellw_f32 neither reads
 * nor writes their rows of Y; the caller either produces exactly those rows with pgt_spmm_csr_rows_f32 — hubs — or drops the layout),
 * [3] = out-of-window slots that did not fit their tile's table (0xFFFF: served through the CSR at run time — correct,
 * slower). */
int pgt_ellw_build(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int64_t nnz,
                   const pgt_ellw* op, uint16_t* slots, float* vals, float* scale, int32_t* far_col, int32_t* far_cnt,
                   int32_t* info, pgt_stream_t stream);

/* pgt_spmm_csr_f32's contract on the ELLW layout of the same operator (the CSR arrays it was built from are passed
 * along: they serve out-of-window slots, and shapes the window kernel does not cover — F not a multiple of 64, operands
 * that are not 16-byte aligned — run the CSR kernels; F = 64 k (node-major batches) runs k column chunks per tile).  Source-scale mode accumulates rounded products with rounded adds in
 * slot order (the reference's `norm * x_j` then scatter-add); per-slot mode is the CSR kernels' fmaf chain, bit for bit. */
int pgt_spmm_ellw_f32(const pgt_ellw* op, const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                      const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt, float alpha,
                      float beta, int64_t F, pgt_stream_t stream);

/* HOST pointers in and out (graph preparation, once per graph; O(E), ~40 ms at 200 000 rows x 8 slots).  A numbering of
 * the rows of a CSR operator whose runs of `tile_rows` consecutive rows are compact patches of the graph — what
 * spmm_ellw64_kernel needs and what the caller's numbering need not be: the reference aggregates over whatever edge list
 * it is given (dcrnn.py:300-313), a sensor graph arrives in file order.  Patches are grown one after the other, always
 * adding the unassigned row with the most neighbours already inside the patch.  order[p] = the caller's row at layout
 * position p (int32 [n_rows]); rowptr_p / col_p = the operator in layout numbering, every row keeping its slots in the
 * caller's order (so the sums round as on the caller's CSR); slot_p[q'] = the caller's slot behind layout slot q'
 * (val_p = val[slot_p]).  Feed them to pgt_ellw_plan (halo 0) / pgt_ellw_build and set pgt_ellw.order.  order_given != 0:
 * `order` is an INPUT — the order found for another operator of the same graph (its transpose, the other diffusion direction:
 * the same undirected neighbourhoods) — and only the operator in that numbering is produced. */
int pgt_tile_order_host(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t tile_rows, int32_t order_given,
                        int32_t* order, int32_t* rowptr_p, int32_t* col_p, int32_t* slot_p);

/* out4[0] = #slots with |col - row| <= 32, out4[1] = #slots with |col - row| <= 96, out4[2] = slots of the longest row,
 * out4[3] = #rows longer than long_len (device int32[4]); the first min(out4[3], long_cap) of those rows are listed in
 * long_rows (int32 [long_cap], any order; NULL / long_cap 0: no list).  The host decides from these, once per prepared
 * graph, whether the ELLW layout applies and which rows go to the long-row kernel. */
int pgt_csr_locality(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t* out4, int32_t* long_rows,
                     int64_t long_cap, int32_t long_len, pgt_stream_t stream);

/* pgt_spmm_csr_f32 for operators with hubs: rows longer than long_len (the n_long rows listed in long_rows, from
 * pgt_csr_locality) are skipped by the row tiles and produced by one 1024-thread workgroup each, the lane groups taking
 * the row's slots round-robin and meeting in LDS (a fixed order: deterministic; the sums of those rows run in a
 * different order than the sequential chain).  Without it a 2 000-slot row costs 660 us and a 20 000-slot row 6.4 ms at
 * N = 200 000 (one lane group walks the row).  F beyond the tile kernels' 256 floats runs pgt_spmm_csr_f32 unchanged. */
int pgt_spmm_csr_long_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                          const int32_t* long_rows, int64_t n_long, int32_t long_len, const float* X, int64_t ldx,
                          float* Y, int64_t ldy, const float* T, int64_t ldt, float alpha, float beta, int64_t F,
                          pgt_stream_t stream);

/* Only the n_listed rows of Y named in `rows` (int32, device; each row once), under pgt_spmm_csr_f32's contract for those rows: one
 * 1024-thread workgroup per row, as in pgt_spmm_csr_long_f32 (the same kernel: the same sums).  The second launch of an operator
 * whose ELLW layout leaves its hub rows out (pgt_ellw_build info[2]): the window kernel for the ordinary rows + this for the hubs.
 * F / (widest vector the operands allow) must not exceed 64 lanes. */
int pgt_spmm_csr_rows_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const int32_t* rows,
                          int64_t n_listed, const float* X, int64_t ldx, float* Y, int64_t ldy, const float* T, int64_t ldt,
                          float alpha, float beta, int64_t F, pgt_stream_t stream);

/* Same with a per-batch dense attention multiplier (ChebConvAttention hop 1, astgcn.py:157,169-171):
 * rows are node-major [N][B][C]; the coefficient of slot q of row i for batch b is val[q] * S[b, i, col[q]]
 * (S is the [B,N,N] spatial attention; Att_norm = norm * spatial_attention[:, row, col]).
 * transpose_s != 0: (rowptr, col, val) is the TRANSPOSED operator (feature gradient) and the coefficient is
 * val[q] * S[b, col[q], i]. */
int pgt_spmm_csr_att_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* S,
                         int64_t n_rows, int64_t B, int64_t C, const float* X, float* Y, int transpose_s,
                         pgt_stream_t stream);

/* Gradient of the above w.r.t. the attention:  dS[b, i, col[q]] += val[q] * <G[i,b,:], X[col[q],b,:]>  for every slot
 * q of every row i (G = gradient of the aggregation's output, X = its input, both [N][B][C]; dS [B,N,N] is
 * accumulated into with fp32 atomics — zero it first).  (torch autograd of astgcn.py:157 in the reference.) */
int pgt_sddmm_att_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int64_t B,
                      int64_t C, const float* G, const float* X, float* dS, pgt_stream_t stream);

/* ---------------------------------------------------------------- small-graph diffusion stack (LDS-resident) */

/* The whole diffusion stack of DConv / BatchedDConv (dcrnn.py:85-106) for a batch of samples sharing one small graph,
 * in ONE launch.  Rows are BATCH-major (m = b*N + n): sample b's [N, C] block of stack segment s starts at
 * TS + s*seg_stride + b*N*C (unit column stride, row stride C).  Segment order: [T0 | T1o T1i | T2o T2i].
 * Reads segment 0, writes segments 1..2K-2:  T1 = P T0,  T2 = 2 P T1 - T0.   K = 2 or 3 (K < 2: no-op).
 * A sample's block — or a column window of it — and both operators must fit a CU's LDS: pgt_dconv_stack_slab_fits() != 0
 * (METR-LA, PeMS-BAY, Chickenpox, EnglandCovid do); otherwise PGT_ERR_INVALID — run the hops with pgt_spmm_csr_f32.
 * nnz_o / nnz_i: number of slots of the two operators (host values; E for a DConv graph). */
int pgt_dconv_stack_slab_fits(int64_t N, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i);
/* The launch shape the two entry points below will use for a batch of n_samples (reporting / tests): plan[0] = column
 * windows per sample (the recursion is column-independent: a work item is sample x window; used when the batch has fewer
 * samples than the device has CUs — B = 64 at METR-LA shape: 20.7 -> 12.1 us — and for blocks that only fit the LDS
 * column by column; 1 with plan[3] = 0: the whole-sample kernels, one 1024-thread workgroup per sample), plan[1] =
 * workgroups per CU, plan[2] = threads per workgroup, plan[3] = tasks per thread; all 0 when the shape is not supported.
 * pgt_tune keys "slab_split" (1 = planned, 0 = whole-sample kernels only, n >= 2 = n windows), "slab_wpc", "slab_threads". */
int pgt_dconv_stack_slab_plan(int64_t N, int64_t n_samples, int64_t C, int64_t K, int64_t nnz_o, int64_t nnz_i,
                              int32_t* plan);
int pgt_dconv_stack_slab_f32(const pgt_csr* fwd_o, const pgt_csr* fwd_i, int64_t nnz_o, int64_t nnz_i, int64_t N,
                             int64_t n_samples, int64_t C, int64_t K, float* TS, int64_t seg_stride,
                             pgt_stream_t stream);
/* Adjoint of the above on the TRANSPOSED operators: G holds d/d[T0 | T1o T1i | T2o T2i] in the same layout; on exit
 * segment 0 holds d/dT0 (segments 1.. are left as they were).  folded != 0: the "- Tx_0" adjoint (G0 -= G2o + G2i)
 * was already applied through the weights of the feature-gradient GEMM. */
int pgt_dconv_stack_slab_bwd_f32(const pgt_csr* bwd_o, const pgt_csr* bwd_i, int64_t nnz_o, int64_t nnz_i, int64_t N,
                                 int64_t n_samples, int64_t C, int64_t K, float* G, int64_t seg_stride, int folded,
                                 pgt_stream_t stream);

/* ---------------------------------------------------------------- dense feature transform (fp32 sums on the matrix cores) */

/* C(m,n) = sum_j A_j[M, seg_k] * Bw[j*seg_k : (j+1)*seg_k, 0:N]  (+ bias[N])  (+ C if accumulate)
 * A_j = A + j*a_seg_stride, row stride lda.  Bw element (k,n) at Bw[k*sbk + n*sbn]  (sbk=ldb,sbn=1: NN; sbk=1,sbn=ldb: NT).
 * The output may be split into column segments of c_seg_n columns: C(m,n) lives at
 * C[(n / c_seg_n)*c_seg_stride + m*ldc + n % c_seg_n]  (c_seg_n = N, c_seg_stride = 0 for a plain matrix).
 * Arithmetic: fp32 sums on the matrix cores.  Most shapes run v_mfma_f32_32x32x2_f32 (an exact fmaf chain); tall products
 * (>= 8192 rows) with K <= 336 run the split-bf16 kernel (three bf16 pieces per fp32 operand, six piece products, fp32
 * accumulation; measured closer to the exact product than the fmaf chain; an INFINITE input element turns its output row
 * into nan instead of +-inf / nan).  pgt_tune("gemm_bx", 0) keeps every shape on the fmaf-chain kernels. */
int pgt_gemm_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                 const float* Bw, int64_t sbk, int64_t sbn, float* C, int64_t ldc, int64_t c_seg_stride,
                 int64_t c_seg_n, const float* bias, int64_t M, int64_t N, int accumulate,
                 pgt_stream_t stream);

/* dW[j*seg_k + c, n] += sum_m A_j[m, c] * G[m, n]      (weight gradient; fp32 atomics into dW)
 * db[n] += sum_m G[m, n] when db != NULL. */
int pgt_gemm_tn_acc_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                        const float* G, int64_t ldg, float* dW, int64_t lddw, float* db, int64_t M, int64_t N,
                        pgt_stream_t stream);

/* The same sums without float atomics: every workgroup STORES its partial dW / db into `ws` and a second pass adds the
 * row slabs in index order, so the result is bitwise reproducible run to run (pgt_gemm_tn_acc_f32's fp32 atomics make
 * dW depend on the order in which workgroups retire).  ws: device scratch of pgt_gemm_tn_det_ws_bytes(...) bytes,
 * 16-byte aligned, owned by the caller.  Costs one extra pass over <= 1024 partial [K, N] blocks. */
size_t pgt_gemm_tn_det_ws_bytes(int64_t n_seg, int64_t seg_k, int64_t N, int64_t lddw);
int pgt_gemm_tn_det_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                        const float* G, int64_t ldg, float* dW, int64_t lddw, float* db, int64_t M, int64_t N,
                        void* ws, size_t ws_bytes, pgt_stream_t stream);

/* The DCRNN gate GEMMs with the gate chain fused into the epilogue (operands A / Bw / bias as pgt_gemm_f32):
 *   pgt_gemm_gru_zr_f32:  zr [M,2O] = sigmoid(A Bw + bias);  xhr[m, f_in + o] = H[m,o] * zr[m, O + o]
 *                         == pgt_gemm_f32 into zr followed by pgt_gru_zr_f32 (dcrnn.py:172-185) (bit for bit on the
 *                         fmaf-chain kernels), without writing and re-reading the pre-activations.
 *   pgt_gemm_gru_h_f32:   ht [M,O] = tanh(A Bw + bias);  Hnew = Z*H + (1-Z)*ht with Z = zr[m, o] (row stride 2O)
 *                         written to out0 and, when non-NULL, out1 == pgt_gemm_f32 + pgt_gru_h_f32 (dcrnn.py:186-192).
 *                         map0 (NULL = plain rows): out0 in a two-level row layout (pgt_rowmap) — the cell of time step t
 *                         writes H_t straight into the reference's [B, T, N, O] result (dcrnn.py:463-475).
 * O must be a multiple of 4, zr / ht 16-byte aligned.  Arithmetic as pgt_gemm_f32; on the split-bf16 kernel the gate chain
 * uses the hardware exp / reciprocal (sigmoid within 3e-7 of the library one, tanh by its odd series below |x| = 0.04), so
 * the fused and the unfused path agree to a few 1e-7, not bit for bit, from 8 192 rows. */
int pgt_gemm_gru_zr_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                        const float* Bw, int64_t sbk, int64_t sbn, const float* bias, float* zr, const float* H,
                        int64_t ldh, float* xhr, int64_t ldxhr, int64_t f_in, int64_t M, int64_t O,
                        pgt_stream_t stream);
int pgt_gemm_gru_h_f32(const float* A, int64_t lda, int64_t a_seg_stride, int64_t n_seg, int64_t seg_k,
                       const float* Bw, int64_t sbk, int64_t sbn, const float* bias, float* ht, const float* zr,
                       const float* H, int64_t ldh, float* out0, int64_t ld0, const pgt_rowmap* map0, float* out1,
                       int64_t ld1, int64_t M, int64_t O, pgt_stream_t stream);

/* ---------------------------------------------------------------- DCRNN cell plumbing (one launch each) */

/* The three DConv weights of a DCRNN cell, Wz / Wr / Wh [2, K, C, O] (conv_x_{z,r,h}.weight, dcrnn.py:26-37, :138-160),
 * packed into the operands of the two gate products: Wzr [(2K-1) C, 2O] (update | reset), Whs [(2K-1) C, O] (candidate);
 * segment 0 = W[0,0] + W[1,0] (the reference multiplies X by both, dcrnn.py:81-83), segment 2k-1+d = W[d,k]; bzr [2O] =
 * bz | br (all three NULL for bias-free cells).  The adjoint scatters the packed gradients back into the parameters'
 * shapes (NULL dWzr / dWhs count as zero).  One launch each way instead of the ~8 / ~10 torch slice / add / cat launches per
 * call — the bulk of a per-snapshot step on a 20-node graph. */
int pgt_dcrnn_pack_weights_f32(const float* Wz, const float* Wr, const float* Wh, const float* bz, const float* br, int64_t K,
                               int64_t C, int64_t O, float* Wzr, float* bzr, float* Whs, pgt_stream_t stream);
int pgt_dcrnn_unpack_weight_grads_f32(const float* dWzr, const float* dbzr, const float* dWhs, int64_t K, int64_t C, int64_t O,
                                      float* dWz, float* dWr, float* dWh, float* dbz, float* dbr, pgt_stream_t stream);
/* Inputs of a DCRNN sequence into its diffusion stacks: X [T * M, Fin] into the input columns of segment 0 of both stacks
 * (TSzr0 / TSh0: [T * M, Fin + O], dcrnn.py:173,185 `torch.cat([X, H])`) and the initial state H0 [M, O] into step 0 of the
 * gate stack — one launch for three strided copies. */
int pgt_dcrnn_stage_f32(const float* X, const float* H0, int64_t T, int64_t M, int64_t Fin, int64_t O, float* TSzr0,
                        float* TSh0, pgt_stream_t stream);

/* ---------------------------------------------------------------- DCRNN cell without diffusion (K = 1), one launch */

/* The whole cell of DCRNN(in, out, K = 1) (dcrnn.py:79-82: DConv is `X @ W[0,0] + X @ W[1,0] + b`, the hop loop never
 * runs; gates dcrnn.py:172-192) — BASELINE configs[0], examples/recurrent/dcrnn_example.py:19-28.  X [N, in] (row stride
 * ldx), H [N, out] or NULL (= zeros, dcrnn.py:167-170), Wz / Wr / Wh in the parameters' own layout [2, 1, in + out, out],
 * bz / br / bh [out] or NULL.  Writes H' [N, out] (row stride ldo) and saved [N, 3 out] = Z | R | candidate for the
 * adjoint.  Limits (pgt_dcrnn_cell_k1_fits): 1 <= N <= 4096, out <= 64, in + out <= 128; PGT_ERR_INVALID beyond them
 * (larger cells run through the general path).
 * Adjoint (one workgroup): G = d/dH' [N, out]; dX [N, in] and dH [N, out] are written when non-NULL; dWz / dWr / dWh
 * [2, 1, in + out, out] are WRITTEN (both halves receive the same gradient), dbz / dbr / dbh [out] written when non-NULL;
 * dP is caller scratch of N * 3 out floats.  Sums run in index order: deterministic. */
int pgt_dcrnn_cell_k1_fits(int64_t N, int64_t Fin, int64_t O);
int pgt_dcrnn_cell_k1_f32(const float* X, int64_t ldx, const float* H, int64_t ldh, const float* Wz, const float* Wr,
                          const float* Wh, const float* bz, const float* br, const float* bh, float* Hnew, int64_t ldo,
                          float* saved, int64_t N, int64_t Fin, int64_t O, pgt_stream_t stream);
int pgt_dcrnn_cell_k1_bwd_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, const float* H, int64_t ldh,
                              const float* Wz, const float* Wr, const float* Wh, const float* saved, float* dX, int64_t lddx,
                              float* dH, int64_t lddh, float* dWz, float* dWr, float* dWh, float* dbz, float* dbr, float* dbh,
                              float* dP, int64_t N, int64_t Fin, int64_t O, pgt_stream_t stream);

/* ---------------------------------------------------------------- one GCN layer on a small graph, one launch */

/* out = A_hat (X W) straight from the edge list (GCNConv_Fixed_W.forward, evolvegcno.py:76-101; gcn_norm as PyG 2.5 / 2.6
 * has it: add_remaining_self_loops — self-loop edges leave the list, node i's loop weighs what its LAST self-loop edge
 * weighed, else 1 (2 for `improved`; always 1 when edge_weight is NULL) —, deg = scatter_add(w, col), coefficient
 * (deg^-1/2[row] w) deg^-1/2[col] with inf -> 0; normalize = 0: the edge list as it is, no loops) for graphs that fit one
 * workgroup's LDS (pgt_gcn_small_fits: N <= 512, E <= 4096, widths <= 64, N * Fo <= 8192) — BASELINE configs[4], a new
 * edge list per snapshot: no sorted operator is built, no workspace, no host check.  edge_index [2, E] int64 (row 0 =
 * sources), edge_weight [E] or NULL, X [N, Fi] (row stride ldx), W [Fi, Fo].  Writes out [N, Fo] (contiguous) and coef
 * [E + N] (the edges' coefficients, 0 for removed / out-of-range edges, then the N loop coefficients) for the adjoint;
 * info[0] is incremented per edge with an endpoint outside [0, N) (such edges are skipped; the caller keeps info zeroed).
 * Every sum runs in edge order (destination lists by counting + a fill in edge order; products rounded before they are
 * added, as message + index_add_ do): deterministic, no float atomics.
 * Adjoint: G = d/d out [N, Fo] (row stride ldg); writes dW [Fi, Fo] = X^T (A_hat^T G) and, when dX != NULL, dX [N, Fi]
 * (row stride lddx) = (A_hat^T G) W^T.  Edge weights receive no gradient (as everywhere in this library). */
int pgt_gcn_small_fits(int64_t N, int64_t E, int64_t Fi, int64_t Fo);
int pgt_gcn_small_f32(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N, int improved,
                      int add_self_loops, int normalize, const float* X, int64_t ldx, const float* W, int64_t Fi, int64_t Fo,
                      float* out, float* coef, int32_t* info, pgt_stream_t stream);
int pgt_gcn_small_bwd_f32(const int64_t* edge_index, const float* coef, int64_t E, int64_t N, int add_self_loops, int normalize,
                          const float* G, int64_t ldg, const float* X, int64_t ldx, const float* W, int64_t Fi, int64_t Fo,
                          float* dW, float* dX, int64_t lddx, pgt_stream_t stream);

/* ---------------------------------------------------------------- EvolveGCN weight evolution (one launch) */

/* W_t from W_{t-1} for one snapshot (evolvegcnh.py:78-102: TopKPooling summary of X_t -> torch.nn.GRU -> weight;
 * evolvegcno.py:170-191 with pool = 0: the weight is the GRU's input and hidden state):
 *   pool != 0:  s_i = tanh(X_i . p / |p|), perm = the k rows with the largest s (ties: lowest index; nan ranks first, as
 *               torch.sort(descending) has it), xt_j = X[perm_j] * s[perm_j];   pool == 0:  xt = Wprev
 *   GRU cell with torch.nn.GRU's parameters (weight_ih_l0 / weight_hh_l0 [3F, F], gates r | z | n; biases [3F] or both NULL)
 *   on the k rows of Wprev [k, F] as its batch -> Wnew [k, F].
 * perm [k] (int32), score [k], gates [4 k F] (r, z, n, W_hn h + b_hn), xt [k F] are kept for the adjoint, which returns
 * every gradient in one launch as well: dWih / dWhh [3F, F], dbih / dbhh [3F], dWprev [k, F], and for pool != 0 dp [F] and
 * the k selected rows of dX [N, F] (the caller zero-fills dX; the selection carries no gradient).  F, k <= 64, N <= 4096.
 * One workgroup each: ~25 (forward) / ~40 (backward) torch and MIOpen launches per snapshot become one. */
int pgt_evolve_weight_f32(const float* X, int64_t ldx, int64_t N, const float* p, const float* Wih, const float* Whh,
                          const float* bih, const float* bhh, const float* Wprev, int64_t F, int64_t k, int pool, float* Wnew,
                          int32_t* perm, float* score, float* gates, float* xt, pgt_stream_t stream);
int pgt_evolve_weight_bwd_f32(const float* dWnew, const float* X, int64_t ldx, int64_t N, const float* p, const float* Wih,
                              const float* Whh, const float* Wprev, const int32_t* perm, const float* score, const float* gates,
                              const float* xt, int64_t F, int64_t k, int pool, float* dX, int64_t lddx, float* dp, float* dWih,
                              float* dWhh, float* dbih, float* dbhh, float* dWprev, pgt_stream_t stream);

/* ---------------------------------------------------------------- GRU gate chains */

/* DCRNN (dcrnn.py:172-192):  pre_zr [M,2*O] holds the two DConv outputs (bias included).
 *   zr = sigmoid(pre_zr) in place;  xhr[m, f_in + o] = H[m,o] * R[m,o]   (candidate input, dcrnn.py:185) */
int pgt_gru_zr_f32(float* pre_zr, const float* H, int64_t ldh, float* xhr, int64_t ldxhr, int64_t f_in,
                   int64_t M, int64_t O, pgt_stream_t stream);
/*   ht = tanh(pre_h) in place;  Hnew = Z*H + (1-Z)*ht  (dcrnn.py:188-192), written to out0 (row stride ld0)
 *   and, when out1 != NULL, also to out1 (row stride ld1). */
int pgt_gru_h_f32(float* pre_h, const float* zr, const float* H, int64_t ldh, float* out0, int64_t ld0,
                  const pgt_rowmap* map0, float* out1, int64_t ld1, int64_t M, int64_t O, pgt_stream_t stream);
/* backward of pgt_gru_h_f32: given dHnew (+ dHnew2 + dHnew3 when non-NULL: the output gradient, the running state
 *   gradient and the later step's gate-stack gradient of H are summed on the fly — no separate accumulation pass),
 *   writes d_pre_h [M,O], d_pre_zr[:, 0:O] (update gate), and
 *   dH (=|+=) dHnew * Z  depending on accumulate_dh.  dH may alias dHnew2.  map_dh / map_h (NULL = plain rows): dHnew / H
 *   in a two-level row layout (pgt_rowmap): the incoming gradient and the previous state are read in place from
 *   [B, T, N, O] tensors. */
int pgt_gru_h_bwd_f32(const float* dHnew, int64_t lddh, const pgt_rowmap* map_dh, const float* dHnew2, int64_t lddh2,
                      const float* dHnew3, int64_t lddh3, const float* zr, const float* H, int64_t ldh,
                      const pgt_rowmap* map_h, const float* ht,
                      float* d_pre_h, float* d_pre_zr, float* dH, int64_t lddhp, int accumulate_dh, int64_t M, int64_t O,
                      pgt_stream_t stream);
/* backward of pgt_gru_zr_f32: dxhr[:, f_in:] is d(H*R);  d_pre_zr[:, O:2O] = dHR*H*R*(1-R);  dH += dHR*R */
int pgt_gru_zr_bwd_f32(const float* dxhr, int64_t lddxhr, int64_t f_in, const float* zr, const float* H,
                       int64_t ldh, const pgt_rowmap* map_h, float* d_pre_zr, float* dH, int64_t lddhp, int64_t M,
                       int64_t O, pgt_stream_t stream);

/* TGCN (temporalgcn.py:82-102): pre_zr [M,2*O] = linear_{z,r}([conv(X), H]);
 *   zr = sigmoid(pre_zr) in place; hr[m,o] = H[m,o]*R[m,o] */
/* (uses pgt_gru_zr_f32 with f_in = 0) */

/* ---------------------------------------------------------------- LSTM gate chains */

/* Peephole LSTM of GConvLSTM (gconv_lstm.py:138-172) and, with wci = wcf = wco = NULL, the plain LSTM of GCLSTM
 * (gc_lstm.py:138-169).  P [M, 4*O] holds the gate pre-activations i | f | c | o (all biases already added by the
 * GEMM that produced it) and is overwritten with the activated gates (kept for the backward pass):
 *   I = s(P_i + wci*C)  F = s(P_f + wcf*C)  T = tanh(P_c)  C' = F*C + I*T  O = s(P_o + wco*C')  H = O * tanh(C') */
int pgt_lstm_gates_f32(float* P, const float* C, int64_t ldc, const float* wci, const float* wcf, const float* wco,
                       float* Hn, int64_t ldh, float* Cn, int64_t ldcn, int64_t M, int64_t O, pgt_stream_t stream);
/* backward: gates = the activated P of the forward call; dCn may be NULL (no gradient into the new cell state).
 * Writes dP [M,4*O] (pre-activation gradients), dC [M,O] (previous cell) and ACCUMULATES the peephole gradients into
 * dw [3*O] (wci | wcf | wco rows; fp32 atomics; required when any peephole weight is given). */
int pgt_lstm_gates_bwd_f32(const float* gates, const float* C, int64_t ldc, const float* Cn, int64_t ldcn,
                           const float* wci, const float* wcf, const float* wco, const float* dH, int64_t lddh,
                           const float* dCn, int64_t lddcn, float* dP, float* dC, int64_t lddc, float* dw, int64_t M,
                           int64_t O, pgt_stream_t stream);

/* ---------------------------------------------------------------- small data movers on the path */

/* dst[m, 0:W] = src[m, 0:W] for m < M (row strides ldd/lds) — concatenation [X, H] (dcrnn.py:173,179,185). */
int pgt_copy2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t M, int64_t W,
                   pgt_stream_t stream);
/* dst[m, 0:W] += src[m, 0:W] */
int pgt_add2d_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t M, int64_t W,
                  pgt_stream_t stream);
/* dst[m, 0:W] = a*x[m,0:W] + b*y[m,0:W]  (y may be NULL) */
int pgt_axpby2d_f32(float* dst, int64_t ldd, const float* x, int64_t ldx, float a, const float* y, int64_t ldy,
                    float b, int64_t M, int64_t W, pgt_stream_t stream);
/* [D0][D1][W] -> [D1][D0][W] blocked transpose of W-float records (batch-major <-> node-major). */
int pgt_swap01_f32(float* dst, const float* src, int64_t D0, int64_t D1, int64_t W, pgt_stream_t stream);

/* ---------------------------------------------------------------- dense attention scores (ASTGCN)
 * S[b] = softmax_dim1( V . sigmoid( L[b] R[b] + bias ) )  (SpatialAttention astgcn.py:226-262 with n = nodes, m = time steps;
 * TemporalAttention :291-328 with n = time steps, m = nodes).  The intermediate layout is [i][b][j] (score-matrix row
 * outermost) so that V . sigma_b for the whole batch is ONE pgt_gemm_f32 call [n, n] x [n, B n] and every stage is
 * coalesced along j.  L [B, n, m], R [B, m, n], bias [n, n], row-major.
 *   pgt_att_sigmoid_scores_f32   sig[i][b][j] = sigmoid(sum_t L[b,i,t] R[b,t,j] + bias[i,j])    (L R is never stored)
 *   pgt_att_softmax_rows_f32     S[b][i][j] = softmax over i of C[i][b][j]                        (F.softmax(., dim=1))
 *   pgt_att_softmax_rows_bwd_f32 dC[i][b][j] = S (dS - sum_i dS S)
 *   pgt_att_sigmoid_bwd_f32      dP[b][i][j] = dsig[i][b][j] sig (1 - sig); dbias[i][j] = sum_b dP   (dbias may be NULL) */
int pgt_att_sigmoid_scores_f32(const float* L, const float* R, const float* bias, int64_t B, int64_t n, int64_t m,
                               float* sig, pgt_stream_t stream);
int pgt_att_softmax_rows_f32(const float* C, int64_t B, int64_t n, float* S, pgt_stream_t stream);
int pgt_att_softmax_rows_bwd_f32(const float* S, const float* dS, int64_t B, int64_t n, float* dC, pgt_stream_t stream);
int pgt_att_sigmoid_bwd_f32(const float* sig, const float* dsig, int64_t B, int64_t n, float* dP, float* dbias,
                            pgt_stream_t stream);

/* Strided batched product of small matrices: C[b, i, j] (+)= sum_k A[b, i, k] B[b, k, j] for b < nb, every operand given by
 * a base pointer and three strides in floats (a batch stride of 0 shares one matrix across the batch; swapping two
 * strides transposes).  The embeddings around ASTGCN's attention matrices and their adjoints (astgcn.py:252-256, :318-322,
 * :437: products with 1 .. 64 rows or columns), instead of one library batched GEMM each.  fmaf chain in k order,
 * deterministic — except a tall contraction into fewer than 512 tiles (K >= 4096: the adjoint of a batch-shared embedding),
 * which is cut along K over the chip and summed into C with fp32 atomics.  Up to 2^31 row tiles and batches (PEMS07 at
 * B = 32: M = B N F = 1.8 M rows), N <= 1 048 560. */
int pgt_bmm_f32(const float* A, int64_t sab, int64_t sai, int64_t sak, const float* B, int64_t sbb, int64_t sbk, int64_t sbj,
                float* C, int64_t scb, int64_t sci, int64_t scj, int64_t nb, int64_t M, int64_t N, int64_t K, int accumulate,
                pgt_stream_t stream);

/* Y[r, 0:C] = LayerNorm(relu(Z[row(r), 0:C])) with gamma / beta [C] (torch.nn.LayerNorm: biased variance, eps inside the
 * root) — the tail of an ASTGCN block, `self._layer_norm(F.relu(X + X_hat))` (astgcn.py:476-478), on the buffer the two
 * convolutions were summed into.  row(r) = (r / row_period) * stride_hi + (r % row_period) * stride_lo (in rows of C
 * floats): a strided time convolution's outputs are picked and its padding rows skipped.  stats [2 * rows] receives
 * (mean, 1 / std) for the adjoint.  C <= 1024.
 * Adjoint: dZ[row(r)] from dY[r] (rows of dZ the map does not reach are not written); dgamma / dbeta [C] are ACCUMULATED
 * into (fp32 atomics: one per column and workgroup, at most 2048 workgroups — the sums over a workgroup's rows are taken
 * in registers and LDS first). */
int pgt_relu_layernorm_f32(const float* Z, int64_t row_period, int64_t stride_hi, int64_t stride_lo, const float* gamma,
                           const float* beta, float eps, int64_t rows, int64_t C, float* Y, float* stats,
                           pgt_stream_t stream);
int pgt_relu_layernorm_bwd_f32(const float* Z, int64_t row_period, int64_t stride_hi, int64_t stride_lo, const float* gamma,
                               const float* stats, const float* dY, int64_t rows, int64_t C, float* dZ, float* dgamma,
                               float* dbeta, pgt_stream_t stream);

/* Gated temporal convolution of an ST-Conv block — TemporalConv.forward (nn/attention/stgcn.py:27-44): three
 * Conv2d(Cin -> Cout, (1, k)) over the time axis and H = relu(P * sigmoid(Q) + R), on the reference's own layout:
 * X [B, T, N, Cin] (rows of ldx floats) -> H [B, T - k + 1, N, Cout] contiguous, no permutes, no im2col: tap dt is the same
 * matrix dt * N rows further down, the three convolutions are ONE product [rows, k Cin] x [k Cin, 3 Cout] on the matrix
 * cores and the gate is computed on the accumulators.  Wp [k * Cin, 3 * Cout]: Wp[dt * Cin + ci, g * Cout + c] =
 * conv_{g+1}.weight[c, ci, 0, dt]; bias3 [3 * Cout] (conv_1 | conv_2 | conv_3) or NULL.  P, S [B, T', N, Cout] (both or
 * neither) receive conv_1's output and sigmoid(conv_2's) for the adjoint.
 * Adjoint of the gate: dZ [(k - 1) N + B T N, 3 Cout] = (dP | dQ | dR) in INPUT row numbering behind (k - 1) N zero rows,
 * zero where a step has no output — the operand of the two gradient products: weight gradient = pgt_gemm_tn_acc_f32 of X
 * as k segments N rows apart against dZ + (k - 1) N rows over (B T - k + 1) N rows; input gradient = pgt_gemm_f32 of dZ as k
 * segments N rows apart against the taps in reverse order. */
int pgt_tconv_glu_f32(const float* X, int64_t ldx, int64_t B, int64_t T, int64_t N, int64_t Cin, int64_t Cout, int64_t k,
                      const float* Wp, const float* bias3, float* H, float* P, float* S, pgt_stream_t stream);
int pgt_tconv_glu_bwd_f32(const float* dH, const float* H, const float* P, const float* S, int64_t B, int64_t T, int64_t N,
                          int64_t Cout, int64_t k, float* dZ, pgt_stream_t stream);

/* BatchNorm2d(num_nodes) of STConv (nn/attention/stgcn.py:129, :156-159; the reference permutes [B, T', N, C] to
 * [B, N, T', C] so that the NODE is the normalised channel) in place on X [R = B T', N, C] contiguous: per node n, mean and
 * biased variance over the R * C values (training) or the running statistics (evaluation), Y = (X - mean) / sqrt(var + eps)
 * * gamma[n] + beta[n] (gamma / beta NULL = 1 / 0).  Training also updates running_mean / running_var [N] (either may be
 * NULL) with `momentum` (torch semantics: unbiased variance) in the same launch.  stats [2 N] receives (mean, 1 / std) for
 * the adjoint (may be NULL in evaluation).  One workgroup per node, sums in a fixed order (deterministic).
 * Adjoint: dX (may be NULL), dgamma / dbeta [N] (STORED, may be NULL); training = 0 treats the statistics as constants. */
int pgt_batchnorm_nodes_f32(const float* X, int64_t R, int64_t N, int64_t C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int training, float* Y,
                            float* stats, pgt_stream_t stream);
int pgt_batchnorm_nodes_bwd_f32(const float* dY, const float* X, const float* stats, const float* gamma, int64_t R, int64_t N,
                                int64_t C, int training, float* dX, float* dgamma, float* dbeta, pgt_stream_t stream);

/* T-GCN cell parameters (nn/recurrent/temporalgcn.py:38-70, :82-102): a gate is linear_g(cat[conv_g(X), H']) with conv_g a
 * PyG GCNConv; both maps are linear, so with AX = A_hat X (one aggregation at the input width for the three gates)
 *   pre_g = [AX | H'] W'_g + b'_g,  W'_g = [Wc_g^T L_g[:, :O]^T ; L_g[:, O:]^T] (Fin + O rows),  b'_g = lb_g + L_g[:, :O] bc_g
 * and the cell runs on pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32.  pack: Wc[g] = conv_g.lin.weight [O, Fin], bc[g] = conv_g.bias
 * [O] | NULL, L[g] = linear_g.weight [O, 2 O], lb[g] = linear_g.bias [O] | NULL for g = z, r, h (host arrays of three device
 * pointers) -> Wzr [Fin + O, 2 O], bzr [2 O], Wh [Fin + O, O], bh [O].  unpack: the adjoint — dWc[g], dL[g] (and dbc[g], dlb[g]
 * where non-NULL) STORED from (dWzr, dbzr, dWh, dbh); one launch each, sums in index order. */
int pgt_tgcn_pack_weights_f32(const float* const Wc[3], const float* const bc[3], const float* const L[3],
                              const float* const lb[3], int64_t Fin, int64_t O, float* Wzr, float* bzr, float* Wh, float* bh,
                              pgt_stream_t stream);
int pgt_tgcn_unpack_weight_grads_f32(const float* dWzr, const float* dbzr, const float* dWh, const float* dbh,
                                     const float* const Wc[3], const float* const bc[3], const float* const L[3], int64_t Fin,
                                     int64_t O, float* const dWc[3], float* const dbc[3], float* const dL[3],
                                     float* const dlb[3], pgt_stream_t stream);

/* Whole DCRNN sequences of small graphs, one workgroup per sample (csrc/seq_small.hip): BatchedDCRNN.forward (nn/recurrent/
 * dcrnn.py:429-475; cell :194-219, gates :172-192, diffusion :85-106) for graphs and widths whose per-sample state fits a
 * workgroup's LDS (pgt_dcrnn_seq_small_fits: both operators, the [2K - 1][N][Fin + O] stack and the gate buffers within 150 KB:
 * METR-LA / PeMS-BAY at hidden <= 8, Chickenpox at hidden 32) — the reference's own configuration BatchedDCRNN(2, 2, K = 3) on 64
 * windows is ~220 launches per training step on the general path and ONE launch each way here.
 *   X[b, t] = X + b * x_stride_b + t * x_stride_t: [N, Fin] rows; out likewise [N, O] rows; H0 [B, N, O] or NULL (zeros);
 *   Wzr [(2K - 1)(Fin + O), 2 O], bzr [2 O] | NULL, Wh [(2K - 1)(Fin + O), O], bh [O] | NULL (pgt_dcrnn_pack_weights_f32's
 *   stacked operands); save [B][T][pgt_dcrnn_seq_small_save_floats] or NULL (inference): both stacks, Z | R and the candidate of
 *   every step for the adjoint.
 * Adjoint (hand-written BPTT): tp_o / tp_i are the TRANSPOSED operators; dX (may be NULL) in X's layout, dH0 [B, N, O] (may be
 * NULL), dWpart [B][(2K - 1)(Fin + O) 3 O + 3 O] = per-sample (dWzr | dWh | dbzr | dbh), ZEROED by the caller and summed over
 * the samples by it (index order: deterministic, no atomics). */
int pgt_dcrnn_seq_small_fits(int64_t N, int64_t E_o, int64_t E_i, int64_t Fin, int64_t O, int64_t K);
int64_t pgt_dcrnn_seq_small_save_floats(int64_t N, int64_t Fin, int64_t O, int64_t K);
int pgt_dcrnn_seq_small_f32(const pgt_csr* op_o, const pgt_csr* op_i, int64_t E_o, int64_t E_i, int64_t N, const float* X,
                            int64_t x_stride_b, int64_t x_stride_t, const float* H0, const float* Wzr, const float* bzr,
                            const float* Wh, const float* bh, int64_t B, int64_t T, int64_t Fin, int64_t O, int64_t K, float* out,
                            int64_t out_stride_b, int64_t out_stride_t, float* save, pgt_stream_t stream);
int pgt_dcrnn_seq_small_bwd_f32(const pgt_csr* tp_o, const pgt_csr* tp_i, int64_t E_o, int64_t E_i, int64_t N, const float* dOut,
                                int64_t g_stride_b, int64_t g_stride_t, const float* out, int64_t out_stride_b,
                                int64_t out_stride_t, const float* H0, const float* save, const float* Wzr, const float* Wh,
                                int64_t B, int64_t T, int64_t Fin, int64_t O, int64_t K, float* dX, int64_t x_stride_b,
                                int64_t x_stride_t, float* dH0, float* dWpart, pgt_stream_t stream);

/* Whole DCRNN sequences at hidden width 64, one workgroup per sample (csrc/seq64.hip): BatchedDCRNN.forward (nn/recurrent/dcrnn.py:
 * 429-475; cell :194-219, gates :172-192, diffusion :85-106) of the benchmarked model BatchedDCRNN(2, 64, K) on graphs whose block
 * and operators fit a CU's LDS (pgt_dcrnn_seq64_fits: O == 64, Fin == 2, K = 2 | 3, N <= 208, two [N, 68] blocks + a 24 KB weight
 * ring + 6 bytes per slot within 160 KB: METR-LA at 1 515 and at 1 722 edges).  ONE launch runs all T steps of every sample: the
 * hops are gathered out of LDS (the diffusion terms are bit-identical to pgt_dconv_stack_slab_f32's), every term meets its weight
 * block while it is in LDS (split-bf16 products on v_mfma_f32_16x16x32_bf16, fp32 accumulation: the arithmetic of
 * pgt_gemm_gru_zr_f32 / pgt_gemm_gru_h_f32 above 8 192 rows), the gate chains run on the accumulators.
 *   Wp: pgt_dcrnn_seq64_pack_floats(K) floats, 16-byte aligned, written by pgt_dcrnn_seq64_pack_f32 from the stacked operands
 *   Wzr [(2K - 1)(Fin + 64), 128], Wh [(2K - 1)(Fin + 64), 64] (pgt_dcrnn_pack_weights_f32) whenever they change;
 *   X[b, t] = X + b * x_stride_b + t * x_stride_t: [N, Fin] rows; out likewise [N, 64] rows; H0 [B, N, 64] or NULL (zeros);
 *   saved for the adjoint (all required; batch-major rows m = b N + n, M = B N): TSzr / TSh = the two diffusion stacks, segment s
 *   of step t at s * seg_stride + t * t_stride + m * (Fin + 64) ([T0 | T1o T1i | T2o T2i], 8-byte aligned, even strides),
 *   ZR [T, M, 128] = Z | R, HT [T, M, 64] = the candidate — the layout pgt_gru_*_bwd_f32 / pgt_dconv_stack_slab_bwd_f32 /
 *   pgt_gemm_tn_acc_f32 take. */
int pgt_dcrnn_seq64_fits(int64_t N, int64_t E_o, int64_t E_i, int64_t Fin, int64_t O, int64_t K);
int64_t pgt_dcrnn_seq64_pack_floats(int64_t K);
int pgt_dcrnn_seq64_pack_f32(const float* Wzr, const float* Wh, int64_t Fin, int64_t K, float* Wp, pgt_stream_t stream);
int pgt_dcrnn_seq64_f32(const pgt_csr* op_o, const pgt_csr* op_i, int64_t E_o, int64_t E_i, int64_t N, const float* X,
                        int64_t x_stride_b, int64_t x_stride_t, const float* H0, const float* Wp, const float* Wzr,
                        const float* bzr, const float* Wh, const float* bh, int64_t B, int64_t T, int64_t Fin, int64_t K,
                        float* out, int64_t out_stride_b, int64_t out_stride_t, float* TSzr, float* TSh,
                        int64_t seg_stride, int64_t t_stride, float* ZR, float* HT, pgt_stream_t stream);

/* The adjoint of the same sequences in one launch (hand-written BPTT; replaces per cell step pgt_gru_h_bwd_f32, two feature-gradient
 * pgt_gemm_f32, two pgt_dconv_stack_slab_bwd_f32 and pgt_gru_zr_bwd_f32), the input being data (no d/dX):
 *   tp_o / tp_i: the TRANSPOSED operators; dOut[b, t] = dOut + b * g_stride_b + t * g_stride_t: [N, 64] rows (the gradient of
 *   out); out / ZR / HT: what pgt_dcrnn_seq64_f32 wrote; Wp: pgt_dcrnn_seq64_pack_floats(K) floats written by
 *   pgt_dcrnn_seq64_pack_bwd_f32 (the weights transposed, segment 0 folded as ops.fold_backward_weight does);
 *   -> dPzr [T, M, 128], dPh [T, M, 64]: the gradients of the gates' pre-activations, STORED — the weight gradients are
 *   pgt_gemm_tn_acc_f32(saved stack, dP) over all T M rows; dH0 [B, N, 64] or NULL; ws: pgt_dcrnn_seq64_bwd_ws_floats(N, B) floats
 *   of scratch.  All operands 16-byte addressable. */
int pgt_dcrnn_seq64_pack_bwd_f32(const float* Wzr, const float* Wh, int64_t Fin, int64_t K, float* Wp, pgt_stream_t stream);
int64_t pgt_dcrnn_seq64_bwd_ws_floats(int64_t N, int64_t B);
int pgt_dcrnn_seq64_bwd_f32(const pgt_csr* tp_o, const pgt_csr* tp_i, int64_t E_o, int64_t E_i, int64_t N, const float* dOut,
                            int64_t g_stride_b, int64_t g_stride_t, const float* out, int64_t out_stride_b, int64_t out_stride_t,
                            const float* H0, const float* ZR, const float* HT, const float* Wp, int64_t B, int64_t T, int64_t Fin,
                            int64_t K, float* dPzr, float* dPh, float* dH0, float* ws, int64_t ws_floats, pgt_stream_t stream);

/* Fused T-GCN cell for hidden width 32 (nn/recurrent/temporalgcn.py:82-130; csrc/tgcn_cell.hip): with AX = A_hat X [M, Fin] and
 * the folded operands of pgt_tgcn_pack_weights_f32,
 *   Z | R = sigmoid([AX | H] Wzr + bzr),  H' = Z H + (1 - Z) tanh([AX | H * R] Wh + bh)
 * in ONE launch: ZR [M, 64] and HT [M, 32] (the candidate) are written for the adjoint, Hn [M, 32] (row stride ldhn) is the new
 * state.  pgt_tgcn_cell_fits: O == 32 and 1 <= Fin <= 30.
 * Adjoint in one launch + a reduction: dH [M, 32] (STORED), dWzr [Fin + 32, 64], dbzr [64] | NULL, dWh [Fin + 32, 32], dbh [32] |
 * NULL (STORED: per-workgroup partial sums in ws — pgt_tgcn_cell_bwd_ws_floats floats — added in index order, deterministic).
 * d/dAX is not produced: a caller that needs the input gradient runs the unfused adjoint (pgt_gru_*_bwd_f32 + pgt_gemm_f32). */
int pgt_tgcn_cell_fits(int64_t Fin, int64_t O);
int64_t pgt_tgcn_cell_bwd_ws_floats(int64_t Fin, int64_t O);
int pgt_tgcn_cell_f32(const float* AX, int64_t ldax, const float* H, int64_t ldh, const float* Wzr, const float* bzr,
                      const float* Wh, const float* bh, int64_t M, int64_t Fin, int64_t O, float* ZR, float* HT, float* Hn,
                      int64_t ldhn, pgt_stream_t stream);
int pgt_tgcn_cell_bwd_f32(const float* dHn, int64_t lddhn, const float* AX, int64_t ldax, const float* H, int64_t ldh,
                          const float* ZR, const float* HT, const float* Wzr, const float* Wh, int64_t M, int64_t Fin, int64_t O,
                          float* dH, int64_t lddh, float* dWzr, float* dbzr, float* dWh, float* dbh, float* ws, int64_t ws_floats,
                          pgt_stream_t stream);
/* The same adjoint with the four weight / bias gradients ADDED to what dWzr / dbzr / dWh / dbh hold (dH is stored): the cells of a
 * T-step loop (examples/indexBatching/tgcn/metr_la_main.py:41-45) share their folded operands and sum their gradients in one buffer. */
int pgt_tgcn_cell_bwd_acc_f32(const float* dHn, int64_t lddhn, const float* AX, int64_t ldax, const float* H, int64_t ldh,
                              const float* ZR, const float* HT, const float* Wzr, const float* Wh, int64_t M, int64_t Fin, int64_t O,
                              float* dH, int64_t lddh, float* dWzr, float* dbzr, float* dWh, float* dbh, float* ws, int64_t ws_floats,
                              pgt_stream_t stream);

/* The read-out behind a recurrent layer (examples/indexBatching/tgcn/metr_la_main.py:43-45: `self.linear(F.relu(h))`, torch.nn.Linear
 * with 1 .. 4 outputs) as one streaming pass each way over the states X [M, K] (row stride ldx, 16-byte addressable), csrc/readout.hip:
 *   forward   Y[m, n] = sum_k act(X[m, k]) W[n, k] + b[n],  act = relu when `relu` != 0, identity otherwise;  W [N, K] row-major
 *             (torch.nn.Linear.weight), b [N] | NULL, Y [M, N] row stride ldy
 *   adjoint   dX[m, k] = act'(X[m, k]) sum_n dY[m, n] W[n, k]  (dX NULL: not wanted),  dW [N, K] = dY^T act(X),  db [N] = column sums
 *             of dY (either NULL: not wanted) — STORED, deterministic: per-workgroup partial sums in ws
 *             (pgt_relu_linear_bwd_ws_floats floats) added in index order.
 * pgt_relu_linear_fits: 4 <= K <= 64, K % 4 == 0, 1 <= N <= 4. */
int pgt_relu_linear_fits(int64_t K, int64_t N);
int64_t pgt_relu_linear_bwd_ws_floats(int64_t K, int64_t N);
int pgt_relu_linear_f32(const float* X, int64_t ldx, const float* W, const float* b, int64_t M, int64_t K, int64_t N, int relu,
                        float* Y, int64_t ldy, pgt_stream_t stream);
int pgt_relu_linear_bwd_f32(const float* X, int64_t ldx, const float* dY, int64_t lddy, const float* W, int64_t M, int64_t K,
                            int64_t N, int relu, float* dX, int64_t lddx, float* dW, float* db, float* ws, int64_t ws_floats,
                            pgt_stream_t stream);

/* torch.optim.Adam's update (no amsgrad; weight_decay = L2 added to the gradient) over one flat buffer of n parameters — the
 * optimizer step of the reference's training loops (examples/indexBatching/DCRNN/pems_ddp.py:86,118-120) on the concatenation of all
 * parameters (every tensor of these models is small: 150 - 76 000 floats in total).  p / m / v updated in place, g read; `step`:
 * one device float, the number of updates so far — advanced by this call, so the whole update is graph-capturable. */
int pgt_adam_f32(float* p, const float* g, float* m, float* v, float* step, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, pgt_stream_t stream);

/* Index-batch window gather (signal/index_dataset.py:32-57; examples/indexBatching: "GPU-index-batching"): for every
 * sample b, X[b] = data[idx[b] : idx[b] + h], Y[b] = data[idx[b] + h : idx[b] + 2 h] from the resident series
 * data [T_total, W] (W = nodes * features), both windows of all B samples in one launch.  time_major != 0 writes
 * [h][B][W] instead of [B][h][W].  idx: int64 [B] on the device; the caller guarantees 0 <= idx[b] <= T_total - 2 h
 * (out-of-range rows are clamped, never read outside the series). */
int pgt_window_gather_f32(const float* data, int64_t T_total, int64_t W, const int64_t* idx, int64_t B, int64_t h,
                          float* X, float* Y, int time_major, pgt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PGT_HIP_H_ */
