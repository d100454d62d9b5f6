// Tile-compact row numbering for the ELLW layout (host side of graph preparation).
//
// The reference aggregates over whatever edge list it is handed (dcrnn.py:300-313: propagate on the caller's node ids); a
// sensor graph arrives in file order.  spmm_ellw64_kernel (spmm.hip) needs the sources of a tile of consecutive rows to lie
// in the tile's own window (plus a short table of outside rows).  A numbering with that property exists for every graph that
// embeds in a low-dimensional space — it is a partition into patches — and the caller's numbering need not be it.  This file
// finds one: rows are taken patch by patch, a patch grown from a seed by always adding the unassigned row with the most
// neighbours already inside the patch (ties: first discovered), the next seed the oldest row left on the rim of finished
// patches.  Nothing is renumbered in HBM: the permutation goes into the layout (pgt_ellw.order) and the kernel reads and
// writes X / Y / T rows through it, so a product in the caller's numbering costs no permutation pass.
//
// O(E) on the host, once per prepared graph (200 000 rows x 8 slots: ~40 ms).  Host pointers in, host pointers out.
#include <algorithm>
#include <vector>

#include "pgt_common.h"

namespace {

// order[p] = the row at position p: patches of tile_rows rows, grown one after the other
void grow_patches(const int32_t* rowptr, const int32_t* col, int n, int64_t nnz, int32_t tile_rows, int32_t* order) {
  // undirected neighbourhood: a row's sources and the rows it is a source of
  std::vector<int32_t> aptr((size_t)n + 1, 0), adj((size_t)2 * nnz);
  for (int i = 0; i < n; ++i)
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) { ++aptr[i + 1]; ++aptr[col[q] + 1]; }
  for (int i = 0; i < n; ++i) aptr[i + 1] += aptr[i];
  {
    std::vector<int32_t> fill(aptr.begin(), aptr.end() - 1);
    for (int i = 0; i < n; ++i)
      for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) { adj[fill[i]++] = col[q]; adj[fill[col[q]]++] = i; }
  }
  constexpr int GMAX = 64;                                 // gains above this share the top bucket
  std::vector<uint8_t> taken((size_t)n, 0);
  std::vector<int32_t> gain((size_t)n, 0), stamp((size_t)n, -1), seen;
  std::vector<int32_t> bucket[GMAX + 1];
  size_t head[GMAX + 1];
  std::vector<int32_t> rim;                                // rows discovered next to a finished patch, oldest first
  size_t rim_head = 0;
  int next_fresh = 0, placed = 0, patch = 0;
  while (placed < n) {
    const int want = std::min<int>(tile_rows, n - placed);
    for (int g = 0; g <= GMAX; ++g) { bucket[g].clear(); head[g] = 0; }
    seen.clear();
    int top = 0, got = 0;
    while (got < want) {
      int v = -1;
      while (top > 0) {                                    // the best row on this patch's frontier
        auto& b = bucket[top];
        while (head[top] < b.size()) {
          const int c = b[head[top]++];
          if (!taken[c] && stamp[c] == patch && std::min(gain[c], GMAX) == top) { v = c; break; }
        }
        if (v >= 0) break;
        --top;
      }
      if (v < 0) {                                         // no frontier (first row of the patch, or the component ended)
        while (rim_head < rim.size() && taken[rim[rim_head]]) ++rim_head;
        if (rim_head < rim.size()) v = rim[rim_head++];
        else { while (taken[next_fresh]) ++next_fresh; v = next_fresh; }
      }
      taken[v] = 1;
      order[placed + got] = v;
      ++got;
      for (int q = aptr[v]; q < aptr[v + 1]; ++q) {
        const int u = adj[q];
        if (taken[u]) continue;
        if (stamp[u] != patch) { stamp[u] = patch; gain[u] = 0; seen.push_back(u); }
        const int g = std::min(++gain[u], GMAX);
        bucket[g].push_back(u);
        if (g > top) top = g;
      }
    }
    for (int u : seen) if (!taken[u]) rim.push_back(u);
    placed += got;
    ++patch;
  }
}

}  // namespace

extern "C" int pgt_tile_order_host(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t tile_rows,
                                   int32_t order_given, int32_t* order, int32_t* rowptr_p, int32_t* col_p, int32_t* slot_p) {
  PGT_REQUIRE(rowptr && col && order && rowptr_p && col_p && slot_p, "pgt_tile_order_host: null pointer");
  PGT_REQUIRE(n_rows >= 1 && n_rows < ((int64_t)1 << 31) - 1024 && tile_rows >= 1, "pgt_tile_order_host: bad size");
  const int n = (int)n_rows;
  const int64_t nnz = rowptr[n];
  PGT_REQUIRE(rowptr[0] == 0 && nnz >= 0 && nnz < ((int64_t)1 << 31), "pgt_tile_order_host: bad rowptr");
  for (int i = 0; i < n; ++i) PGT_REQUIRE(rowptr[i + 1] >= rowptr[i], "pgt_tile_order_host: rowptr not monotone");
  for (int64_t q = 0; q < nnz; ++q) PGT_REQUIRE(col[q] >= 0 && col[q] < n, "pgt_tile_order_host: column out of range");
  if (order_given) {
    // the order found for another operator of the same graph (its transpose, the other direction: the same undirected
    // neighbourhoods, so the same patches serve): only the operator in that numbering is produced
    std::vector<uint8_t> hit((size_t)n, 0);
    for (int p = 0; p < n; ++p) {
      PGT_REQUIRE(order[p] >= 0 && order[p] < n && !hit[order[p]], "pgt_tile_order_host: the given order is not a permutation");
      hit[order[p]] = 1;
    }
  } else {
    grow_patches(rowptr, col, n, nnz, tile_rows, order);
  }
  // the operator in the new numbering, every row keeping its slots IN THE ORDER of the caller's CSR (the sums then round
  // exactly as on the caller's CSR); slot_p[q'] = the caller's slot behind new slot q' (values: val_p = val[slot_p])
  std::vector<int32_t> pos((size_t)n);
  for (int p = 0; p < n; ++p) pos[order[p]] = p;
  rowptr_p[0] = 0;
  for (int p = 0; p < n; ++p) {
    const int i = order[p];
    int w = rowptr_p[p];
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q, ++w) { col_p[w] = pos[col[q]]; slot_p[w] = q; }
    rowptr_p[p + 1] = w;
  }
  return PGT_OK;
}
