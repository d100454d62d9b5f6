"""Graph handles: device-resident CSR operators and their ELLW layouts, graph preparation, the identity-keyed graph cache.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""
import ctypes
from collections import OrderedDict

import os
import warnings

import torch

from .. import _lib
from .._lib import CsrStruct, DConvGraphStruct, EllwStruct, SymGraphStruct, PgtError, check_tensor, ptr, stream_of

F32 = torch.float32
I32 = torch.int32


# --------------------------------------------------------------------------------------------- graph handles

class Csr:
    """CSR operator by destination row.  `halo` is the operator's measured locality: 32 / 96 when at least 95 % of the
    slots have |col - row| within that distance (locality-ordered node numbering), else 0; `max_len` the longest row.
    `ellw` caches the ELLW layout (pgt_ellw) built for the F = 64 LDS-window kernel on first use."""
    __slots__ = ("rowptr", "col", "val", "n_rows", "halo", "max_len", "nnz", "ellw", "long_rows", "family", "short_len", "left_rows")   # ellw: None | Ellw | False

    def __init__(self, n_rows, cap, device):
        self.n_rows = n_rows
        self.halo = 0
        self.max_len = -1
        self.nnz = -1
        self.ellw = None
        self.long_rows = None      # int32 device list of the rows longer than LONG_ROW slots (hubs), or None
        self.left_rows = None      # int32 device list of the rows an ELLW layout leaves out (longer than its 32 slots: hubs and the odd
        #                            junction), when they are few (ELLW_LEFT_MAX_FRACTION); None: no such rows, or too many for the layout
        self.short_len = -1        # the longest row among the others (what that layout is planned for)
        self.family = None         # dict shared by the operators of one graph (forward / transposed, both directions): what one of
        #                            them learned about a renumbering serves the others (same undirected neighbourhoods)
        self.rowptr = torch.zeros(n_rows + 1, dtype=I32, device=device)
        self.col = torch.zeros(max(cap, 1), dtype=I32, device=device)
        self.val = torch.zeros(max(cap, 1), dtype=F32, device=device)

    def struct(self):
        return CsrStruct(ptr(self.rowptr), ptr(self.col), ptr(self.val))


LONG_ROW = 128         # rows with more slots go to the long-row kernel (one workgroup per row: pgt_spmm_csr_long_f32)
LONG_ROW_CAP = 4096    # at most this many long rows are listed; an operator with more keeps the plain row tiles
ELLW_MAX_SLOTS = 32    # the widest row of an ELLW layout (pgt_ellw_plan); longer rows are left out of it when they are few:
ELLW_LEFT_MIN, ELLW_LEFT_MAX_FRACTION = 64, 1.0 / 400      # at most max(64, n_rows / 400) of them — about one per tile, so that each can
#                                                              ride with a tile — else the operator keeps the CSR kernels
ELLW_MIN_ROWS = 4096   # below this the whole X fits a CU's L1/L2 slice anyway; keep the CSR row tiles
# Locality-ordered operators (>= 95 % of the slots within +-32 / +-96 rows, rows of at most 32 slots) take the ELLW
# layout at F = 64: 21 us against 33 us for the CSR row tiles at N = 200 000, in-degree 8.
# PGT_ELLW=0 keeps every operator on the CSR kernels (A/B).
USE_ELLW = os.environ.get("PGT_ELLW", "1") != "0"
# hubs of an ELLW operator: their slots ride with the tiles of the window kernel (pgt_ellw.hub_*) instead of a workgroup per hub
# behind it (pgt_spmm_csr_rows_f32).  PGT_HUB_FOLD=0: the second form (A/B)
USE_HUB_FOLD = os.environ.get("PGT_HUB_FOLD", "1") != "0"
# a renumbered layout hands the kernel its outside rows' X rows directly (pgt_ellw.far_src).  PGT_FAR_SRC=0: through `order` (A/B)
USE_FAR_SRC = os.environ.get("PGT_FAR_SRC", "1") != "0"


def measure_locality(csrs):
    """Set `halo`, `max_len`, `nnz` on each operator from pgt_csr_locality (one small launch per operator, ONE host
    read for all)."""
    lib = _lib.get_lib()
    todo = [c for c in csrs if c.n_rows >= ELLW_MIN_ROWS]
    if not todo:
        return
    family = {}
    for c in todo:
        c.family = family
    dev = todo[0].rowptr.device
    out = torch.zeros(len(todo), 5, dtype=I32, device=dev)
    lists = [torch.empty(LONG_ROW_CAP, dtype=I32, device=dev) for _ in todo]
    for i, c in enumerate(todo):
        lib.call("pgt_csr_locality", ptr(c.rowptr), ptr(c.col), c.n_rows, ptr(out[i]), ptr(lists[i]), LONG_ROW_CAP,
                 LONG_ROW, stream_of(lib, c.rowptr))
        out[i, 4:5].copy_(c.rowptr[c.n_rows:c.n_rows + 1])
    for c, lst, (n32, n96, max_len, n_long, nnz) in zip(todo, lists, out.tolist()):
        c.halo = 32 if 20 * n32 >= 19 * nnz > 0 else (96 if 20 * n96 >= 19 * nnz > 0 else 0)
        c.max_len, c.nnz = max_len, nnz
        c.long_rows = lst[:n_long] if 0 < n_long <= LONG_ROW_CAP else None
        c.short_len, c.left_rows = max_len, None
    # an operator whose rows fit the ELLW layout but for a few (a hub, the odd junction with 40 in-edges) keeps the layout: those
    # rows are left out of it and produced separately (Ellw.left_out).  Graph preparation: one more host read, for all of them.
    wide = [c for c in todo if c.max_len > ELLW_MAX_SLOTS]
    if wide:
        found = []
        for c in wide:
            ln = c.rowptr[1:c.n_rows + 1] - c.rowptr[:c.n_rows]
            over = ln > ELLW_MAX_SLOTS
            found.append((torch.nonzero(over).flatten().to(I32), torch.where(over, torch.zeros_like(ln), ln).max()))
        short = torch.stack([f[1] for f in found]).tolist()
        for c, (rows, _), m in zip(wide, found, short):
            if 0 < rows.numel() <= max(ELLW_LEFT_MIN, int(c.n_rows * ELLW_LEFT_MAX_FRACTION)):
                c.left_rows, c.short_len = rows.contiguous(), int(m)
        hubs = [c.left_rows for c in wide if c.left_rows is not None]
        if hubs:      # (a hub of one operator is a hub COLUMN of its transpose: the patch order of the graph is grown without any of them)
            family["hub_nodes"] = torch.unique(torch.cat(hubs).cpu().long())


class Ellw:
    """The ELLW layout of one Csr (include/pgt_hip.h: pgt_ellw) — slot block, coefficient block or per-source scale
    table, geometry — built on the device by pgt_ellw_build."""

    def __init__(self, csr, halo):
        lib = _lib.get_lib()
        dev = csr.rowptr.device
        self.halo = int(halo)
        # rows longer than the layout's 32 slots (csr.left_rows: hubs, the odd junction) are left out of it: the window kernel skips them
        # and pgt_spmm_csr_rows_f32 produces them (ops.spmm); the layout is planned for the longest of the OTHER rows
        lr = getattr(csr, "left_rows", None)
        self.left_out = 0 if lr is None else int(lr.numel())
        self.plan_len = int(csr.max_len) if lr is None else int(csr.short_len)
        # first as a source-scaled operator (P_o of DConv); the build verifies that and reports the slots outside
        # their window — if it is not, lay it out again with per-slot coefficients (whose plan leaves fewer far rows)
        mismatch = self._build(lib, csr, dev, True)
        if mismatch:
            self._build(lib, csr, dev, False)
        if self.left_out and USE_HUB_FOLD:
            self._hub_tables(self._caller_csr or csr, dev)

    def _hub_tables(self, csr, dev):
        """The hubs' slots cut into pieces that ride with the tiles (pgt_ellw.hub_*): piece s of hub h = its slots
        [s C_h, (s + 1) C_h), C_h = ceil(len_h / split), at most P = 2 lane groups' worth per tile.  Graph preparation: torch
        indexing, one host read (the hubs' lengths)."""
        hubs = csr.left_rows
        n_hub = int(hubs.numel())
        split = int(self.n_tiles // n_hub)
        lanes, waves = (32, 8) if self.config == 2 else (64, 16)
        P = 2 * lanes
        if split < 1:
            return
        a = csr.rowptr[hubs.long()].long()
        lens = csr.rowptr[hubs.long() + 1].long() - a
        C = (lens + split - 1) // split
        if int(C.max()) > P:                                   # (a hub too long for its pieces: pgt_spmm_csr_rows_f32 keeps it)
            return
        j = torch.arange(P, device=dev).view(1, 1, P)
        s_ = torch.arange(split, device=dev).view(1, split, 1)
        off = s_ * C.view(-1, 1, 1) + j
        live = (j < C.view(-1, 1, 1)) & (off < lens.view(-1, 1, 1))
        q = (a.view(-1, 1, 1) + off).clamp_(max=max(int(csr.nnz) - 1, 0))
        self.hub_col = torch.where(live, csr.col[q], torch.full_like(csr.col[q], -1)).contiguous().view(-1)
        self.hub_val = torch.where(live, csr.val[q], torch.zeros_like(csr.val[q])).contiguous().view(-1)
        self.hub_rows = hubs.contiguous()
        self.hub_partial = torch.empty(n_hub * split * waves * 64, dtype=F32, device=dev)
        self.hub_split = split

    def _build(self, lib, csr, dev, source_scaled):
        tr, w, cfg, nt, fr = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int32(0)
        lib.call("pgt_ellw_plan", csr.n_rows, self.halo, self.plan_len, 1 if source_scaled else 0, ctypes.byref(tr),
                 ctypes.byref(w), ctypes.byref(cfg), ctypes.byref(nt), ctypes.byref(fr))
        self.tile_rows, self.width, self.n_tiles, self.config = tr.value, w.value, nt.value, cfg.value
        self.far_rows = fr.value
        total = self.n_tiles * self.tile_rows * self.width
        self.slots = torch.empty(total, dtype=torch.int16, device=dev)      # uint16 bit patterns
        vals = None if source_scaled else torch.empty(total, dtype=F32, device=dev)
        scale = torch.empty(csr.n_rows, dtype=F32, device=dev) if source_scaled else None
        far_col = torch.empty(self.n_tiles * self.far_rows, dtype=I32, device=dev)
        far_cnt = torch.empty(self.n_tiles, dtype=I32, device=dev)
        info = torch.zeros(4, dtype=I32, device=dev)
        geo = EllwStruct(None, None, None, self.tile_rows, self.halo, self.width, self.config, self.n_tiles, None,
                         self.far_rows)
        lib.call("pgt_ellw_build", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.n_rows, int(csr.nnz),
                 ctypes.byref(geo), ptr(self.slots), ptr(vals), ptr(scale), ptr(far_col), ptr(far_cnt), ptr(info),
                 stream_of(lib, csr.rowptr))
        # far = slots outside their window, far_csr = those that did not fit the tile's table (served through the CSR)
        self.far, mismatch, overflow, self.far_csr = info.tolist()        # one host sync per new operator
        if overflow != self.left_out:
            raise PgtError(f"ELLW: {overflow} row(s) longer than the planned width {self.width}, {self.left_out} listed as hubs")
        self.scale, self.vals = scale, vals
        self.far_col = far_col if self.far else None                   # no table: the kernel skips its loads
        return mismatch if source_scaled else 0

    order = None      # int32 [n_rows] of a renumbered layout (RenumberedEllw), None: the caller's numbering
    left_out = 0      # rows the layout leaves out (csr.left_rows): ops.spmm produces them with pgt_spmm_csr_rows_f32 ...
    hub_col = hub_val = hub_rows = hub_partial = None       # ... unless their pieces ride with the tiles (F = 64: _hub_tables)
    hub_split = 0
    _caller_csr = None   # RenumberedEllw: the operator in the caller's numbering (the layout's own CSR is in layout numbering)
    far_src = None    # a renumbered layout: order[far_col], the outside rows' X rows in the caller's numbering
    csr = None        # the operator in LAYOUT numbering (RenumberedEllw); None: the caller's own CSR serves the layout

    def struct(self):
        return EllwStruct(ptr(self.slots), ptr(self.vals), ptr(self.scale), self.tile_rows, self.halo, self.width,
                          self.config, self.n_tiles, ptr(self.far_col), self.far_rows, ptr(self.order),
                          ptr(self.hub_col), ptr(self.hub_val), ptr(self.hub_rows), ptr(self.hub_partial),
                          0 if self.hub_col is None else int(self.hub_rows.numel()), self.hub_split, ptr(self.far_src))


class _LayoutCsr:
    """The operator of a renumbered layout: rowptr / col / val in layout numbering (what pgt_ellw_build reads and what
    serves a slot that found no place in its tile's table)."""
    __slots__ = ("rowptr", "col", "val", "n_rows", "max_len", "nnz", "left_rows", "short_len")


class RenumberedEllw(Ellw):
    """The ELLW layout of an operator whose tiles are NOT compact in the caller's numbering, in a numbering where they are
    (pgt_tile_order_host: patches grown on the host, once per graph).  Nothing moves in HBM: `order` goes into the layout and
    the kernel reads / writes whole X / Y / T rows through it, so `spmm` still answers in the caller's numbering and a
    K-hop stack chains hop after hop without a permutation pass.  Rows keep their slots in the caller's order: the sums
    round exactly as on the caller's CSR."""

    def __init__(self, csr, order=None):
        """order: (host int32 order, tile_rows) found for another operator of the same graph, or None: grow the patches here."""
        lib = _lib.get_lib()
        dev = csr.rowptr.device
        n, nnz = csr.n_rows, int(csr.nnz)
        tr, w, cfg, nt, fr = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int32(0)
        left = getattr(csr, "left_rows", None)                       # rows wider than the layout (hubs, junctions): left out of it
        plan_len = int(csr.short_len) if left is not None else int(csr.max_len)
        lib.call("pgt_ellw_plan", n, 0, plan_len, 1, ctypes.byref(tr), ctypes.byref(w), ctypes.byref(cfg),
                 ctypes.byref(nt), ctypes.byref(fr))
        rowptr_h = csr.rowptr[:n + 1].cpu().contiguous()            # graph preparation: one round trip per operator
        col_h = csr.col[:nnz].cpu().contiguous()
        given = order is not None and order[1] == tr.value and order[0].numel() == n
        order_h = order[0] if given else torch.empty(n, dtype=I32)
        rowptr_p = torch.empty(n + 1, dtype=I32)
        col_p, slot_p = torch.empty(max(nnz, 1), dtype=I32), torch.empty(max(nnz, 1), dtype=I32)
        fam = getattr(csr, "family", None) or {}
        hub_nodes = fam.get("hub_nodes")                             # nodes with a wide row in ANY operator of the graph (host int64)
        if hub_nodes is None and left is not None:
            hub_nodes = left.cpu().long()
        if hub_nodes is not None and hub_nodes.numel() and not given and nnz:
            # the patches are grown on the operator WITHOUT its hubs, as rows and as sources: a hub that joins a patch would pull its
            # thousands of neighbours — rows from all over the graph — in behind it (the layout then overflows its tables and is
            # rejected); the order found, the full operator is laid out in it below
            is_hub = torch.zeros(n, dtype=torch.bool)
            is_hub[hub_nodes] = True
            lens = (rowptr_h[1:] - rowptr_h[:-1]).long()
            row_of = torch.repeat_interleave(torch.arange(n), lens)
            keep = ~(is_hub[row_of] | is_hub[col_h.long()])
            rowptr_e = torch.zeros(n + 1, dtype=I32)
            rowptr_e[1:] = torch.cumsum(torch.bincount(row_of[keep], minlength=n), 0)
            col_e = col_h[keep].contiguous()
            scratch = [torch.empty(max(int(col_e.numel()), 1), dtype=I32) for _ in range(2)]
            lib.call("pgt_tile_order_host", rowptr_e.data_ptr(), col_e.data_ptr(), n, tr.value, 0, order_h.data_ptr(),
                     rowptr_p.data_ptr(), scratch[0].data_ptr(), scratch[1].data_ptr())
            given = True
        lib.call("pgt_tile_order_host", rowptr_h.data_ptr(), col_h.data_ptr(), n, tr.value, 1 if given else 0, order_h.data_ptr(),
                 rowptr_p.data_ptr(), col_p.data_ptr(), slot_p.data_ptr())
        self.order_host = (order_h, tr.value)
        lay = _LayoutCsr()
        lay.rowptr, lay.col = rowptr_p.to(dev), col_p.to(dev)
        lay.val = csr.val[:nnz][slot_p[:nnz].to(dev).long()] if nnz else csr.val[:1].clone()
        lay.n_rows, lay.max_len, lay.nnz = n, csr.max_len, nnz
        lay.left_rows, lay.short_len = None, plan_len
        if left is not None:                                         # ... by their LAYOUT positions (what pgt_ellw_build sees)
            inv = torch.empty(n, dtype=torch.int64)
            inv[order_h.long()] = torch.arange(n)
            lay.left_rows = inv[left.cpu().long()].to(I32).to(dev)
        self.csr, self.order = lay, order_h.to(dev)
        self._caller_csr = csr                                       # the hub tables name X / Y rows: the caller's numbering
        super().__init__(lay, 0)
        if self.far_col is not None and USE_FAR_SRC:
            fc = self.far_col.long()
            self.far_src = torch.where(fc >= 0, self.order[fc.clamp(min=0)], torch.full_like(self.far_col, -1)).contiguous()


# an operator that is NOT a band (fewer than 95 % of the slots within +-96 rows) may still have compact tiles — a mesh
# numbered along a space-filling curve: 83 % within +-32, the rest in the patch's ring — which the layout's per-tile table
# of distinct outside rows carries: build it with the narrow halo and keep it when (almost) every slot found a place
ELLW_COMPACT_MAX_CSR_FRACTION = 0.002      # slots left to the CSR path (0xFFFF) for the layout to be kept
# an operator whose tiles are not compact in the caller's numbering either (a mesh numbered row by row: 0.48 of HBM on the
# CSR row tiles; a shuffled one: 0.22) is laid out in a numbering of the library's own (RenumberedEllw).  PGT_RENUMBER=0: off
USE_RENUMBER = os.environ.get("PGT_RENUMBER", "1") != "0"


def ellw_of(csr):
    """The cached ELLW layout of `csr`, built on first use; None when the layout does not apply."""
    e = getattr(csr, "ellw", None)
    if e is False:                     # tried and rejected
        return None
    hubs = getattr(csr, "left_rows", None) is not None
    plan_len = getattr(csr, "short_len", -1) if hubs else getattr(csr, "max_len", -1)
    if e is None and 0 <= plan_len <= 32 and getattr(csr, "nnz", 0) > 0:
        if getattr(csr, "halo", 0) > 0:
            e = csr.ellw = Ellw(csr, csr.halo)
        elif csr.n_rows >= ELLW_MIN_ROWS:
            cand = Ellw(csr, 32)           # (compact tiles: hubs / junctions are left out of this layout like of a band's)
            fam = getattr(csr, "family", None)
            if fam is None:
                fam = {}
            if cand.far_csr > ELLW_COMPACT_MAX_CSR_FRACTION * csr.nnz and USE_RENUMBER and not fam.get("no_patches"):
                cand = RenumberedEllw(csr, fam.get("order"))
                if cand.far_csr <= ELLW_COMPACT_MAX_CSR_FRACTION * csr.nnz:
                    fam["order"] = cand.order_host          # the graph's other operators lay themselves out in the same patches
                else:
                    fam["no_patches"] = True                # (a graph without locality: its other operators need not try)
            if cand.far_csr <= ELLW_COMPACT_MAX_CSR_FRACTION * csr.nnz:
                e = csr.ellw = cand
            else:
                csr.ellw = False
    return e


def _window_kernel_covers(*operands):
    """What spmm_ellw64_kernel needs of X / Y / T (a renumbered layout has no CSR fallback inside the C entry point: the
    caller's CSR is used from here instead): 16-byte aligned rows."""
    return all(t is None or (t.data_ptr() % 16 == 0 and (t.size(0) <= 1 or t.stride(0) % 4 == 0)) for t in operands)


def _edge_inputs(lib, edge_index, edge_weight):
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError("edge_index must have shape [2, E]")
    check_tensor(lib, edge_index, "edge_index", torch.int64)
    ei = edge_index.contiguous()
    ew = None
    if edge_weight is not None:
        ew = edge_weight
        if ew.requires_grad and torch.is_grad_enabled():
            # graph preparation runs outside autograd: the reference's GCNConv / ChebConv would propagate a gradient
            # to learnable edge weights, this path does not — say so instead of dropping it silently
            warnings.warn("pytorch_geometric_temporal_amd: edge_weight requires grad, but graph preparation is not "
                          "differentiable here — no gradient will reach edge_weight", stacklevel=3)
        if ew.dtype != F32:
            ew = ew.to(F32)
        check_tensor(lib, ew, "edge_weight", F32)
        ew = ew.contiguous()
        if ew.numel() != ei.size(1):
            raise ValueError("edge_weight must have one entry per edge")
    return ei, ew


class DConvGraph:
    """Device-resident operators of DConv / BatchedDConv (dcrnn.py:59-77, :277-290) for one (edge_index, edge_weight)."""

    def __init__(self, edge_index, edge_weight, num_nodes, validate=True, strict_dense=False):
        lib = _lib.get_lib()
        ei, ew = _edge_inputs(lib, edge_index, edge_weight)
        dev = ei.device
        E, N = ei.size(1), int(num_nodes)
        self.N, self.E, self.device = N, E, dev
        self.fwd_o, self.fwd_i = Csr(N, E, dev), Csr(N, E, dev)
        self.bwd_o, self.bwd_i = Csr(N, E, dev), Csr(N, E, dev)
        self.deg_out = torch.zeros(max(N, 1), dtype=F32, device=dev)
        self.deg_in = torch.zeros(max(N, 1), dtype=F32, device=dev)
        self.info = torch.zeros(4, dtype=I32, device=dev)
        ws_bytes = lib.prep_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        st = DConvGraphStruct(self.fwd_o.struct(), self.fwd_i.struct(), self.bwd_o.struct(), self.bwd_i.struct(),
                              ptr(self.deg_out), ptr(self.deg_in), ptr(self.info))
        lib.call("pgt_dconv_prep", ptr(ei), ptr(ew), E, N, ctypes.byref(st), ptr(ws), ws_bytes, stream_of(lib, ei))
        measure_locality((self.fwd_o, self.fwd_i, self.bwd_o, self.bwd_i))
        if validate:
            dup, zero, oob, nonfinite = self.info.tolist()  # one host sync per *new* graph
            self.finite = nonfinite == 0          # every coefficient of both operators is finite (no node without in- / out-edges)
            if oob:
                raise IndexError(f"edge_index has {oob} endpoint(s) outside [0, {N})")
            if strict_dense and (dup or zero):
                # DConv's dense path (to_dense_adj sums duplicates, dense_to_sparse drops zeros) makes the reversed
                # edge list shorter than norm_in and the reference fails with a shape mismatch in message().
                raise RuntimeError(
                    f"DConv: edge list has {dup} duplicate edge(s) and {zero} zero weight(s); the reference's dense "
                    f"adjacency path (dcrnn.py:59-77) cannot broadcast norm_in over the shortened reverse edge list")


class SymGraph:
    """GCN-normalised (gcn_norm) or scaled-Laplacian (ChebConv.__norm__) operator and its transpose."""

    def __init__(self, kind, edge_index, edge_weight, num_nodes, improved=False, add_self_loops=True,
                 normalization="sym", lambda_max=None, variant=0, validate=True, batch=None):
        lib = _lib.get_lib()
        ei, ew = _edge_inputs(lib, edge_index, edge_weight)
        dev = ei.device
        E, N = ei.size(1), int(num_nodes)
        self.N, self.E, self.device = N, E, dev
        cap = E + 2 * N
        self.fwd, self.bwd = Csr(N, cap, dev), Csr(N, cap, dev)
        self.deg = torch.zeros(max(N, 1), dtype=F32, device=dev)
        self.info = torch.zeros(4, dtype=I32, device=dev)
        ws_bytes = lib.prep_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        st = SymGraphStruct(self.fwd.struct(), self.bwd.struct(), ptr(self.deg), ptr(self.info))
        if kind == "gcn":
            lib.call("pgt_gcn_prep", ptr(ei), ptr(ew), E, N, int(bool(improved)), int(bool(add_self_loops)),
                     ctypes.byref(st), ptr(ws), ws_bytes, stream_of(lib, ei))
        elif kind == "cheb":
            norm_code = {None: 0, "sym": 1, "rw": 2}[normalization]
            if batch is not None:
                # one lambda_max per graph of a disjoint batch (astgcn.py:97-98): lambda_max [G] and batch [N] stay on the device
                lam_vec = lambda_max.detach().to(device=dev, dtype=F32).contiguous().view(-1)
                labels = batch.detach().to(device=dev, dtype=torch.int64).contiguous().view(-1)
                if labels.numel() < N:
                    raise IndexError(f"batch has {labels.numel()} labels for {N} nodes")
                lib.call("pgt_cheb_prep_graphs", ptr(ei), ptr(ew), E, N, norm_code, ptr(labels), ptr(lam_vec), lam_vec.numel(),
                         int(variant), ctypes.byref(st), ptr(ws), ws_bytes, stream_of(lib, ei))
            else:
                lam = float("nan") if lambda_max is None else float(lambda_max)
                lib.call("pgt_cheb_prep", ptr(ei), ptr(ew), E, N, norm_code, lam, int(variant), ctypes.byref(st),
                         ptr(ws), ws_bytes, stream_of(lib, ei))
        else:
            raise ValueError(kind)
        measure_locality((self.fwd, self.bwd))
        if validate:
            info = self.info.tolist()
            if info[2]:
                raise IndexError(f"edge_index has {info[2]} endpoint(s) outside [0, {N})")
            if info[3]:
                raise IndexError(f"batch has {info[3]} label(s) outside [0, {lambda_max.numel()}) (one lambda_max per graph)")


def tensor_version(t):
    """In-place version counter of `t`; inference tensors (created under torch.inference_mode()) do not track one and
    cannot be mutated in place outside inference mode, so a constant stands in for it."""
    return 0 if t.is_inference() else t._version


class _GraphCache:
    """Identity-keyed cache (data_ptr + in-place version counter), never torch.equal (no host sync per forward;
    the reference compares with torch.equal twice per BatchedDCRNN forward, dcrnn.py:446-447).

    What identity cannot see: a write that leaves `_version` alone — `edge_weight.data.mul_(2)`, `edge_weight.data.copy_(...)`, a
    kernel of the caller's own writing through the pointer.  Two ways to have it seen: `GRAPH_CACHE.forget(edge_index, edge_weight)`
    after such a write, or `verify = True` (PGT_GRAPH_VERIFY=1): every hit then compares a checksum of the VALUES with the one taken
    when the operators were built — one host synchronisation per forward, which is what the reference's own `torch.equal` costs."""

    def __init__(self, capacity=128):      # a dynamic-graph signal holds one edge list per snapshot (England-Covid: 53)
        self.capacity = capacity
        self._d = OrderedDict()
        self.verify = os.environ.get("PGT_GRAPH_VERIFY", "0") == "1"

    @staticmethod
    def _tkey(t):
        if t is None:
            return None
        return (t.data_ptr(), tensor_version(t), tuple(t.shape), tuple(t.stride()), str(t.device), t.dtype)

    @staticmethod
    def _checksum(edge_index, edge_weight):
        """Two device scalars that change with (almost) any change of the values: position-weighted sums of the endpoints and of
        the weights' bit patterns."""
        ei = edge_index.reshape(-1).to(torch.int64)
        pos = torch.arange(1, ei.numel() + 1, device=ei.device, dtype=torch.int64)
        c = [(ei * pos).sum()]
        if edge_weight is not None:
            w = edge_weight.detach().reshape(-1).contiguous().to(torch.float32).view(torch.int32).to(torch.int64)
            c.append((w * pos[:w.numel()]).sum())
        return torch.stack(c)

    def get(self, tag, edge_index, edge_weight, extra, builder):
        # inference tensors carry no version counter, so an in-place edit under torch.inference_mode() would go unnoticed:
        # graphs given as inference tensors are prepared on every call instead of being cached
        if any(t is not None and t.is_inference() for t in (edge_index, edge_weight)):
            return builder()
        key = (tag, self._tkey(edge_index), self._tkey(edge_weight), extra)
        hit = self._d.get(key)
        if hit is not None:
            if self.verify and not bool(torch.equal(hit[3], self._checksum(edge_index, edge_weight))):
                del self._d[key]           # same tensors, other values: a write that did not bump the version counter
            else:
                self._d.move_to_end(key)
                return hit[0]
        g = builder()
        # keep the key tensors alive so their storage (data_ptr) cannot be recycled while the entry lives
        self._d[key] = (g, edge_index, edge_weight, self._checksum(edge_index, edge_weight) if self.verify else None)
        if len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def forget(self, edge_index, edge_weight=None):
        """Drop every entry built from these tensors (after a write through `.data` that the version counter did not record)."""
        ki, kw = self._tkey(edge_index), self._tkey(edge_weight)
        for key in [k for k in self._d if k[1] == ki and (edge_weight is None or k[2] == kw)]:
            del self._d[key]

    def clear(self):
        self._d.clear()


GRAPH_CACHE = _GraphCache()


def dconv_graph(edge_index, edge_weight, num_nodes, strict_dense=False):
    return GRAPH_CACHE.get("dconv", edge_index, edge_weight, (int(num_nodes), bool(strict_dense)),
                           lambda: DConvGraph(edge_index, edge_weight, num_nodes, strict_dense=strict_dense))


def gcn_graph(edge_index, edge_weight, num_nodes, improved=False, add_self_loops=True):
    return GRAPH_CACHE.get("gcn", edge_index, edge_weight, (int(num_nodes), bool(improved), bool(add_self_loops)),
                           lambda: SymGraph("gcn", edge_index, edge_weight, num_nodes, improved=improved,
                                            add_self_loops=add_self_loops))


class RawGraph:
    """The edge list as it is (no normalisation, no added self-loops): out[i] = sum_{e: col_e = i} w_e x[row_e], the
    propagate of `GCNConv_Fixed_W(normalize=False)` (evolvegcno.py:92-101), and its transpose for the gradient.
    Built with device sorts (stable by destination / by source: slots keep the edge order inside a row, like the
    prep kernels); `.fwd` / `.bwd` are what `propagate` takes."""

    def __init__(self, edge_index, edge_weight, num_nodes):
        lib = _lib.get_lib()
        ei, ew = _edge_inputs(lib, edge_index, edge_weight)
        dev, E, N = ei.device, ei.size(1), int(num_nodes)
        if E:
            lo, hi = torch.aminmax(ei)                            # ONE reduction and one host read per new edge tensor
            lo, hi = torch.stack((lo, hi)).tolist()
            if lo < 0 or hi >= N:
                raise IndexError(f"edge_index has endpoint(s) outside [0, {N})")
        w = ew if ew is not None else torch.ones(E, dtype=F32, device=dev)
        self.N, self.E, self.device = N, E, dev
        self.fwd, self.bwd = Csr(N, E, dev), Csr(N, E, dev)
        for csr, row, col in ((self.fwd, ei[1], ei[0]), (self.bwd, ei[0], ei[1])):
            order = torch.argsort(row, stable=True)
            counts = torch.bincount(row, minlength=N)
            csr.rowptr[1:] = torch.cumsum(counts, 0).to(I32)
            if E:
                csr.col[:E] = col[order].to(I32)
                csr.val[:E] = w[order]
        measure_locality((self.fwd, self.bwd))


def raw_graph(edge_index, edge_weight, num_nodes):
    return GRAPH_CACHE.get("raw", edge_index, edge_weight, (int(num_nodes),),
                           lambda: RawGraph(edge_index, edge_weight, num_nodes))


class SmallEdges:
    """An edge list handed to the one-launch small-graph kernels as it is (csrc/small_gcn.hip builds its lists in LDS):
    shape / dtype / range checked ONCE per tensor (identity-keyed, like every prepared graph), nothing sorted, no workspace."""
    _info = {}

    def __init__(self, edge_index, edge_weight, num_nodes):
        lib = _lib.get_lib()
        ei, ew = _edge_inputs(lib, edge_index, edge_weight)
        N, E = int(num_nodes), ei.size(1)
        if E:
            lo, hi = torch.aminmax(ei)                            # ONE reduction and one host read per new edge tensor
            lo, hi = torch.stack((lo, hi)).tolist()
            if lo < 0 or hi >= N:
                raise IndexError(f"edge_index has endpoint(s) outside [0, {N})")
        self.ei, self.ew, self.E, self.N = ei, ew, E, N
        key = str(ei.device)
        if key not in SmallEdges._info:
            SmallEdges._info[key] = torch.zeros(4, dtype=I32, device=ei.device)
        self.info = SmallEdges._info[key]


def small_edges(edge_index, edge_weight, num_nodes):
    return GRAPH_CACHE.get("small", edge_index, edge_weight, (int(num_nodes),),
                           lambda: SmallEdges(edge_index, edge_weight, num_nodes))


def cheb_lambda(lambda_max, batch):
    """ChebConv / ChebConvAttention's `lambda_max` argument as (scalar or None, per-graph tensor or None): a tensor with more
    than one value selects one lambda per graph through `batch` (astgcn.py:97-98, PyG ChebConv.__norm__); `batch` beside a
    single value (or none) changes nothing there either."""
    if isinstance(lambda_max, torch.Tensor) and lambda_max.numel() > 1:
        if batch is None:
            # the reference divides the [E' + N] Laplacian entries by the tensor as it is: a size mismatch unless it is per entry
            raise RuntimeError(f"lambda_max has {lambda_max.numel()} values: pass `batch` (one graph label per node) with it")
        return None, lambda_max
    return (None if lambda_max is None else float(lambda_max)), None


def cheb_graph(edge_index, edge_weight, num_nodes, normalization="sym", lambda_max=None, variant=0, batch=None):
    if batch is not None and isinstance(lambda_max, torch.Tensor) and lambda_max.numel() > 1:
        # per-graph lambda_max: prepared on every call (keyed by four tensors' identity it would rarely hit)
        return SymGraph("cheb", edge_index, edge_weight, num_nodes, normalization=normalization, lambda_max=lambda_max,
                        variant=variant, batch=batch)
    lam = None if lambda_max is None else float(lambda_max)
    return GRAPH_CACHE.get("cheb", edge_index, edge_weight, (int(num_nodes), normalization, lam, int(variant)),
                           lambda: SymGraph("cheb", edge_index, edge_weight, num_nodes, normalization=normalization,
                                            lambda_max=lam, variant=variant))
