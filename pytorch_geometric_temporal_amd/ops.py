"""Host-side wrappers over the C ABI (include/pgt_hip.h): graph handles, raw kernel calls on torch-owned device
memory, and the autograd Functions the nn.Module mirrors are built from.  PyTorch is plumbing here: it owns the
buffers and the stream; every arithmetic step on the path is a HIP kernel behind the C ABI.
"""
import ctypes
import math
from collections import OrderedDict

import os
import warnings

import torch

from . import _lib
from ._lib import CsrStruct, DConvGraphStruct, EllwStruct, RowMapStruct, SymGraphStruct, PgtError, check_tensor, ptr, stream_of

F32 = torch.float32
I32 = torch.int32


# --------------------------------------------------------------------------------------------- graph handles

class Csr:
    """CSR operator by destination row.  `halo` is the operator's measured locality: 32 / 96 when at least 95 % of the
    slots have |col - row| within that distance (locality-ordered node numbering), else 0; `max_len` the longest row.
    `ellw` caches the ELLW layout (pgt_ellw) built for the F = 64 LDS-window kernel on first use."""
    __slots__ = ("rowptr", "col", "val", "n_rows", "halo", "max_len", "nnz", "ellw", "long_rows", "family", "short_len")   # ellw: None | Ellw | False

    def __init__(self, n_rows, cap, device):
        self.n_rows = n_rows
        self.halo = 0
        self.max_len = -1
        self.nnz = -1
        self.ellw = None
        self.long_rows = None      # int32 device list of the rows longer than LONG_ROW slots (hubs), or None
        self.short_len = -1        # the longest row among the others (what an ELLW layout that leaves the hubs out is planned for)
        self.family = None         # dict shared by the operators of one graph (forward / transposed, both directions): what one of
        #                            them learned about a renumbering serves the others (same undirected neighbourhoods)
        self.rowptr = torch.zeros(n_rows + 1, dtype=I32, device=device)
        self.col = torch.zeros(max(cap, 1), dtype=I32, device=device)
        self.val = torch.zeros(max(cap, 1), dtype=F32, device=device)

    def struct(self):
        return CsrStruct(ptr(self.rowptr), ptr(self.col), ptr(self.val))


LONG_ROW = 128         # rows with more slots go to the long-row kernel (one workgroup per row: pgt_spmm_csr_long_f32)
LONG_ROW_CAP = 4096    # at most this many long rows are listed; an operator with more keeps the plain row tiles
ELLW_MIN_ROWS = 4096   # below this the whole X fits a CU's L1/L2 slice anyway; keep the CSR row tiles
# Locality-ordered operators (>= 95 % of the slots within +-32 / +-96 rows, rows of at most 32 slots) take the ELLW
# layout at F = 64: 21 us against 33 us for the CSR row tiles at N = 200 000, in-degree 8.
# PGT_ELLW=0 keeps every operator on the CSR kernels (A/B).
USE_ELLW = os.environ.get("PGT_ELLW", "1") != "0"
# Weight-gradient GEMMs of step t on a side stream while the main stream runs the BPTT chain of step t-1 (same
# arithmetic, fp32 atomics into dW either way).  Measured on MI355X at METR-LA shape, B = 1024: 23.81 ms per step with
# the overlap vs 23.79 ms with one whole-sequence weight-gradient GEMM at the end -> off by default.
OVERLAP_WEIGHT_GRADIENTS = False
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


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
        c.short_len = max_len
    hubs = [c for c in todo if c.long_rows is not None]
    if hubs:                             # (graph preparation of an operator with hubs: one more read, for all of them)
        lens = [c.rowptr[1:c.n_rows + 1] - c.rowptr[:c.n_rows] for c in hubs]
        short = torch.stack([torch.where(ln <= LONG_ROW, ln, torch.zeros_like(ln)).max() for ln in lens]).tolist()
        for c, m in zip(hubs, short):
            c.short_len = int(m)


class Ellw:
    """The ELLW layout of one Csr (include/pgt_hip.h: pgt_ellw) — slot block, coefficient block or per-source scale
    table, geometry — built on the device by pgt_ellw_build."""

    def __init__(self, csr, halo):
        lib = _lib.get_lib()
        dev = csr.rowptr.device
        self.halo = int(halo)
        # hubs (rows longer than LONG_ROW slots, listed in csr.long_rows) are left out of the layout: the window kernel skips them
        # and pgt_spmm_csr_rows_f32 produces them (ops.spmm); the layout is planned for the longest of the OTHER rows
        lr = getattr(csr, "long_rows", None)
        self.left_out = 0 if lr is None else int(lr.numel())
        self.plan_len = int(csr.max_len) if lr is None else int(csr.short_len)
        # first as a source-scaled operator (P_o of DConv); the build verifies that and reports the slots outside
        # their window — if it is not, lay it out again with per-slot coefficients (whose plan leaves fewer far rows)
        mismatch = self._build(lib, csr, dev, True)
        if mismatch:
            self._build(lib, csr, dev, False)

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
    left_out = 0      # hub rows the layout leaves out (csr.long_rows): ops.spmm produces them with pgt_spmm_csr_rows_f32
    csr = None        # the operator in LAYOUT numbering (RenumberedEllw); None: the caller's own CSR serves the layout

    def struct(self):
        return EllwStruct(ptr(self.slots), ptr(self.vals), ptr(self.scale), self.tile_rows, self.halo, self.width,
                          self.config, self.n_tiles, ptr(self.far_col), self.far_rows, ptr(self.order))


class _LayoutCsr:
    """The operator of a renumbered layout: rowptr / col / val in layout numbering (what pgt_ellw_build reads and what
    serves a slot that found no place in its tile's table)."""
    __slots__ = ("rowptr", "col", "val", "n_rows", "max_len", "nnz")


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
        lib.call("pgt_ellw_plan", n, 0, int(csr.max_len), 1, ctypes.byref(tr), ctypes.byref(w), ctypes.byref(cfg),
                 ctypes.byref(nt), ctypes.byref(fr))
        rowptr_h = csr.rowptr[:n + 1].cpu().contiguous()            # graph preparation: one round trip per operator
        col_h = csr.col[:nnz].cpu().contiguous()
        given = order is not None and order[1] == tr.value and order[0].numel() == n
        order_h = order[0] if given else torch.empty(n, dtype=I32)
        rowptr_p = torch.empty(n + 1, dtype=I32)
        col_p, slot_p = torch.empty(max(nnz, 1), dtype=I32), torch.empty(max(nnz, 1), dtype=I32)
        lib.call("pgt_tile_order_host", rowptr_h.data_ptr(), col_h.data_ptr(), n, tr.value, 1 if given else 0, order_h.data_ptr(),
                 rowptr_p.data_ptr(), col_p.data_ptr(), slot_p.data_ptr())
        self.order_host = (order_h, tr.value)
        lay = _LayoutCsr()
        lay.rowptr, lay.col = rowptr_p.to(dev), col_p.to(dev)
        lay.val = csr.val[:nnz][slot_p[:nnz].to(dev).long()] if nnz else csr.val[:1].clone()
        lay.n_rows, lay.max_len, lay.nnz = n, csr.max_len, nnz
        self.csr, self.order = lay, order_h.to(dev)
        super().__init__(lay, 0)


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
    hubs = getattr(csr, "long_rows", None) is not None
    plan_len = getattr(csr, "short_len", -1) if hubs else getattr(csr, "max_len", -1)
    if e is None and 0 <= plan_len <= 32 and getattr(csr, "nnz", 0) > 0:
        if getattr(csr, "halo", 0) > 0:
            e = csr.ellw = Ellw(csr, csr.halo)
        elif hubs:
            csr.ellw = False               # (compact-tile and renumbered layouts are not built around hubs)
        elif csr.n_rows >= ELLW_MIN_ROWS:
            cand = Ellw(csr, 32)
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


def gcn_small_fits(N, E, Fi, Fo):
    return bool(_lib.get_lib()._pgt_gcn_small_fits(int(N), int(E), int(Fi), int(Fo)))


class GcnSmallFunction(torch.autograd.Function):
    """out = A_hat (x W) from the raw edge list in one launch each way (pgt_gcn_small_f32; evolvegcno.py:76-101)."""

    @staticmethod
    def forward(ctx, x, W, edges, improved, add_self_loops, normalize):
        lib = _lib.get_lib()
        check_tensor(lib, x, "x")
        check_tensor(lib, W, "W")
        N, Fi = x.shape
        if W.dim() != 2 or W.size(0) != Fi:
            raise ValueError(f"W must be [{Fi}, out], got {tuple(W.shape)}")
        if N != edges.N:
            raise ValueError(f"x has {N} rows, the graph {edges.N} nodes")
        Fo = W.size(1)
        if x.stride(1) != 1 or (N > 1 and x.stride(0) < Fi):
            x = x.contiguous()
        W = W.contiguous()
        out = torch.empty(N, Fo, dtype=F32, device=x.device)
        coef = torch.empty(edges.E + N, dtype=F32, device=x.device)
        lib.call("pgt_gcn_small_f32", ptr(edges.ei), ptr(edges.ew), edges.E, N, int(bool(improved)), int(bool(add_self_loops)),
                 int(bool(normalize)), ptr(x), x.stride(0) if N > 1 else Fi, ptr(W), Fi, Fo, ptr(out), ptr(coef),
                 ptr(edges.info), stream_of(lib, x))
        ctx.save_for_backward(x, W, coef)
        ctx.edges = edges
        ctx.flags = (int(bool(add_self_loops)), int(bool(normalize)))
        return out

    @staticmethod
    def backward(ctx, G):
        lib = _lib.get_lib()
        x, W, coef = ctx.saved_tensors
        edges = ctx.edges
        N, Fi = x.shape
        Fo = W.size(1)
        if G.stride(1) != 1 or (N > 1 and G.stride(0) < Fo):
            G = G.contiguous()
        dW = torch.empty(Fi, Fo, dtype=F32, device=x.device)
        dX = torch.empty(N, Fi, dtype=F32, device=x.device) if ctx.needs_input_grad[0] else None
        lib.call("pgt_gcn_small_bwd_f32", ptr(edges.ei), ptr(coef), edges.E, N, ctx.flags[0], ctx.flags[1], ptr(G),
                 G.stride(0) if N > 1 else Fo, ptr(x), x.stride(0) if N > 1 else Fi, ptr(W), Fi, Fo, ptr(dW), ptr(dX), Fi,
                 stream_of(lib, G))
        return dX, dW, None, None, None, None


def gcn_small(x, W, edge_index, edge_weight, improved=False, add_self_loops=True, normalize=True):
    return GcnSmallFunction.apply(x, W, small_edges(edge_index, edge_weight, x.size(0)), improved, add_self_loops, normalize)


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


# --------------------------------------------------------------------------------------------- kernel timer

class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on (torch's
    current stream is the stream handed to the C ABI).  Used by bench.py for the live roofline figures; disabled
    (None) on the timed path.

    An event pair brackets the host call that issues the kernel, so a host stall between the first record and the launch (the
    allocator, the interpreter's collector, a descheduled process) lands in that launch's time.  One such stall of 72 ms made a
    119 us product read 3 012 us on average over 24 launches (round 5, profiles/r05f_bench_full.json).  A launch that took more
    than STALL_FACTOR times the MEDIAN of the launches of the same shape is therefore set aside and REPORTED
    (`set_aside_launches`, `set_aside_ms`), never silently dropped; averages are over the others."""
    STALL_FACTOR = 10.0

    def __init__(self):
        self.records = {}   # kind -> list of (start_event, end_event, work, group) ; work = algorithmic bytes or flops
        self.tagged = {}

    def launch(self, kind, work, fn, tag=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        group = (kind,) + tuple(tag) if tag is not None else (kind, work)
        rec = (e0, e1, work, group)
        self.records.setdefault(kind, []).append(rec)
        if tag is not None:
            self.tagged.setdefault(group, []).append(rec)

    def _times(self):
        """{id(record): (ms, kept)} with the per-shape stall rule applied."""
        torch.cuda.synchronize()
        groups = {}
        for recs in self.records.values():
            for r in recs:
                groups.setdefault(r[3], []).append(r)
        out = {}
        for recs in groups.values():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            med = sorted(ms)[len(ms) // 2]
            for r, t in zip(recs, ms):
                out[id(r)] = (t, t <= self.STALL_FACTOR * med or len(ms) < 3)
        return out

    @staticmethod
    def _stats(recs, times):
        kept = [times[id(r)][0] for r in recs if times[id(r)][1]]
        aside = [times[id(r)][0] for r in recs if not times[id(r)][1]]
        d = {"launches": len(kept), "avg_us": 1e3 * sum(kept) / max(len(kept), 1), "total_ms": sum(kept)}
        if aside:
            d["set_aside_launches"] = len(aside)
            d["set_aside_ms"] = sum(aside)
        return d

    def by_tag(self):
        """Per (kind, shape...) mean launch time; the shape tags are the C-ABI size arguments."""
        times = self._times()
        out = []
        for tag, recs in self.tagged.items():
            out.append({"tag": list(tag), **self._stats(recs, times), "work_per_launch": recs[0][2]})
        return sorted(out, key=lambda r: -r["total_ms"])

    def summary(self):
        times = self._times()
        out = {}
        for kind, recs in self.records.items():
            kept = [r for r in recs if times[id(r)][1]]
            out[kind] = {**self._stats(recs, times), "work_per_launch": sum(r[2] for r in kept) / max(len(kept), 1)}
        return out


KERNEL_TIMER = None


def _timed(kind, work, fn, tag=None):
    if KERNEL_TIMER is None:
        fn()
    else:
        KERNEL_TIMER.launch(kind, work, fn, tag)


def spmm_algorithmic_bytes(n_rows, nnz, F, with_t):
    """int32 rowptr + int32 col + fp32 val + read X once + write Y once (+ read T)."""
    return 4 * (n_rows + 1) + 8 * nnz + 4 * n_rows * F * (3 if with_t else 2)


# --------------------------------------------------------------------------------------------- raw kernel calls

def _rows(t, name):
    """(pointer, row stride) of a 2-D view with unit column stride."""
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise ValueError(f"{name} must be 2-D with unit column stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return ptr(t), (t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0)))


def spmm(csr, X, Y, T=None, alpha=1.0, beta=0.0, ellw=None):
    """Y = alpha * A @ X + beta * T on [n_rows, F] views.  F = 64 on a locality-ordered operator runs the ELLW
    LDS-window kernel (pgt_spmm_ellw_f32), everything else the CSR kernels (pgt_spmm_csr_f32); `ellw` = False / True
    overrides the choice (True: build the layout with the operator's measured halo, or +-32 when it has none)."""
    lib = _lib.get_lib()
    for t, n in ((X, "X"), (Y, "Y")) + (((T, "T"),) if T is not None else ()):
        check_tensor(lib, t, n)
    if X.size(0) != csr.n_rows or Y.shape != X.shape or (T is not None and T.shape != X.shape):
        raise ValueError(f"spmm shape mismatch: rows {csr.n_rows}, X {tuple(X.shape)}, Y {tuple(Y.shape)}")
    xp, ldx = _rows(X, "X")
    yp, ldy = _rows(Y, "Y")
    tp, ldt = _rows(T, "T") if T is not None else (ptr(None), 0)
    st = stream_of(lib, X)
    op = None
    if X.size(1) % 64 == 0 and X.size(1) > 0 and (ellw if ellw is not None else USE_ELLW):
        if ellw and not getattr(csr, "ellw", None):
            _force_ellw(csr)
        op = ellw_of(csr)
        if op is not None and (op.order is not None or op.left_out) and not (
                _window_kernel_covers(X, Y, T) and (csr.n_rows + 456) * max(ldx, ldy, ldt) < 2 ** 31):
            op = None                  # (the C entry point's own fallback is the CSR row tiles on ALL rows: not for these two)
    work = spmm_algorithmic_bytes(csr.n_rows, csr.col.numel(), X.size(1), T is not None) if KERNEL_TIMER else 0
    if op is not None:
        es = op.struct()
        lay = op.csr or csr
        hubs = csr.long_rows if op.left_out else None

        def window_then_hubs():
            lib.call("pgt_spmm_ellw_f32", ctypes.byref(es), ptr(lay.rowptr), ptr(lay.col), ptr(lay.val), csr.n_rows, xp, ldx,
                     yp, ldy, tp, ldt, float(alpha), float(beta), X.size(1), st)
            if hubs is not None:       # the rows the layout leaves out: one workgroup each (the window kernel did not touch them)
                lib.call("pgt_spmm_csr_rows_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.n_rows, ptr(hubs), hubs.numel(),
                         xp, ldx, yp, ldy, tp, ldt, float(alpha), float(beta), X.size(1), st)
        _timed("spmm", work, window_then_hubs)
        return Y
    lr = getattr(csr, "long_rows", None)
    if lr is not None:       # hubs: the row tiles skip them, one workgroup per long row produces them
        _timed("spmm", work, lambda: lib.call(
            "pgt_spmm_csr_long_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.n_rows, ptr(lr), lr.numel(),
            LONG_ROW, xp, ldx, yp, ldy, tp, ldt, float(alpha), float(beta), X.size(1), st))
        return Y
    _timed("spmm", work, lambda: lib.call(
        "pgt_spmm_csr_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.n_rows, xp, ldx, yp, ldy, tp, ldt,
        float(alpha), float(beta), X.size(1), st))
    return Y


def _force_ellw(csr, halo=None):
    """Build the ELLW layout of `csr` regardless of its size / measured locality (tests, A/B runs): correct for any
    operator with rows of at most 32 slots — out-of-window slots are served through the CSR arrays."""
    if getattr(csr, "max_len", -1) < 0 or getattr(csr, "nnz", -1) < 0:
        rp = csr.rowptr[:csr.n_rows + 1]
        csr.nnz = int(rp[csr.n_rows])
        csr.max_len = int((rp[1:] - rp[:-1]).max()) if csr.n_rows else 0
    plan_len = csr.short_len if getattr(csr, "long_rows", None) is not None else csr.max_len     # hubs are left out of the layout
    if plan_len > 32 or csr.nnz <= 0:
        return None
    csr.ellw = Ellw(csr, halo or csr.halo or 32)
    return csr.ellw


def _force_renumbered(csr):
    """Build the renumbered ELLW layout of `csr` regardless of its size / whether it pays (tests, A/B runs)."""
    if getattr(csr, "max_len", -1) < 0 or getattr(csr, "nnz", -1) < 0:
        rp = csr.rowptr[:csr.n_rows + 1]
        csr.nnz = int(rp[csr.n_rows])
        csr.max_len = int((rp[1:] - rp[:-1]).max()) if csr.n_rows else 0
    if csr.max_len > 32 or csr.nnz <= 0:
        return None
    csr.ellw = RenumberedEllw(csr)
    return csr.ellw


def gemm(A, lda, a_seg_stride, n_seg, seg_k, Bw, sbk, sbn, C, ldc, c_seg_stride, c_seg_n, bias, M, N,
         accumulate=False):
    """pgt_gemm_f32 on raw (tensor-as-base-pointer, strides) operands; see include/pgt_hip.h."""
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (Bw, "Bw"), (C, "C")):
        check_tensor(lib, t, n)
    if bias is not None:
        check_tensor(lib, bias, "bias")
    st = stream_of(lib, C)
    _timed("gemm", 2.0 * M * N * n_seg * seg_k, lambda: lib.call(
        "pgt_gemm_f32", ptr(A), lda, a_seg_stride, n_seg, seg_k, ptr(Bw), sbk, sbn, ptr(C), ldc, c_seg_stride,
        c_seg_n, ptr(bias), M, N, int(bool(accumulate)), st),
        tag=("NT" if (sbk == 1 and sbn != 1) else "NN", M, N, n_seg, seg_k, c_seg_n, int(bool(accumulate))))
    return C


# DCRNN backward: skip the input columns of the stack gradient when the input needs no gradient (A/B: PGT_SKIP_X=0)
SKIP_INPUT_COLUMNS_WHEN_UNUSED = os.environ.get("PGT_SKIP_X", "1") != "0"
# DCRNN cell forward: sigmoid / H*R and tanh / blend inside the gate GEMMs' epilogues (A/B: PGT_FUSE_GATES=0)
FUSE_GATE_EPILOGUES = os.environ.get("PGT_FUSE_GATES", "1") != "0"
# feature-gradient GEMM of the hidden columns (S*O = 320 output columns): 1 = a 256-column product on the persistent
# deferred-store kernel + a 64-column remainder, 0 = one 320-column product (three 128-wide column tiles, the last masked)
SPLIT_FEATURE_GRADIENT = os.environ.get("PGT_SPLIT_FG", "1") != "0"
ONE_FEATURE_GRADIENT = os.environ.get("PGT_ONE_FG", "1") != "0"
# DCRNN backward: the gate stack's d/dH joins the running state gradient inside the next gate-backward kernel (A/B: PGT_FOLD_DH=0
# = a separate accumulation pass per time step)
FOLD_STATE_GRADIENT = os.environ.get("PGT_FOLD_DH", "1") != "0"
ONE_FEATURE_GRADIENT_MIN_ROWS = int(os.environ.get("PGT_ONE_FG_MIN_ROWS", "8192"))   # tests lower it to drive the 320-column product at small sizes
# weight / bias gradients without float atomics (pgt_gemm_tn_det_f32): per-slab partial sums + one pass that adds them in a fixed
# order — bitwise reproducible run to run, like the reference's CPU path.  The default since round 6 (the second pass takes 15 us
# per product, the products themselves the same time either way: 2.32 against 2.2 - 2.4 ms per training step at the benchmark
# shape); PGT_DETERMINISTIC=0 or ops.DETERMINISTIC_WEIGHT_GRADIENTS = False returns to fp32 atomics into dW.
DETERMINISTIC_WEIGHT_GRADIENTS = os.environ.get("PGT_DETERMINISTIC", "1") != "0"


def gemm_gru_zr(A, lda, a_seg_stride, n_seg, seg_k, Bw, sbk, sbn, bias, zr, H, xhr, f_in):
    """pgt_gemm_gru_zr_f32: zr [M, 2O] = sigmoid(A Bw + bias), xhr[:, f_in:] = H * zr[:, O:] (== gemm + _gru_zr)."""
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (Bw, "Bw"), (zr, "zr"), (H, "H"), (xhr, "xhr")):
        check_tensor(lib, t, n)
    M, O2 = zr.shape
    hp, ldh = _rows(H, "H")
    xp, ldx = _rows(xhr, "xhr")
    _timed("gemm", 2.0 * M * O2 * n_seg * seg_k, lambda: lib.call(
        "pgt_gemm_gru_zr_f32", ptr(A), lda, a_seg_stride, n_seg, seg_k, ptr(Bw), sbk, sbn, ptr(bias), ptr(zr), hp, ldh,
        xp, ldx, f_in, M, O2 // 2, stream_of(lib, zr)), tag=("NN+zr", M, O2, n_seg, seg_k, O2, 0))


class RowMap:
    """A [M, W] operand inside a larger tensor (pgt_rowmap): row m at `base` + (m // period) * stride_hi +
    (m % period) * ld floats.  `base` is a tensor view whose data_ptr is row 0; `width` the row length."""
    __slots__ = ("base", "ld", "period", "stride_hi", "width", "_st")

    def __init__(self, base, ld, period, stride_hi, width):
        self.base, self.ld, self.period, self.stride_hi, self.width = base, int(ld), int(period), int(stride_hi), int(width)
        self._st = RowMapStruct(self.period, self.stride_hi)

    def ref(self):
        return ctypes.byref(self._st)


def _rows_or_map(t, name):
    """(pointer, row stride, pgt_rowmap* or NULL) of a plain 2-D view or a RowMap."""
    if isinstance(t, RowMap):
        return ptr(t.base), t.ld, t.ref()
    p, ld = _rows(t, name)
    return p, ld, None


def gemm_gru_h(A, lda, a_seg_stride, n_seg, seg_k, Bw, sbk, sbn, bias, ht, zr, H, out0, out1=None):
    """pgt_gemm_gru_h_f32: ht [M, O] = tanh(A Bw + bias), Hnew = Z H + (1 - Z) ht -> out0 (, out1) (== gemm + _gru_h).
    out0 may be a RowMap (H_t straight into a [B, T, N, O] tensor)."""
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (Bw, "Bw"), (ht, "ht"), (zr, "zr"), (H, "H"), (out0.base if isinstance(out0, RowMap) else out0, "out0")):
        check_tensor(lib, t, n)
    M, O = ht.shape
    hp, ldh = _rows(H, "H")
    op, ld0, m0 = _rows_or_map(out0, "out0")
    o1, ld1 = _rows(out1, "out1") if out1 is not None else (ptr(None), 0)
    _timed("gemm", 2.0 * M * O * n_seg * seg_k, lambda: lib.call(
        "pgt_gemm_gru_h_f32", ptr(A), lda, a_seg_stride, n_seg, seg_k, ptr(Bw), sbk, sbn, ptr(bias), ptr(ht), ptr(zr),
        hp, ldh, op, ld0, m0, o1, ld1, M, O, stream_of(lib, ht)), tag=("NN+h", M, O, n_seg, seg_k, O, 0))


_DET_WS = {}


def _det_workspace(device, n_floats):
    """Scratch of the atomics-free weight gradient, one buffer per device, grown on demand (the entry point sizes it for
    the most slabs any schedule launches; allocating it per call — T times per BPTT — is what the advisor flagged)."""
    key = (device.type, device.index)
    ws = _DET_WS.get(key)
    if ws is None or ws.numel() < n_floats:
        ws = _DET_WS[key] = torch.empty(n_floats, dtype=F32, device=device)
    return ws


def gemm_tn_acc(A, lda, a_seg_stride, n_seg, seg_k, G, ldg, dW, lddw, db, M, N):
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (G, "G"), (dW, "dW")):
        check_tensor(lib, t, n)
    if db is not None:
        check_tensor(lib, db, "db")
    st = stream_of(lib, G)
    if DETERMINISTIC_WEIGHT_GRADIENTS:
        # no float atomics: per-slab partial sums in a scratch buffer, added in slab order (bitwise reproducible)
        nbytes = int(lib._pgt_gemm_tn_det_ws_bytes(n_seg, seg_k, N, lddw))
        ws = _det_workspace(G.device, max(nbytes // 4, 1))
        _timed("gemm_tn", 2.0 * M * N * n_seg * seg_k, lambda: lib.call(
            "pgt_gemm_tn_det_f32", ptr(A), lda, a_seg_stride, n_seg, seg_k, ptr(G), ldg, ptr(dW), lddw, ptr(db), M, N,
            ptr(ws), nbytes, st), tag=(M, N, n_seg, seg_k))
        return dW
    _timed("gemm_tn", 2.0 * M * N * n_seg * seg_k, lambda: lib.call(
        "pgt_gemm_tn_acc_f32", ptr(A), lda, a_seg_stride, n_seg, seg_k, ptr(G), ldg, ptr(dW), lddw, ptr(db), M, N, st),
        tag=(M, N, n_seg, seg_k))
    return dW


def linear_fwd(X2, W_kn, bias, out=None):
    """out[M,N] = X2[M,K] @ W_kn[K,N] + bias (plain matrices; W_kn may be any 2-D strided view)."""
    M, K = X2.shape
    N = W_kn.size(1)
    if out is None:
        out = torch.empty(M, N, dtype=F32, device=X2.device)
    _, lda = _rows(X2, "X2")
    gemm(X2, lda, 0, 1, K, W_kn, W_kn.stride(0), W_kn.stride(1), out, out.stride(0), 0, N, bias, M, N)
    return out


def copy2d(dst, src):
    lib = _lib.get_lib()
    check_tensor(lib, dst, "dst"); check_tensor(lib, src, "src")
    dp, ldd = _rows(dst, "dst")
    sp, lds = _rows(src, "src")
    _timed("mover", 8.0 * src.numel() if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_copy2d_f32", dp, ldd, sp, lds, src.size(0), src.size(1), stream_of(lib, dst)))


def add2d(dst, src):
    lib = _lib.get_lib()
    check_tensor(lib, dst, "dst"); check_tensor(lib, src, "src")
    dp, ldd = _rows(dst, "dst")
    sp, lds = _rows(src, "src")
    _timed("mover", 12.0 * src.numel() if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_add2d_f32", dp, ldd, sp, lds, src.size(0), src.size(1), stream_of(lib, dst)))


def axpby2d(dst, x, a, y=None, b=0.0):
    lib = _lib.get_lib()
    check_tensor(lib, dst, "dst"); check_tensor(lib, x, "x")
    dp, ldd = _rows(dst, "dst")
    xp, ldx = _rows(x, "x")
    yp, ldy = _rows(y, "y") if y is not None else (ptr(None), 0)
    _timed("mover", 4.0 * x.numel() * (2 if y is None else 3) if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_axpby2d_f32", dp, ldd, xp, ldx, float(a), yp, ldy, float(b), x.size(0), x.size(1), stream_of(lib, dst)))


def swap01(src, D0, D1, W):
    """[D0][D1][W] -> [D1][D0][W] (batch-major <-> node-major)."""
    lib = _lib.get_lib()
    check_tensor(lib, src, "src")
    src = src.contiguous()
    dst = torch.empty(D1, D0, W, dtype=F32, device=src.device)
    _timed("mover", 8.0 * src.numel() if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_swap01_f32", ptr(dst), ptr(src), D0, D1, W, stream_of(lib, src)))
    return dst


def window_gather(data, starts, horizon, time_major=False):
    """(X, Y) index-batch windows of a resident series `data` [T, ...] at the int64 start indices `starts` [B]
    (pgt_window_gather_f32): X[b] = data[s_b : s_b + h], Y[b] = data[s_b + h : s_b + 2 h]; shapes [B, h, ...] or, with
    time_major, [h, B, ...].  The caller guarantees 0 <= s_b <= T - 2 h (the start indices live on the device: checking
    them here would be a host sync per batch; IndexDataset.gather validates its index array once) — the kernel clamps rows
    that fall outside instead of reading past the series."""
    lib = _lib.get_lib()
    check_tensor(lib, data, "data")
    check_tensor(lib, starts, "starts", torch.int64)
    data = data.contiguous()
    starts = starts.contiguous()
    T_total, B, h = data.size(0), starts.numel(), int(horizon)
    W = data[0].numel() if T_total else 0
    lead = (h, B) if time_major else (B, h)
    X = torch.empty(*lead, *data.shape[1:], dtype=F32, device=data.device)
    Y = torch.empty_like(X)
    lib.call("pgt_window_gather_f32", ptr(data), T_total, W, ptr(starts), B, h, ptr(X), ptr(Y), int(bool(time_major)),
             stream_of(lib, data))
    return X, Y


class Swap01(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, D0, D1, W):
        ctx.dims = (D0, D1, W)
        return swap01(x, D0, D1, W)

    @staticmethod
    def backward(ctx, g):
        D0, D1, W = ctx.dims
        return swap01(g, D1, D0, W), None, None, None


# --------------------------------------------------------------------------------------------- diffusion stack

def _stack_fwd(g, TS, t, K, Nn):
    """T_0 = TS[0,t] given; fill T_k^{o,i} (dcrnn.py:85-106): T_1 = P T_0, T_k = 2 P T_{k-1} - T_0 (Tx_0 is never
    advanced in the reference, dcrnn.py:106 — reproduced).  Segment order: [T0, T1o, T1i, T2o, T2i, ...]."""
    T0 = TS[0, t].view(Nn, -1)
    for k in range(1, K):
        for d, csr in enumerate((g.fwd_o, g.fwd_i)):
            src = T0 if k == 1 else TS[2 * (k - 1) - 1 + d, t].view(Nn, -1)
            dst = TS[2 * k - 1 + d, t].view(Nn, -1)
            if k == 1:
                spmm(csr, src, dst)
            else:
                spmm(csr, src, dst, T=T0, alpha=2.0, beta=-1.0)


def fold_backward_weight(Wst, K, C):
    """For K <= 3 the "- Tx_0" terms of the recursion only touch the LAST hop, so their adjoint
    (G_0 -= G_k^o + G_k^i) is linear in dPRE and folds into the segment-0 rows of the weight used by the
    feature-gradient GEMM: G_0 = dPRE (W_0 - sum_{k>=2} W_k^o + W_k^i)^T.  Removes 2(K-2) streaming passes over
    [M, C] per stack.  (K >= 4 keeps the explicit form: there G_k is updated before it is subtracted.)"""
    if K < 3 or K > 3:
        return Wst, False
    Wb = Wst.clone()
    for k in range(2, K):
        for d in range(2):
            j = 2 * k - 1 + d
            Wb[0:C] -= Wst[j * C:(j + 1) * C]
    return Wb, True


def _stack_bwd(g, G, K, Nn, folded=False):
    """Adjoint of _stack_fwd on G [S][M][C] (in place); on exit G[0] holds d/dT_0.  `folded`: the G_0 -= G_k terms
    were already applied through fold_backward_weight."""
    G0 = G[0].view(Nn, -1)
    for k in range(K - 1, 1, -1):
        for d, csr in enumerate((g.bwd_o, g.bwd_i)):
            Gk = G[2 * k - 1 + d].view(Nn, -1)
            Gp = G[2 * (k - 1) - 1 + d].view(Nn, -1)
            spmm(csr, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
            if not folded:
                axpby2d(G0, Gk, -1.0, G0, 1.0)
    if K > 1:
        for d, csr in enumerate((g.bwd_o, g.bwd_i)):
            spmm(csr, G[1 + d].view(Nn, -1), G0, T=G0, alpha=1.0, beta=1.0)


def slab_fits(g, C, K):
    """True when the LDS-resident one-launch diffusion stack (pgt_dconv_stack_slab_f32) covers this shape."""
    if K < 2:
        return False
    lib = _lib.get_lib()
    return bool(lib._pgt_dconv_stack_slab_fits(g.N, int(C), int(K), g.E, g.E))


def slab_plan(g, C, K, n_samples=1 << 20):
    """(column windows per sample, workgroups per CU, threads, tasks per thread) of the stack launch for this shape and batch
    (pgt_dconv_stack_slab_plan); (1, 1, 1024, 0) = the whole-sample kernels, zeros = not supported."""
    lib = _lib.get_lib()
    out = (ctypes.c_int32 * 4)()
    lib.call("pgt_dconv_stack_slab_plan", g.N, int(n_samples), int(C), int(K), g.E, g.E, out)
    return tuple(out)


def _slab_fwd(g, TS0, seg_stride, n_samples, C, K):
    """TS0: the [n_samples*N, C] block of segment 0 (batch-major rows); the other segments follow at seg_stride."""
    lib = _lib.get_lib()
    so, si = g.fwd_o.struct(), g.fwd_i.struct()
    work = (5 if K >= 3 else 3) * 4 * TS0.numel() if KERNEL_TIMER else 0
    _timed("stack", work, lambda: lib.call(
        "pgt_dconv_stack_slab_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, g.N, n_samples, C, K, ptr(TS0),
        seg_stride, stream_of(lib, TS0)))


def _slab_bwd(g, G0, seg_stride, n_samples, C, K, folded):
    lib = _lib.get_lib()
    so, si = g.bwd_o.struct(), g.bwd_i.struct()
    work = (6 if K >= 3 else 4) * 4 * G0.numel() if KERNEL_TIMER else 0
    _timed("stack", work, lambda: lib.call(
        "pgt_dconv_stack_slab_bwd_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, g.N, n_samples, C, K, ptr(G0),
        seg_stride, int(bool(folded)), stream_of(lib, G0)))


class _StackWeight(torch.autograd.Function):
    """DConv weight [2,K,C,O] -> [(2K-1)*C, O] with three copies forward and three backward (torch's slice / cat graph
    of the same rearrangement costs ~30 tiny launches per weight and backward pass)."""

    @staticmethod
    def forward(ctx, weight):
        _, K, C, O = weight.shape
        out = torch.empty((2 * K - 1) * C, O, dtype=weight.dtype, device=weight.device)
        torch.add(weight[0, 0], weight[1, 0], out=out[:C])
        if K > 1:
            out[C:].view(K - 1, 2, C, O).copy_(weight[:, 1:].permute(1, 0, 2, 3))
        ctx.shape = weight.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        _, K, C, O = ctx.shape
        dw = torch.empty(ctx.shape, dtype=dout.dtype, device=dout.device)
        dw[0, 0].copy_(dout[:C])
        dw[1, 0].copy_(dout[:C])
        if K > 1:
            dw[:, 1:].copy_(dout[C:].reshape(K - 1, 2, C, O).permute(1, 0, 2, 3))
        return dw


def stack_weight(weight):
    """DConv weight [2,K,C,O] -> stacked [(2K-1)*C, O] matching the segment order of _stack_fwd.
    Segment 0 carries W[0,0] + W[1,0] (the reference computes X@W[0,0] + X@W[1,0], dcrnn.py:81-83)."""
    return _StackWeight.apply(weight)


class CellWeightsFunction(torch.autograd.Function):
    """conv_x_{z,r,h}.weight [2, K, C, O] (+ the z / r biases) -> (Wzr [(2K-1) C, 2O], bzr [2O] | None, Wh [(2K-1) C, O]):
    the stacked operands of the two gate products, one launch forward and one backward (pgt_dcrnn_pack_weights_f32)."""

    @staticmethod
    def forward(ctx, Wz, Wr, Wh, bz, br):
        lib = _lib.get_lib()
        for t, n in ((Wz, "conv_x_z.weight"), (Wr, "conv_x_r.weight"), (Wh, "conv_x_h.weight")):
            check_tensor(lib, t, n)
        _, K, C, O = Wz.shape
        if Wr.shape != Wz.shape or Wh.shape != Wz.shape or (bz is None) != (br is None):
            raise ValueError("CellWeightsFunction: the three convolutions must have one shape and agree on bias")
        dev = Wz.device
        S = 2 * K - 1
        Wzr = torch.empty(S * C, 2 * O, dtype=F32, device=dev)
        Whs = torch.empty(S * C, O, dtype=F32, device=dev)
        bzr = torch.empty(2 * O, dtype=F32, device=dev) if bz is not None else None
        CellWeightsFunction.repack((Wz, Wr, Wh, bz, br), (Wzr, bzr, Whs))
        ctx.dims = (K, C, O)
        ctx.has_bias = bz is not None
        if bzr is None:
            return Wzr, None, Whs
        return Wzr, bzr, Whs

    @staticmethod
    def repack(params, packed):
        """The pack launch alone, into operands that already exist (nn/_states.py packed_once refreshes cached operands with it)."""
        lib = _lib.get_lib()
        Wz, Wr, Wh, bz, br = params
        Wzr, bzr, Whs = packed
        _, K, C, O = Wz.shape
        Wzc, Wrc, Whc = Wz.contiguous(), Wr.contiguous(), Wh.contiguous()
        lib.call("pgt_dcrnn_pack_weights_f32", ptr(Wzc), ptr(Wrc), ptr(Whc), ptr(bz.contiguous() if bz is not None else None),
                 ptr(br.contiguous() if br is not None else None), K, C, O, ptr(Wzr), ptr(bzr), ptr(Whs), stream_of(lib, Wzr))

    @staticmethod
    def backward(ctx, dWzr, dbzr, dWhs):
        lib = _lib.get_lib()
        K, C, O = ctx.dims
        ref = dWzr if dWzr is not None else dWhs
        if ref is None:
            return None, None, None, None, None
        dev = ref.device
        dWz, dWr, dWh = (torch.empty(2, K, C, O, dtype=F32, device=dev) for _ in range(3))
        dbz = dbr = None
        if ctx.has_bias:
            dbz, dbr = torch.empty(O, dtype=F32, device=dev), torch.empty(O, dtype=F32, device=dev)
            if dbzr is None:
                dbzr = torch.zeros(2 * O, dtype=F32, device=dev)
        lib.call("pgt_dcrnn_unpack_weight_grads_f32", ptr(dWzr.contiguous() if dWzr is not None else None),
                 ptr(dbzr.contiguous() if (ctx.has_bias and dbzr is not None) else None),
                 ptr(dWhs.contiguous() if dWhs is not None else None), K, C, O, ptr(dWz), ptr(dWr), ptr(dWh), ptr(dbz), ptr(dbr),
                 stream_of(lib, dWz))
        return dWz, dWr, dWh, dbz, dbr


def cell_k1_fits(N, Fin, O):
    """Whether the one-launch K = 1 cell (csrc/small_cell.hip) takes this shape (else: the general path)."""
    return bool(_lib.get_lib()._pgt_dcrnn_cell_k1_fits(int(N), int(Fin), int(O)))


class DCRNNCellK1Function(torch.autograd.Function):
    """DCRNN(in, out, K = 1) cell step, one launch forward and one backward (pgt_dcrnn_cell_k1_f32; dcrnn.py:79-82 +
    172-192): X [N, in], H [N, out] | None, the three convolutions' parameters as they are ([2, 1, in + out, out], [out])."""

    @staticmethod
    def forward(ctx, X, H, Wz, Wr, Wh, bz, br, bh):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        N, Fin = X.shape
        O = Wz.shape[3]
        if Wz.shape != (2, 1, Fin + O, O) or Wr.shape != Wz.shape or Wh.shape != Wz.shape:
            raise ValueError(f"DCRNN cell (K = 1): weights must be [2, 1, {Fin + O}, {O}], got {tuple(Wz.shape)}")
        if X.stride(1) != 1:
            X = X.contiguous()
        if H is not None:
            check_tensor(lib, H, "H")
            if H.shape != (N, O):
                raise ValueError(f"H must be [{N}, {O}], got {tuple(H.shape)}")
            if H.stride(1) != 1:
                H = H.contiguous()
        Wz, Wr, Wh = Wz.contiguous(), Wr.contiguous(), Wh.contiguous()
        out = torch.empty(N, O, dtype=F32, device=X.device)
        saved = torch.empty(N, 3 * O, dtype=F32, device=X.device)
        lib.call("pgt_dcrnn_cell_k1_f32", ptr(X), X.stride(0) if N > 1 else Fin, ptr(H),
                 (H.stride(0) if N > 1 else O) if H is not None else 0, ptr(Wz), ptr(Wr), ptr(Wh), ptr(bz), ptr(br), ptr(bh),
                 ptr(out), O, ptr(saved), N, Fin, O, stream_of(lib, X))
        ctx.save_for_backward(X, H, Wz, Wr, Wh, saved)
        ctx.has_bias = (bz is not None, br is not None, bh is not None)
        return out

    @staticmethod
    def backward(ctx, G):
        lib = _lib.get_lib()
        X, H, Wz, Wr, Wh, saved = ctx.saved_tensors
        N, Fin = X.shape
        O = Wz.shape[3]
        C = Fin + O
        if G.stride(1) != 1 or (N > 1 and G.stride(0) < O):
            G = G.contiguous()
        need = ctx.needs_input_grad
        dev = X.device
        # one allocation: the three weight gradients, the three bias gradients, the kernel's scratch
        nW = 2 * C * O
        buf = torch.empty(3 * nW + 3 * O + N * 3 * O, dtype=F32, device=dev)
        dWz, dWr, dWh = (buf[i * nW:(i + 1) * nW].view(2, 1, C, O) for i in range(3))
        dbz, dbr, dbh = (buf[3 * nW + i * O:3 * nW + (i + 1) * O] if ctx.has_bias[i] else None for i in range(3))
        dP = buf[3 * nW + 3 * O:]
        dX = torch.empty(N, Fin, dtype=F32, device=dev) if need[0] else None
        dH = torch.empty(N, O, dtype=F32, device=dev) if (H is not None and need[1]) else None
        lib.call("pgt_dcrnn_cell_k1_bwd_f32", ptr(G), G.stride(0) if N > 1 else O, ptr(X), X.stride(0) if N > 1 else Fin,
                 ptr(H), (H.stride(0) if N > 1 else O) if H is not None else 0, ptr(Wz), ptr(Wr), ptr(Wh), ptr(saved),
                 ptr(dX), Fin, ptr(dH), O, ptr(dWz), ptr(dWr), ptr(dWh), ptr(dbz), ptr(dbr), ptr(dbh), ptr(dP), N, Fin, O,
                 stream_of(lib, G))
        return dX, dH, dWz, dWr, dWh, dbz, dbr, dbh


class DConvFunction(torch.autograd.Function):
    """H = DConv(X) for node-major X [N*B, C]: diffusion stack (SpMM) + one segmented MFMA GEMM."""

    @staticmethod
    def forward(ctx, X, Wst, bias, g, K, B):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        M, C = X.shape
        Nn = g.N
        if M != Nn * B:
            raise ValueError(f"X has {M} rows, expected num_nodes*B = {Nn * B}")
        S = 2 * K - 1
        O = Wst.size(1)
        TS = torch.empty(S, 1, M, C, dtype=F32, device=X.device)
        copy2d(TS[0, 0], X)
        slab = B == 1 and slab_fits(g, C, K)      # one sample: batch-major == node-major
        if slab:
            _slab_fwd(g, TS[0, 0], M * C, 1, C, K)
        else:
            _stack_fwd(g, TS, 0, K, Nn)
        Wc = Wst.contiguous()
        out = torch.empty(M, O, dtype=F32, device=X.device)
        gemm(TS, C, M * C, S, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K, ctx.B, ctx.slab = g, K, B, slab
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc)
        return out

    @staticmethod
    def backward(ctx, dH):
        TS, Wc = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        S, _, M, C = TS.shape
        O = Wc.size(1)
        dH = dH.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros_like(Wc)
            db = torch.zeros(O, dtype=F32, device=dH.device) if ctx.has_bias else None
            gemm_tn_acc(TS, C, M * C, S, C, dH, O, dW, O, db, M, O)
        if ctx.needs_input_grad[0]:
            G = torch.empty(S, M, C, dtype=F32, device=dH.device)
            Wb, folded = fold_backward_weight(Wc, K, C)
            gemm(dH, O, 0, 1, O, Wb, 1, O, G, C, M * C, C, None, M, S * C)
            if ctx.slab:
                _slab_bwd(g, G[0], M * C, 1, C, K, folded)
            else:
                _stack_bwd(g, G, K, g.N, folded)
            dX = G[0]
        return dX, dW, db, None, None, None


# --------------------------------------------------------------------------------------------- DCRNN sequence

def _gru_zr(pre_zr, H, xhr, f_in):
    lib = _lib.get_lib()
    M, O2 = pre_zr.shape
    hp, ldh = _rows(H, "H")
    xp, ldx = _rows(xhr, "xhr")
    _timed("gate", 12.0 * M * O2 if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_gru_zr_f32", ptr(pre_zr), hp, ldh, xp, ldx, f_in, M, O2 // 2, stream_of(lib, pre_zr)))


def _gru_h(pre_h, zr, H, out0, out1=None):
    lib = _lib.get_lib()
    M, O = pre_h.shape
    hp, ldh = _rows(H, "H")
    op, ld0, m0 = _rows_or_map(out0, "out0")
    o1, ld1 = _rows(out1, "out1") if out1 is not None else (ptr(None), 0)
    _timed("gate", 4.0 * M * O * (5 if out1 is None else 6) if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_gru_h_f32", ptr(pre_h), ptr(zr), hp, ldh, op, ld0, m0, o1, ld1, M, O, stream_of(lib, pre_h)))


def _gru_h_bwd(dHn, zr, H, ht, d_pre_h, d_pre_zr, dH, accumulate, dHn2=None, dHn3=None):
    """dHn and H may be RowMaps (read in place from [B, T, N, O] tensors); dHn2 / dHn3: further addends of d/dH'."""
    lib = _lib.get_lib()
    M, O = ht.shape
    gp, ldg, mg = _rows_or_map(dHn, "dHn")
    g2, ldg2 = _rows(dHn2, "dHn2") if dHn2 is not None else (ptr(None), 0)
    g3, ldg3 = _rows(dHn3, "dHn3") if dHn3 is not None else (ptr(None), 0)
    hp, ldh, mh = _rows_or_map(H, "H")
    dp, ldd = _rows(dH, "dH")
    # algorithmic bytes: dH' (+ its further addends), z, H, tanh in; d_pre_h, d_pre_z, dH out (+ dH in when accumulating)
    work = 4.0 * M * O * (7 + (dHn2 is not None) + (dHn3 is not None) + bool(accumulate)) if KERNEL_TIMER else 0
    _timed("gate_bwd", work, lambda: lib.call(
        "pgt_gru_h_bwd_f32", gp, ldg, mg, g2, ldg2, g3, ldg3, ptr(zr), hp, ldh, mh, ptr(ht), ptr(d_pre_h), ptr(d_pre_zr), dp,
        ldd, int(bool(accumulate)), M, O, stream_of(lib, ht)), tag=("h", M, O))


def _gru_zr_bwd(dxhr, f_in, zr, H, d_pre_zr, dH):
    lib = _lib.get_lib()
    M, O2 = zr.shape
    xp, ldx = _rows(dxhr, "dxhr")
    hp, ldh, mh = _rows_or_map(H, "H")
    dp, ldd = _rows(dH, "dH")
    # algorithmic bytes: d(H R), r, H, dH in; d_pre_r, dH out
    _timed("gate_bwd", 12.0 * M * O2 if KERNEL_TIMER else 0, lambda: lib.call(
        "pgt_gru_zr_bwd_f32", xp, ldx, f_in, ptr(zr), hp, ldh, mh, ptr(d_pre_zr), dp, ldd, M, O2 // 2,
        stream_of(lib, zr)), tag=("zr", M, O2 // 2))


class _StateLayout:
    """The [M, O] slice of time step t inside a contiguous [B, T, N, O] tensor, as RowMaps: batch-major rows
    m = b*N + n -> period N, stride_hi T*N*O, ld O; node-major rows m = n*B + b -> period B, stride_hi O, ld T*N*O."""

    def __init__(self, tensor, T, N, B, O, batch_major):
        if tensor.shape != (B, T, N, O) or not tensor.is_contiguous():
            raise ValueError(f"expected a contiguous [B, T, N, O] = {(B, T, N, O)} tensor, got {tuple(tensor.shape)}")
        self.flat, self.step_stride, self.O = tensor.view(-1), N * O, O
        self.args = (O, N, T * N * O) if batch_major else (T * N * O, B, O)      # (ld, period, stride_hi)

    def step(self, t):
        ld, period, hi = self.args
        return RowMap(self.flat[t * self.step_stride:], ld, period, hi, self.O)


# widest hidden state the one-workgroup-per-sample sequence kernels are used for (beyond it the in-kernel scalar products lose to
# the MFMA path even where the state would still fit the LDS); PGT_SEQ_SMALL=0 turns them off (A/B)
# T-GCN cell at hidden width 32: the fused one-launch forward / adjoint (csrc/tgcn_cell.hip); PGT_TGCN_FUSED=0 = the two
# fused-epilogue products + gate kernels (A/B, and the path of every other width)
USE_TGCN_FUSED = os.environ.get("PGT_TGCN_FUSED", "1") != "0"
SEQ_SMALL_MAX_O = int(os.environ.get("PGT_SEQ_SMALL_MAX_O", "8"))
USE_SEQ_SMALL = os.environ.get("PGT_SEQ_SMALL", "1") != "0"


def seq_small_fits(g, Fin, O, K):
    """Whether pgt_dcrnn_seq_small_f32 takes this graph / width (per-sample state within a workgroup's LDS)."""
    return bool(_lib.get_lib()._pgt_dcrnn_seq_small_fits(g.N, g.E, g.E, int(Fin), int(O), int(K)))


class DCRNNSeqSmallFunction(torch.autograd.Function):
    """BatchedDCRNN.forward / a DCRNN cell step for small graphs and narrow states: the whole sequence of a sample in one
    workgroup, one launch forward and one backward (csrc/seq_small.hip).  X [B, T, N, Fin], H0 [B, N, O] | None ->
    [B, T, N, O] (the reference's layout); Wzr / bzr / Wh / bh = the stacked operands of CellWeightsFunction."""

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        if X.dim() != 4:
            raise ValueError(f"DCRNNSeqSmallFunction: X must be [B, T, N, in], got {tuple(X.shape)}")
        B, T, N, Fin = X.shape
        O = Wh.size(1)
        S, C = 2 * K - 1, Fin + O
        # the C entry point only null-checks its pointers: a wrong width here would be an out-of-bounds device read
        if N != g.N:
            raise ValueError(f"X has {N} nodes, the graph {g.N}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O):
            raise ValueError(f"DCRNNSeqSmallFunction: inconsistent operand shapes: X has {Fin} input channels, the stacked weights "
                             f"{tuple(Wzr.shape)} / {tuple(Wh.shape)} expect in + out = {Wzr.size(0) // S if S else 0} with out = {O} "
                             f"(K = {K})")
        if (bzr is not None and bzr.shape != (2 * O,)) or (bh is not None and bh.shape != (O,)):
            raise ValueError("DCRNNSeqSmallFunction: inconsistent bias shapes")
        Xc = X.contiguous()
        H0c = None
        if H0 is not None:
            check_tensor(lib, H0, "H")
            if H0.shape != (B, N, O):
                raise ValueError(f"H must be {(B, N, O) if B > 1 else (N, O)}, got {tuple(H0.shape[1:] if B == 1 and H0.dim() == 3 else H0.shape)}")
            H0c = H0.contiguous()
        dev = X.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        out = torch.empty(B, T, N, O, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad)
        save = None
        if need:
            per = int(lib._pgt_dcrnn_seq_small_save_floats(N, Fin, O, K))
            save = torch.empty(B * T * per, dtype=F32, device=dev)
        so, si = g.fwd_o.struct(), g.fwd_i.struct()
        _timed("seq_small", 4.0 * (Xc.numel() + out.numel() + (save.numel() if need else 0)) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_dcrnn_seq_small_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, N, ptr(Xc), T * N * Fin, N * Fin, ptr(H0c),
            ptr(Wzr_c), ptr(bzr), ptr(Wh_c), ptr(bh), B, T, Fin, O, K, ptr(out), T * N * O, N * O, ptr(save), stream_of(lib, out)))
        ctx.g, ctx.K = g, K
        ctx.has = (H0 is not None, bzr is not None, bh is not None)
        if need:
            ctx.save_for_backward(out, H0c, save, Wzr_c, Wh_c)
        ctx.dims = (B, T, N, Fin, O)
        return out

    @staticmethod
    def backward(ctx, dOut):
        lib = _lib.get_lib()
        out, H0c, save, Wzr_c, Wh_c = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        B, T, N, Fin, O = ctx.dims
        C, S = Fin + O, 2 * K - 1
        dev = out.device
        dOc = dOut.contiguous()
        need = ctx.needs_input_grad
        dX = torch.empty(B, T, N, Fin, dtype=F32, device=dev) if need[0] and Fin > 0 else None
        dH0 = torch.empty(B, N, O, dtype=F32, device=dev) if (ctx.has[0] and need[1]) else None
        nW = S * C * 3 * O + 3 * O
        part = torch.zeros(B, nW, dtype=F32, device=dev)
        to, ti = g.bwd_o.struct(), g.bwd_i.struct()
        _timed("seq_small", 4.0 * (dOc.numel() + out.numel() + save.numel()) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_dcrnn_seq_small_bwd_f32", ctypes.byref(to), ctypes.byref(ti), g.E, g.E, N, ptr(dOc), T * N * O, N * O, ptr(out),
            T * N * O, N * O, ptr(H0c), ptr(save), ptr(Wzr_c), ptr(Wh_c), B, T, Fin, O, K, ptr(dX), T * N * Fin, N * Fin, ptr(dH0),
            ptr(part), stream_of(lib, out)))
        dW = part.sum(dim=0) if B > 1 else part[0]            # the samples in index order: deterministic
        n1, n2 = S * C * 2 * O, S * C * 3 * O
        dWzr, dWh = dW[:n1].view(S * C, 2 * O), dW[n1:n2].view(S * C, O)
        dbzr = dW[n2:n2 + 2 * O] if ctx.has[1] else None
        dbh = dW[n2 + 2 * O:] if ctx.has[2] else None
        return dX, dH0, dWzr, dbzr, dWh, dbh, None, None


class DCRNNSeqFunction(torch.autograd.Function):
    """T steps of the DCRNN GRU cell (dcrnn.py:172-219 / :406-475) on node-major rows m = n*B + b.

    X [T, M, F_in], H0 [M, O] -> Hall [T, M, O].  Per step: [X_t, H] -> diffusion stack -> one MFMA GEMM for the
    update and reset gates together (the reference aggregates [X,H] twice) -> sigmoid / H*R -> stack -> GEMM ->
    tanh / blend.  Backward is hand-written BPTT on the transposed operators; the weight gradients of all T steps
    are one split-K GEMM over the saved stacks.
    """

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K, B, batch_major=False, btno=False):
        """batch_major: rows m = b*N + n and the diffusion stacks run as ONE LDS-resident launch per conv
        (pgt_dconv_stack_slab_f32; requires slab_fits); otherwise rows m = n*B + b and one launch per hop.
        btno: the hidden states are returned as the reference returns them, a contiguous [B, T, N, O] tensor
        (torch.stack(outputs, dim=1), dcrnn.py:463-475): the candidate-gate epilogue of step t stores H_t straight into
        out[:, t] through a two-level row map (pgt_rowmap) and the backward pass reads the incoming gradient and the
        previous states in place from that layout — no transposition pass either way."""
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, H0, "H0")
        X = X.contiguous()
        T, M, Fin = X.shape
        O = Wh.size(1)
        C = Fin + O
        S = 2 * K - 1
        Nn = g.N
        if M != Nn * B:
            raise ValueError(f"X has {M} rows per step, expected num_nodes*B = {Nn * B}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O) or H0.shape != (M, O):
            raise ValueError("DCRNNSeqFunction: inconsistent operand shapes")
        dev = X.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        TSzr = torch.empty(S, T, M, C, dtype=F32, device=dev)
        TSh = torch.empty(S, T, M, C, dtype=F32, device=dev)
        ZR = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(T, M, O, dtype=F32, device=dev)
        Hout = torch.empty((B, T, Nn, O) if btno else (T, M, O), dtype=F32, device=dev)
        H0c = H0.contiguous()
        state = _StateLayout(Hout, T, Nn, B, O, batch_major) if btno else None
        seg = T * M * C
        slab = bool(batch_major) or (B == 1 and slab_fits(g, C, K))
        if slab and K > 1 and not slab_fits(g, C, K):
            raise ValueError("DCRNNSeqFunction: batch-major rows need the LDS-resident stack (slab_fits)")

        def stack(TSx, t):
            if K < 2:
                return
            if slab:
                _slab_fwd(g, TSx[0, t], seg, B, C, K)
            else:
                _stack_fwd(g, TSx, t, K, Nn)

        # the input columns of segment 0 of both stacks (all T steps) and H0 into step 0 of the gate stack: one launch
        lib.call("pgt_dcrnn_stage_f32", ptr(X), ptr(H0c), T, M, Fin, O, ptr(TSzr[0]), ptr(TSh[0]), stream_of(lib, X))
        fuse = FUSE_GATE_EPILOGUES and O % 4 == 0
        for t in range(T):
            # H_{t-1}: the plain [M, O] state, or (btno: the states live in the [B, T, N, O] result) the hidden columns of
            # this step's stack segment 0, which the previous step's blend wrote
            Hp = H0c if t == 0 else (TSzr[0, t][:, Fin:] if btno else Hout[t - 1])
            Hnext = TSzr[0, t + 1][:, Fin:] if t + 1 < T else None
            Ht = state.step(t) if btno else Hout[t]
            stack(TSzr, t)
            if fuse:
                gemm_gru_zr(TSzr[0, t], C, seg, S, C, Wzr_c, 2 * O, 1, bzr, ZR[t], Hp, TSh[0, t], Fin)
                stack(TSh, t)
                gemm_gru_h(TSh[0, t], C, seg, S, C, Wh_c, O, 1, bh, HT[t], ZR[t], Hp, Ht, Hnext)
            else:
                gemm(TSzr[0, t], C, seg, S, C, Wzr_c, 2 * O, 1, ZR[t], 2 * O, 0, 2 * O, bzr, M, 2 * O)
                _gru_zr(ZR[t], Hp, TSh[0, t], Fin)
                stack(TSh, t)
                gemm(TSh[0, t], C, seg, S, C, Wh_c, O, 1, HT[t], O, 0, O, bh, M, O)
                _gru_h(HT[t], ZR[t], Hp, Ht, Hnext)
        ctx.g, ctx.K, ctx.B, ctx.Fin, ctx.slab = g, K, B, Fin, slab
        ctx.btno, ctx.batch_major = btno, bool(batch_major)
        ctx.has_bias = (bzr is not None, bh is not None)
        ctx.save_for_backward(TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dOut):
        if _seq64_adjoint_applies(ctx):
            # hidden 64, the input is data: all T steps of the adjoint in ONE launch (csrc/seq64.hip) — faster than the six
            # launches per step at every batch size (B = 64: 1.32 -> 1.11 ms, B = 1024: 8.8 -> 6.9 ms)
            return (None,) + _seq64_adjoint(ctx, dOut) + (None,) * 5
        TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c = ctx.saved_tensors
        g, K, Fin = ctx.g, ctx.K, ctx.Fin
        S, T, M, C = TSzr.shape
        O = HT.size(2)
        Nn = g.N
        dev = dOut.device
        dOut = dOut.contiguous()
        if ctx.btno:      # gradient and states in the reference's [B, T, N, O] layout, read in place step by step
            grad_in = _StateLayout(dOut, T, Nn, ctx.B, O, ctx.batch_major)
            states = _StateLayout(Hout, T, Nn, ctx.B, O, ctx.batch_major)
        need_x = ctx.needs_input_grad[0]
        dX = torch.zeros(T, M, Fin, dtype=F32, device=dev) if need_x else None
        dH = torch.zeros(M, O, dtype=F32, device=dev)      # running d/dH_t
        dPzr = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        dPh = torch.empty(T, M, O, dtype=F32, device=dev)
        Wh_b, folded = fold_backward_weight(Wh_c, K, C)
        Wzr_b, _ = fold_backward_weight(Wzr_c, K, C)
        # When the input needs no gradient (the usual case: X is data), only the hidden-state columns of the stack
        # gradient are ever read, and the adjoint of the stack acts on every column independently.  The whole backward
        # stack then runs on those O columns alone: the feature-gradient GEMMs use the weight rows of the hidden
        # columns (S*O = 320 output columns instead of S*C = 330, i.e. 2.5 instead of 3 128-wide column tiles) and
        # write [S][M][O] segments -- 256-byte rows, float4 stores, no 8-byte holes where the input columns would be
        # -- and the stack adjoint reads and writes 64- instead of 66-wide rows.
        skip_x = (SKIP_INPUT_COLUMNS_WHEN_UNUSED and not need_x and Fin > 0 and O % 4 == 0 and S > 1)
        Cb, Fb = (O, 0) if skip_x else (C, Fin)                      # width of the stack gradient, its first H column
        G = torch.empty(S, M, Cb, dtype=F32, device=dev)
        if skip_x:
            WhH = Wh_b.view(S, C, O)[:, Fin:, :].reshape(S * O, O).contiguous()
            WzrH = Wzr_b.view(S, C, 2 * O)[:, Fin:, :].reshape(S * O, 2 * O).contiguous()
            NH = S * O
            n1 = (NH // 128) * 128 if (SPLIT_FEATURE_GRADIENT and NH % 128 != 0 and NH > 128 and ((NH // 128) * 128) % O == 0) else NH
            # tall batches: ONE product over all S*O <= 320 columns on the symmetric split-bf16 kernel (dP is read once;
            # column blocks 8 and 9 ride along as second blocks) instead of 256 columns + a 64-column remainder
            # (from 8 192 rows, the split-bf16 kernels' own floor: at B = 64, M = 13 248, the one product is 4.5 % of the step
            # faster than 256 + 64 since the round-3 kernels, 2.55 -> 2.44 ms)
            if ONE_FEATURE_GRADIENT and M >= ONE_FEATURE_GRADIENT_MIN_ROWS and 128 < NH <= 320 and NH % 32 == 0 and 2 * O <= 128 and O % 32 == 0:
                n1 = NH

        def feature_grad(dP, Wfull, WH, Kd):
            """G[s] = dP W_s^T for every stack segment (the stack adjoint consumes G in place)."""
            if not skip_x:
                gemm(dP, Kd, 0, 1, Kd, Wfull, 1, Kd, G, C, M * C, C, None, M, S * C)
                return
            gemm(dP, Kd, 0, 1, Kd, WH, 1, Kd, G, O, M * O, O, None, M, n1)
            if n1 < NH:                                              # the narrow remainder: 64-wide tiles, no padding
                gemm(dP, Kd, 0, 1, Kd, WH[n1:], 1, Kd, G[n1 // O], O, M * O, O, None, M, NH - n1)
        B = ctx.B
        seg = T * M * C
        need_wzr = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        need_wh = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dWzr = torch.zeros_like(Wzr_c) if need_wzr else None
        dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if (need_wzr and ctx.has_bias[0]) else None
        dWh = torch.zeros_like(Wh_c) if need_wh else None
        dbh = torch.zeros(O, dtype=F32, device=dev) if (need_wh and ctx.has_bias[1]) else None
        overlap = OVERLAP_WEIGHT_GRADIENTS and dev.type == "cuda" and KERNEL_TIMER is None and (need_wzr or need_wh)
        if overlap:
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)          # dW / db zero-fills and the saved stacks are ready

        def weight_grads(t):
            """dW += stack_t^T dPRE_t for one time step (pgt_gemm_tn_acc_f32 accumulates)."""
            if need_wzr:
                gemm_tn_acc(TSzr[0, t], C, seg, S, C, dPzr[t], 2 * O, dWzr, 2 * O, dbzr, M, 2 * O)
            if need_wh:
                gemm_tn_acc(TSh[0, t], C, seg, S, C, dPh[t], O, dWh, O, dbh, M, O)

        def stack_bwd():
            if K < 2:
                return
            if ctx.slab:
                _slab_bwd(g, G[0], M * Cb, B, Cb, K, folded)
            else:
                _stack_bwd(g, G, K, Nn, folded)

        for t in range(T - 1, -1, -1):
            Hp = H0c if t == 0 else (states.step(t - 1) if ctx.btno else Hout[t - 1])
            # d/dH_t = dOut[t] + running state gradient, summed inside the gate-backward kernel
            # (+ the later step's gate-stack gradient of H, still sitting in G[0]: no accumulation pass of its own)
            _gru_h_bwd(grad_in.step(t) if ctx.btno else dOut[t], ZR[t], Hp, HT[t], dPh[t], dPzr[t], dH, accumulate=False,
                       dHn2=dH, dHn3=None if (t == T - 1 or not FOLD_STATE_GRADIENT) else G[0][:, Fb:])
            # candidate conv: dT = dPh Wh^T ; adjoint of the stack
            feature_grad(dPh[t], Wh_b, WhH if skip_x else None, O)
            stack_bwd()
            _gru_zr_bwd(G[0], Fb, ZR[t], Hp, dPzr[t], dH)
            if overlap:                     # dPh[t], dPzr[t] are final: their weight gradients go to the side stream
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    weight_grads(t)
            if need_x:
                copy2d(dX[t], G[0][:, :Fin])
            # gate convs
            feature_grad(dPzr[t], Wzr_b, WzrH if skip_x else None, 2 * O)
            stack_bwd()
            if not FOLD_STATE_GRADIENT:
                add2d(dH, G[0][:, Fb:])
            if need_x:
                add2d(dX[t], G[0][:, :Fin])
        if ctx.needs_input_grad[1] and FOLD_STATE_GRADIENT:
            add2d(dH, G[0][:, Fb:])              # d/dH0: the first step's gate-stack gradient joins here
        if overlap:
            main.wait_stream(side)
        else:
            if need_wzr:
                gemm_tn_acc(TSzr, C, seg, S, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, T * M, 2 * O)
            if need_wh:
                gemm_tn_acc(TSh, C, seg, S, C, dPh, O, dWh, O, dbh, T * M, O)
        dH0 = dH if ctx.needs_input_grad[1] else None
        return dX, dH0, dWzr, dbzr, dWh, dbh, None, None, None, None, None


# hidden width 64 on a graph whose block fits a CU's LDS: the whole T-step forward of every sample in ONE launch (csrc/seq64.hip);
# PGT_SEQ64=0 = the per-step launches of DCRNNSeqFunction (A/B)
USE_SEQ64 = os.environ.get("PGT_SEQ64", "1") != "0"
# smallest batch that takes it: a sample occupies ONE CU for the whole sequence, so below ~100 samples the per-step launches —
# which spread every step over all 256 CUs — are faster (B = 64: 0.82 ms against 0.98 ms forward; B = 256: 1.68 against 1.11)
SEQ64_MIN_BATCH = int(os.environ.get("PGT_SEQ64_MIN_B", "96"))
USE_SEQ64_BWD = os.environ.get("PGT_SEQ64_BWD", "1") != "0"      # 0: the per-step adjoint launches behind the one-launch forward (A/B)


def seq64_fits(g, Fin, O, K):
    """Whether pgt_dcrnn_seq64_f32 takes this graph / width: hidden 64, two input channels, K = 2 | 3, the sample's block + both
    operators + the weight ring within a CU's LDS — and finite operator coefficients (a node without incoming edges makes
    DConv's 1 / deg infinite, dcrnn.py:71-77: inf / nan placement is the general path's speciality, csrc/gemm_bx.hip)."""
    return bool(getattr(g, "finite", False)) and bool(_lib.get_lib()._pgt_dcrnn_seq64_fits(g.N, g.E, g.E, int(Fin), int(O), int(K)))


def _seq64_adjoint(ctx, dOut):
    """(dH0, dWzr, dbzr, dWh, dbh) of a hidden-64 sequence whose input is data: the whole BPTT in ONE launch (csrc/seq64.hip) on
    what either forward path saved (both stacks, Z | R, the candidates, the states in the reference's [B, T, N, O] layout,
    batch-major rows) + the two weight-gradient products over all T steps."""
    lib = _lib.get_lib()
    TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c = ctx.saved_tensors
    g, K = ctx.g, ctx.K
    S, T, M, C = TSzr.shape
    O = HT.size(2)
    N = g.N
    B, Fin = M // N, C - O
    dev = dOut.device
    dOc = dOut.contiguous()
    need = ctx.needs_input_grad
    dPzr = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
    dPh = torch.empty(T, M, O, dtype=F32, device=dev)
    dH0 = torch.empty(M, O, dtype=F32, device=dev) if need[1] else None
    Wp = torch.empty(int(lib._pgt_dcrnn_seq64_pack_floats(K)), dtype=F32, device=dev)
    lib.call("pgt_dcrnn_seq64_pack_bwd_f32", ptr(Wzr_c), ptr(Wh_c), Fin, K, ptr(Wp), stream_of(lib, Wp))
    nws = int(lib._pgt_dcrnn_seq64_bwd_ws_floats(N, B))
    ws = torch.empty(nws, dtype=F32, device=dev)
    to, ti = g.bwd_o.struct(), g.bwd_i.struct()
    work = 4.0 * (dOc.numel() + Hout.numel() + ZR.numel() + HT.numel() + dPzr.numel() + dPh.numel()) if KERNEL_TIMER else 0
    _timed("seq64", work, lambda: lib.call(
        "pgt_dcrnn_seq64_bwd_f32", ctypes.byref(to), ctypes.byref(ti), g.E, g.E, N, ptr(dOc), T * N * O, N * O, ptr(Hout), T * N * O,
        N * O, ptr(H0c), ptr(ZR), ptr(HT), ptr(Wp), B, T, Fin, K, ptr(dPzr), ptr(dPh), ptr(dH0), ptr(ws), nws,
        stream_of(lib, dPh)), tag=("bwd", B, T, N))
    seg = T * M * C
    dWzr = dbzr = dWh = dbh = None
    if need[2] or need[3]:
        dWzr = torch.zeros_like(Wzr_c)
        dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if ctx.has_bias[0] else None
        gemm_tn_acc(TSzr, C, seg, S, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, T * M, 2 * O)
    if need[4] or need[5]:
        dWh = torch.zeros_like(Wh_c)
        dbh = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias[1] else None
        gemm_tn_acc(TSh, C, seg, S, C, dPh, O, dWh, O, dbh, T * M, O)
    return dH0, dWzr, dbzr, dWh, dbh


def _seq64_adjoint_applies(ctx):
    """The one-launch adjoint takes a DCRNNSeqFunction / DCRNNSeq64Function context: hidden 64 with two input channels on a graph
    pgt_dcrnn_seq64_fits covers, batch-major rows with the states in the reference's layout, the input being data."""
    if not (USE_SEQ64 and USE_SEQ64_BWD) or ctx.needs_input_grad[0] or not (ctx.btno and ctx.batch_major and ctx.slab):
        return False
    TSzr, _, _, HT = ctx.saved_tensors[:4]
    return seq64_fits(ctx.g, TSzr.size(3) - HT.size(2), HT.size(2), ctx.K)


class DCRNNSeq64Function(torch.autograd.Function):
    """BatchedDCRNN.forward at hidden width 64 (dcrnn.py:429-475): X [B, T, N, Fin], H0 [B * N, O] | None -> [B, T, N, O] in ONE
    launch for all T steps (csrc/seq64.hip: one workgroup per sample, diffusion terms and products never leave the CU); the
    launch leaves behind exactly what DCRNNSeqFunction.forward saves (both stacks, Z | R, the candidates, the states in the
    reference's layout), so the hand-written BPTT of DCRNNSeqFunction.backward runs on it unchanged."""

    @staticmethod
    def forward(ctx, X, H0, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        if X.dim() != 4:
            raise ValueError(f"DCRNNSeq64Function: X must be [B, T, N, in], got {tuple(X.shape)}")
        B, T, N, Fin = X.shape
        O = Wh.size(1)
        S, C = 2 * K - 1, Fin + O
        M = B * N
        if N != g.N:
            raise ValueError(f"X has {N} nodes, the graph {g.N}")
        if Wzr.shape != (S * C, 2 * O) or Wh.shape != (S * C, O):
            raise ValueError(f"DCRNNSeq64Function: inconsistent operand shapes: X has {Fin} input channels, the stacked weights "
                             f"{tuple(Wzr.shape)} / {tuple(Wh.shape)} (K = {K})")
        if (bzr is not None and bzr.shape != (2 * O,)) or (bh is not None and bh.shape != (O,)):
            raise ValueError("DCRNNSeq64Function: inconsistent bias shapes")
        if not lib._pgt_dcrnn_seq64_fits(N, g.E, g.E, Fin, O, K):
            raise ValueError("DCRNNSeq64Function: shape not covered (seq64_fits)")
        dev = X.device
        Xc = X.contiguous()
        H0c = None
        if H0 is not None:
            check_tensor(lib, H0, "H0")
            if H0.shape != (M, O):
                raise ValueError(f"H0 must be {(M, O)}, got {tuple(H0.shape)}")
            H0c = H0.contiguous()
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        Wp = torch.empty(int(lib._pgt_dcrnn_seq64_pack_floats(K)), dtype=F32, device=dev)
        lib.call("pgt_dcrnn_seq64_pack_f32", ptr(Wzr_c), ptr(Wh_c), Fin, K, ptr(Wp), stream_of(lib, Wp))
        TSzr = torch.empty(S, T, M, C, dtype=F32, device=dev)
        TSh = torch.empty(S, T, M, C, dtype=F32, device=dev)
        ZR = torch.empty(T, M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(T, M, O, dtype=F32, device=dev)
        Hout = torch.empty(B, T, N, O, dtype=F32, device=dev)
        so, si = g.fwd_o.struct(), g.fwd_i.struct()
        work = 4.0 * (Xc.numel() + Hout.numel() + TSzr.numel() + TSh.numel() + ZR.numel() + HT.numel()) if KERNEL_TIMER else 0
        _timed("seq64", work, lambda: lib.call(
            "pgt_dcrnn_seq64_f32", ctypes.byref(so), ctypes.byref(si), g.E, g.E, N, ptr(Xc), T * N * Fin, N * Fin, ptr(H0c), ptr(Wp),
            ptr(Wzr_c), ptr(bzr), ptr(Wh_c), ptr(bh), B, T, Fin, K, ptr(Hout), T * N * O, N * O, ptr(TSzr), ptr(TSh), T * M * C, M * C,
            ptr(ZR), ptr(HT), stream_of(lib, Hout)), tag=("fwd", B, T, N))
        if any(ctx.needs_input_grad):
            if H0c is None:
                H0c = torch.zeros(M, O, dtype=F32, device=dev)
            ctx.g, ctx.K, ctx.B, ctx.Fin, ctx.slab = g, K, B, Fin, True
            ctx.btno, ctx.batch_major = True, True
            ctx.has_bias = (bzr is not None, bh is not None)
            ctx.dims = (B, T, N, Fin)
            ctx.save_for_backward(TSzr, TSh, ZR, HT, H0c, Hout, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dOut):
        if not _seq64_adjoint_applies(ctx):
            # the input wants a gradient (or the A/B switch is off): the per-step adjoint launches on what the forward saved
            dX, dH0, dWzr, dbzr, dWh, dbh = DCRNNSeqFunction.backward(ctx, dOut)[:6]
            if dX is not None:                                 # [T, B N, Fin] (time-major steps of batch-major rows) -> X's layout
                B, T, N, Fin = ctx.dims
                dX = dX.view(T, B, N, Fin).permute(1, 0, 2, 3).contiguous()
            return dX, dH0, dWzr, dbzr, dWh, dbh, None, None
        return (None,) + _seq64_adjoint(ctx, dOut) + (None, None)


# --------------------------------------------------------------------------------------------- generic building blocks

class SpmmFunction(torch.autograd.Function):
    """Y = alpha * A @ X on [n_rows, F] (propagate with aggr="add"); the gradient runs on the transposed operator."""

    @staticmethod
    def forward(ctx, X, fwd, bwd, alpha):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        Y = torch.empty_like(Xc)
        spmm(fwd, Xc, Y, alpha=alpha)
        ctx.bwd, ctx.alpha = bwd, alpha
        return Y

    @staticmethod
    def backward(ctx, dY):
        dYc = dY.contiguous()
        dX = torch.empty_like(dYc)
        spmm(ctx.bwd, dYc, dX, alpha=ctx.alpha)
        return dX, None, None, None


def propagate(g, X2d, alpha=1.0):
    """g: SymGraph (fwd/bwd).  X2d [N, F] -> A @ X2d with autograd."""
    return SpmmFunction.apply(X2d, g.fwd, g.bwd, float(alpha))


class LinearFunction(torch.autograd.Function):
    """Y[M,N] = X[M,K] @ W_kn[K,N] (+ bias) on the fp32 MFMA GEMM; W_kn may be any strided 2-D view (e.g. weight.t())."""

    @staticmethod
    def forward(ctx, X, W_kn, bias):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        Y = linear_fwd(Xc, W_kn, bias)
        ctx.save_for_backward(Xc, W_kn)
        ctx.has_bias = bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        Xc, W_kn = ctx.saved_tensors
        M, K = Xc.shape
        N = W_kn.size(1)
        dYc = dY.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            dX = torch.empty(M, K, dtype=F32, device=dYc.device)
            # dX = dY @ W_kn^T : B element (k' = n, n' = k) is W_kn[k, n]
            gemm(dYc, N, 0, 1, N, W_kn, W_kn.stride(1), W_kn.stride(0), dX, K, 0, K, None, M, K)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros(K, N, dtype=F32, device=dYc.device)
            db = torch.zeros(N, dtype=F32, device=dYc.device) if ctx.has_bias else None
            gemm_tn_acc(Xc, K, 0, 1, K, dYc, N, dW, N, db, M, N)
        return dX, dW, db


def linear(X2d, W_kn, bias=None):
    return LinearFunction.apply(X2d, W_kn, bias)


class ReadoutFunction(torch.autograd.Function):
    """The read-out the reference's models apply to the states of a recurrent layer — `linear(relu(h))` or `linear(h)` with a
    torch.nn.Linear of 1 .. 4 outputs (examples/indexBatching/tgcn/metr_la_main.py:43-45, examples/recurrent/dcrnn_example.py:27-31)
    — as one streaming pass each way (csrc/readout.hip): X [M, K] rows (the PRE-relu states), weight [N, K] as torch.nn.Linear
    holds it, bias [N] | None -> Y [M, N]; the adjoint reads X once and writes dX once (relu's mask applied in the same pass),
    weight / bias gradients from per-workgroup partial sums added in index order (deterministic)."""

    @staticmethod
    def forward(ctx, X, weight, bias, relu):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, weight, "weight")
        M, K = X.shape
        N = weight.size(0)
        if weight.shape != (N, K) or (bias is not None and bias.shape != (N,)):
            raise ValueError(f"read-out: weight must be [out, {K}] and bias [out], got {tuple(weight.shape)}")
        xp, ldx = _rows(X, "X")
        W = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        Y = torch.empty(M, N, dtype=F32, device=X.device)
        _timed("readout", 4.0 * M * (K + N) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_relu_linear_f32", xp, ldx, ptr(W), ptr(b), M, K, N, int(bool(relu)), ptr(Y), N, stream_of(lib, Y)), tag=("fwd", M, K, N))
        ctx.save_for_backward(X, W)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        X, W = ctx.saved_tensors
        M, K = X.shape
        N = W.size(0)
        dYc = dY.contiguous()
        xp, ldx = _rows(X, "X")
        need = ctx.needs_input_grad
        dX = torch.empty(M, K, dtype=F32, device=X.device) if need[0] else None
        dW = torch.empty(N, K, dtype=F32, device=X.device) if need[1] else None
        db = torch.empty(N, dtype=F32, device=X.device) if (ctx.has_bias and need[2]) else None
        nws = int(lib._pgt_relu_linear_bwd_ws_floats(K, N))
        ws = _det_workspace(X.device, nws)
        _timed("readout", 4.0 * M * (2 * K + N) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_relu_linear_bwd_f32", xp, ldx, ptr(dYc), N, ptr(W), M, K, N, int(ctx.relu), ptr(dX), K, ptr(dW), ptr(db), ptr(ws), nws,
            stream_of(lib, X)), tag=("bwd", M, K, N))
        return dX, dW, db, None


def readout_fits(x, weight, bias):
    """Whether the streaming read-out kernels take this call: fp32 rows of 4 .. 64 (a multiple of 4) floats that are 16-byte
    addressable, 1 .. 4 outputs."""
    if x.dim() < 2 or weight.dim() != 2 or x.size(-1) != weight.size(1):
        return False
    K, N = weight.size(1), weight.size(0)
    return bool(_lib.get_lib()._pgt_relu_linear_fits(int(K), int(N)))


def readout(x, weight, bias, relu):
    """linear(relu(x)) / linear(x) over the last dimension of x through ReadoutFunction (rows in memory order, no copy when x is
    contiguous)."""
    lead = x.shape[:-1]
    K = x.size(-1)
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    return ReadoutFunction.apply(x2, weight, bias, relu).view(*lead, weight.size(0))


# --------------------------------------------------------------------------------------------- T-GCN cell

def _graph_task():
    """Identity of the backward walk in flight (-1 outside one): what a deposit is stamped with."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


class _PackDeposit:
    """Where the fused T-GCN cells that share one set of folded operands (nn/_states.py packed_once) sum their weight / bias
    gradients: one [dWzr | dWh | dbzr | dbh] buffer the adjoint kernels ACCUMULATE into (pgt_tgcn_cell_bwd_acc_f32).  Every cell
    returns None for the four operands to autograd — except the first one to run in a backward walk, which hands out a ZERO dbh of
    its own: one defined gradient is what makes the engine call TGCNWeightsFunction.backward once all the cells of that walk have
    run, and that is where the sums are taken from (`take`) and ADDED to whatever autograd itself brought (cells of the same pack
    on the non-deposit path).  The buffer belongs to ONE walk: it is stamped with the engine's graph-task id, and a deposit that
    meets another walk's buffer (torch.autograd.grad w.r.t. H0 never reaches the weights node; a backward that raised half way)
    starts over instead of accumulating onto it.  Cells that did not run (a loss that does not reach them) never deposited."""

    def __init__(self):
        self.buf, self.views, self.task = None, None, None

    def slot(self, C, O, device):
        """(views dWzr, dbzr, dWh, dbh of the buffer, first): `first` = nothing deposited in THIS walk yet (the kernel stores)."""
        task = _graph_task()
        first = self.buf is None or self.task != task
        if first:
            self.buf = torch.empty(C * 3 * O + 3 * O, dtype=F32, device=device)
            b = self.buf
            self.views = (b[:C * 2 * O].view(C, 2 * O), b[C * 3 * O:C * 3 * O + 2 * O], b[C * 2 * O:C * 3 * O].view(C, O),
                          b[C * 3 * O + 2 * O:])
            self.task = task
        return self.views, first

    def take(self):
        """The sums of the walk in flight (None if its cells deposited nothing; another walk's leftovers are dropped)."""
        v, stale = self.views, self.task != _graph_task()
        self.buf, self.views, self.task = None, None, None
        return None if stale else v


class TGCNWeightsFunction(torch.autograd.Function):
    """The module's parameters -> the operands of the two gate products of the fused cell (csrc/tgcn.hip):
    conv_{z,r,h}.lin.weight [O, Fin], conv_{z,r,h}.bias, linear_{z,r,h}.weight [O, 2O], linear_{z,r,h}.bias
    -> Wzr [Fin + O, 2O], bzr [2O], Wh [Fin + O, O], bh [O]; one launch forward, one backward."""

    @staticmethod
    def _operands(params):
        Wcz, Wcr, Wch, bcz, bcr, bch, Lz, Lr, Lh, lbz, lbr, lbh = params
        Wc = [t.contiguous() for t in (Wcz, Wcr, Wch)]
        L = [t.contiguous() for t in (Lz, Lr, Lh)]
        bc = [None if t is None else t.contiguous() for t in (bcz, bcr, bch)]
        lb = [None if t is None else t.contiguous() for t in (lbz, lbr, lbh)]
        return Wc, bc, L, lb

    @staticmethod
    def repack(params, packed):
        """The pack launch alone, into operands that already exist (nn/_states.py packed_once refreshes cached operands with it)."""
        lib = _lib.get_lib()
        Wc, bc, L, lb = TGCNWeightsFunction._operands(params)
        O, Fin = Wc[0].shape
        Wzr, bzr, Wh, bh = packed
        lib.call("pgt_tgcn_pack_weights_f32", _lib.ptr3(*Wc), _lib.ptr3(*bc), _lib.ptr3(*L), _lib.ptr3(*lb), Fin, O, ptr(Wzr),
                 ptr(bzr), ptr(Wh), ptr(bh), stream_of(lib, Wzr))

    @staticmethod
    def forward(ctx, *params):
        lib = _lib.get_lib()
        Wc, bc, L, lb = TGCNWeightsFunction._operands(params)
        for t in Wc + L:
            check_tensor(lib, t, "T-GCN parameter")
        O, Fin = Wc[0].shape
        if any(w.shape != (O, Fin) for w in Wc) or any(l.shape != (O, 2 * O) for l in L):
            raise ValueError("T-GCN: the three gates must share one shape (lin.weight [out, in], linear.weight [out, 2 out])")
        dev = Wc[0].device
        C = Fin + O
        buf = torch.empty(C * 3 * O + 3 * O, dtype=F32, device=dev)
        Wzr, Wh = buf[:C * 2 * O].view(C, 2 * O), buf[C * 2 * O:C * 3 * O].view(C, O)
        bzr, bh = buf[C * 3 * O:C * 3 * O + 2 * O], buf[C * 3 * O + 2 * O:]
        TGCNWeightsFunction.repack(params, (Wzr, bzr, Wh, bh))
        # The folded operands are products of two parameters each, so the adjoint needs the parameters.  They are kept as plain
        # attributes, not through save_for_backward: the operands of one pack may feed several independent graphs (packed_once:
        # o1 = m(x1); o2 = m(x2); o1.backward(); o2.backward()), and autograd frees saved tensors after the first walk.  Inputs
        # held by their own node form no reference cycle; the in-place check save_for_backward would have made is done by hand.
        ctx.kept = (Wc, L, bc)
        # the cells fed by these operands sum their weight / bias gradients into ONE buffer (TGCNCellFunction.backward) instead of
        # handing autograd four small tensors per cell to add up (44 five-microsecond adds per T = 12 step): see _PackDeposit
        ctx.deposit = _PackDeposit()
        Wzr._pgt_deposit = ctx.deposit
        ctx.set_materialize_grads(False)
        ctx.versions = tuple(tensor_version(t) for t in params if t is not None)
        ctx.kept_sources = tuple(t for t in params if t is not None)
        ctx.bias_mask = tuple(t is not None for t in bc)
        ctx.lb_mask = tuple(t is not None for t in lb)
        ctx.dims = (Fin, O)
        return Wzr, bzr, Wh, bh

    @staticmethod
    def backward(ctx, dWzr, dbzr, dWh, dbh):
        lib = _lib.get_lib()
        if tuple(tensor_version(t) for t in ctx.kept_sources) != ctx.versions:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                               "a T-GCN parameter changed between the forward pass and this backward pass")
        Wc, L, bc = ctx.kept
        dep = ctx.deposit.take()
        if dep is not None:                                   # what the fused cells summed among themselves (+ anything autograd brought)
            dWzr = dep[0] if dWzr is None else dWzr + dep[0]
            dbzr = dep[1] if dbzr is None else dbzr + dep[1]
            dWh = dep[2] if dWh is None else dWh + dep[2]
            # the incoming dbh = the first cell's zero trigger + whatever cells on the non-deposit path returned (autograd sums
            # them out of place): the deposit is added to it, never substituted for it
            dbh = dep[3] if dbh is None else dbh + dep[3]
        if dWzr is None and dbzr is None and dWh is None and dbh is None:
            return (None,) * 12
        Fin, O = ctx.dims
        dev = Wc[0].device
        C = Fin + O
        z = lambda *shape: torch.zeros(*shape, dtype=F32, device=dev)
        dWzr = dWzr.contiguous() if dWzr is not None else z(C, 2 * O)
        dbzr = dbzr.contiguous() if dbzr is not None else z(2 * O)
        dWh = dWh.contiguous() if dWh is not None else z(C, O)
        dbh = dbh.contiguous() if dbh is not None else z(O)
        per = O * Fin + O + O * 2 * O + O
        buf = torch.empty(3 * per, dtype=F32, device=dev)
        dWc = [buf[g * per:g * per + O * Fin].view(O, Fin) for g in range(3)]
        dbc = [buf[g * per + O * Fin:g * per + O * Fin + O] if ctx.bias_mask[g] else None for g in range(3)]
        dL = [buf[g * per + O * Fin + O:g * per + O * Fin + O + 2 * O * O].view(O, 2 * O) for g in range(3)]
        dlb = [buf[(g + 1) * per - O:(g + 1) * per] if ctx.lb_mask[g] else None for g in range(3)]
        lib.call("pgt_tgcn_unpack_weight_grads_f32", ptr(dWzr), ptr(dbzr), ptr(dWh), ptr(dbh), _lib.ptr3(*Wc), _lib.ptr3(*bc),
                 _lib.ptr3(*L), Fin, O, _lib.ptr3(*dWc), _lib.ptr3(*dbc), _lib.ptr3(*dL), _lib.ptr3(*dlb), stream_of(lib, buf))
        return (*dWc, *dbc, *dL, *dlb)


class TGCNCellFunction(torch.autograd.Function):
    """One T-GCN GRU step (temporalgcn.py:82-130).  X [M, Fin], H [M, O] -> H' [M, O] with M = num_nodes * Bt rows, either
    node-major (m = n * Bt + b) or — `batch_major` — batch-major (m = b * N + n: TGCN2's own [B, N, .] layout, so the hidden
    state is never transposed; only the Fin input columns travel to the node-major layout the aggregation wants and back).

    The three GCNConv gates share ONE aggregation AX = A_hat X at the input width (the reference aggregates three times at
    width O), and conv_g -> linear_g is folded into one product per gate pair on the operand [AX | H'] (TGCNWeightsFunction):
        Z | R = sigmoid([AX | H] Wzr + bzr)  + H * R          pgt_gemm_gru_zr_f32 (gate chain in the GEMM epilogue)
        H'    = Z H + (1 - Z) tanh([AX | H * R] Wh + bh)      pgt_gemm_gru_h_f32
    Backward: the gate adjoints of the DCRNN cell (pgt_gru_h_bwd_f32 / pgt_gru_zr_bwd_f32), two input-gradient products,
    two weight-gradient products, one transposed aggregation when X needs a gradient."""

    @staticmethod
    def forward(ctx, X, H, Wzr, bzr, Wh, bh, g, Bt, batch_major):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, H, "H")
        Xc = X.contiguous()
        M, Fin = Xc.shape
        O = Wh.size(1)
        C = Fin + O
        N = g.N
        if M != N * Bt or H.shape != (M, O):
            raise ValueError(f"TGCN: X has {M} rows, H {tuple(H.shape)}, expected num_nodes*B = {N * Bt} rows of {O}")
        if Wzr.shape != (C, 2 * O) or Wh.shape != (C, O) or (bzr is not None and bzr.shape != (2 * O,)) or \
                (bh is not None and bh.shape != (O,)):
            raise ValueError(f"TGCN: X has {Fin} input channels, the folded weights {tuple(Wzr.shape)} / {tuple(Wh.shape)} expect "
                             f"{Wzr.size(0) - O} (in_channels of the module)")
        dev = Xc.device
        AX = TGCNCellFunction._aggregate(g.fwd, Xc, N, Bt, Fin, batch_major)
        ZR = torch.empty(M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(M, O, dtype=F32, device=dev)
        Hn = torch.empty(M, O, dtype=F32, device=dev)
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        ctx.g, ctx.Bt, ctx.batch_major = g, Bt, batch_major
        if USE_TGCN_FUSED and lib._pgt_tgcn_cell_fits(Fin, O):
            # hidden width 32: the whole row-local part of the cell in ONE launch (csrc/tgcn_cell.hip), H read in place
            Hs = H if H.stride(1) == 1 else H.contiguous()
            hp, ldh = _rows(Hs, "H")
            _timed("tgcn_cell", 4.0 * M * (Fin + 5 * O) if KERNEL_TIMER else 0, lambda: lib.call(
                "pgt_tgcn_cell_f32", ptr(AX), Fin, hp, ldh, ptr(Wzr_c), ptr(bzr), ptr(Wh_c), ptr(bh), M, Fin, O, ptr(ZR), ptr(HT),
                ptr(Hn), O, stream_of(lib, Hn)), tag=("fwd", M, Fin, O))
            ctx.fused = True
            ctx.deposit = getattr(Wzr, "_pgt_deposit", None) if (Wzr.requires_grad and bzr is not None and bh is not None) else None
            ctx.save_for_backward(AX, Hs, ZR, HT, Wzr_c, Wh_c)
            return Hn
        XH = torch.empty(M, C, dtype=F32, device=dev)          # [AX | H]
        XHR = torch.empty(M, C, dtype=F32, device=dev)         # [AX | H * R]
        copy2d(XH[:, :Fin], AX)
        copy2d(XH[:, Fin:], H if H.stride(1) == 1 else H.contiguous())
        copy2d(XHR[:, :Fin], AX)
        Hv = XH[:, Fin:]
        if FUSE_GATE_EPILOGUES and O % 4 == 0:
            gemm_gru_zr(XH, C, 0, 1, C, Wzr_c, 2 * O, 1, bzr, ZR, Hv, XHR, Fin)
            gemm_gru_h(XHR, C, 0, 1, C, Wh_c, O, 1, bh, HT, ZR, Hv, Hn)
        else:
            gemm(XH, C, 0, 1, C, Wzr_c, 2 * O, 1, ZR, 2 * O, 0, 2 * O, bzr, M, 2 * O)
            _gru_zr(ZR, Hv, XHR, Fin)
            gemm(XHR, C, 0, 1, C, Wh_c, O, 1, HT, O, 0, O, bh, M, O)
            _gru_h(HT, ZR, Hv, Hn)
        ctx.fused = False
        ctx.save_for_backward(XH, XHR, ZR, HT, Wzr_c, Wh_c)
        return Hn

    @staticmethod
    def _aggregate(csr, Xc, N, Bt, W, batch_major):
        """A X on [M, W] rows (node-major, or batch-major through two small transposition passes of the W columns)."""
        M = Xc.size(0)
        if batch_major and Bt > 1:
            Xnm = swap01(Xc, Bt, N, W)
            Y = torch.empty(N, Bt * W, dtype=F32, device=Xc.device)
            spmm(csr, Xnm.view(N, Bt * W), Y)
            return swap01(Y, N, Bt, W).view(M, W)
        Y = torch.empty(M, W, dtype=F32, device=Xc.device)
        spmm(csr, Xc.view(N, Bt * W), Y.view(N, Bt * W))
        return Y

    @staticmethod
    def backward(ctx, dHn):
        g, Bt = ctx.g, ctx.Bt
        N = g.N
        need = ctx.needs_input_grad
        if ctx.fused:
            AX, Hs, ZR, HT, Wzr_c, Wh_c = ctx.saved_tensors
            M, Fin = AX.shape
            O = Wh_c.size(1)
            C = Fin + O
            dev = AX.device
            dHn = dHn if (dHn.dim() == 2 and dHn.stride(1) == 1) else dHn.contiguous()
            if not need[0]:
                # the whole adjoint in one launch + a reduction of the per-workgroup weight-gradient sums (csrc/tgcn_cell.hip)
                lib = _lib.get_lib()
                gp, ldg = _rows(dHn, "dHn")
                hp, ldh = _rows(Hs, "H")
                dH = torch.empty(M, O, dtype=F32, device=dev)
                dep = ctx.deposit if all(need[2:6]) else None
                if dep is not None:
                    (dWzr, dbzr, dWh, dbh), first = dep.slot(C, O, dev)
                    entry = "pgt_tgcn_cell_bwd_f32" if first else "pgt_tgcn_cell_bwd_acc_f32"
                else:
                    buf = torch.empty(C * 3 * O + 3 * O, dtype=F32, device=dev)
                    dWzr, dWh = buf[:C * 2 * O].view(C, 2 * O), buf[C * 2 * O:C * 3 * O].view(C, O)
                    dbzr, dbh = buf[C * 3 * O:C * 3 * O + 2 * O], buf[C * 3 * O + 2 * O:]
                    entry, first = "pgt_tgcn_cell_bwd_f32", True
                nws = int(lib._pgt_tgcn_cell_bwd_ws_floats(Fin, O))
                ws = _det_workspace(dev, nws)
                _timed("tgcn_cell", 4.0 * M * (Fin + 6 * O) if KERNEL_TIMER else 0, lambda: lib.call(
                    entry, gp, ldg, ptr(AX), Fin, hp, ldh, ptr(ZR), ptr(HT), ptr(Wzr_c), ptr(Wh_c), M, Fin, O,
                    ptr(dH), O, ptr(dWzr), ptr(dbzr), ptr(dWh), ptr(dbh), ptr(ws), nws, stream_of(lib, dH)), tag=("bwd", M, Fin, O))
                if dep is not None:
                    # summed in the deposit; the first cell's zero dbh is the one defined gradient that brings autograd to
                    # TGCNWeightsFunction.backward, which reads the deposit (never a slice of the deposit itself: autograd may add
                    # other cells' gradients to it out of place)
                    return None, (dH if need[1] else None), None, None, None, (torch.zeros_like(dbh) if first else None), None, None, None
                return None, (dH if need[1] else None), dWzr, dbzr, dWh, dbh, None, None, None
            # the input gradient is wanted: rebuild the unfused operands ([AX | H], [AX | H * R]) and run the general adjoint
            XH = torch.empty(M, C, dtype=F32, device=dev)
            XHR = torch.empty(M, C, dtype=F32, device=dev)
            copy2d(XH[:, :Fin], AX)
            copy2d(XH[:, Fin:], Hs)
            copy2d(XHR[:, :Fin], AX)
            torch.mul(Hs, ZR[:, O:], out=XHR[:, Fin:])
        else:
            XH, XHR, ZR, HT, Wzr_c, Wh_c = ctx.saved_tensors
        M, C = XH.shape
        O = Wh_c.size(1)
        Fin = C - O
        dev = XH.device
        dHn = dHn.contiguous()
        Hv = XH[:, Fin:]
        d_pre_h = torch.empty(M, O, dtype=F32, device=dev)
        d_pre_zr = torch.empty(M, 2 * O, dtype=F32, device=dev)
        dH = torch.empty(M, O, dtype=F32, device=dev)
        _gru_h_bwd(dHn, ZR, Hv, HT, d_pre_h, d_pre_zr, dH, accumulate=False)
        dXHR = torch.empty(M, C, dtype=F32, device=dev)
        gemm(d_pre_h, O, 0, 1, O, Wh_c, 1, O, dXHR, C, 0, C, None, M, C)             # B(k = o, n = c) = Wh[c, o]
        _gru_zr_bwd(dXHR, Fin, ZR, Hv, d_pre_zr, dH)                                  # d_pre_r; dH += d(H R) R
        dXH = torch.empty(M, C, dtype=F32, device=dev)
        gemm(d_pre_zr, 2 * O, 0, 1, 2 * O, Wzr_c, 1, 2 * O, dXH, C, 0, C, None, M, C)
        add2d(dH, dXH[:, Fin:])
        dWzr = dbzr = dWh = dbh = None
        if need[2] or need[3]:
            dWzr = torch.zeros(C, 2 * O, dtype=F32, device=dev)
            dbzr = torch.zeros(2 * O, dtype=F32, device=dev)
            gemm_tn_acc(XH, C, 0, 1, C, d_pre_zr, 2 * O, dWzr, 2 * O, dbzr, M, 2 * O)
        if need[4] or need[5]:
            dWh = torch.zeros(C, O, dtype=F32, device=dev)
            dbh = torch.zeros(O, dtype=F32, device=dev)
            gemm_tn_acc(XHR, C, 0, 1, C, d_pre_h, O, dWh, O, dbh, M, O)
        dX = None
        if need[0]:
            dAX = torch.empty(M, Fin, dtype=F32, device=dev)
            axpby2d(dAX, dXH[:, :Fin], 1.0, dXHR[:, :Fin], 1.0)
            dX = TGCNCellFunction._aggregate(g.bwd, dAX, N, Bt, Fin, ctx.batch_major)
        return dX, (dH if need[1] else None), dWzr, dbzr, dWh, dbh, None, None, None


# --------------------------------------------------------------------------------------------- Chebyshev convolution

def _cheb_stack_fwd(g, TS, K, N):
    """Tx_0 = TS[0] given; Tx_1 = L Tx_0, Tx_k = 2 L Tx_{k-1} - Tx_{k-2} (PyG ChebConv.forward) on node-major rows."""
    for k in range(1, K):
        src, dst = TS[k - 1].view(N, -1), TS[k].view(N, -1)
        if k == 1:
            spmm(g.fwd, src, dst)
        else:
            spmm(g.fwd, src, dst, T=TS[k - 2].view(N, -1), alpha=2.0, beta=-1.0)


def _cheb_stack_bwd(g, G, K, N):
    """Adjoint of _cheb_stack_fwd on G [K][M][C] in place, highest order first: G_{k-1} += 2 L^T G_k ; G_{k-2} -= G_k;
    on exit G[0] holds d/dTx_0."""
    for k in range(K - 1, 1, -1):
        Gk = G[k].view(N, -1)
        Gp = G[k - 1].view(N, -1)
        spmm(g.bwd, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
        axpby2d(G[k - 2], G[k], -1.0, G[k - 2], 1.0)
    if K > 1:
        G0 = G[0].view(N, -1)
        spmm(g.bwd, G[1].view(N, -1), G0, T=G0, alpha=1.0, beta=1.0)


class ChebGRUCellFunction(torch.autograd.Function):
    """One GConvGRU cell step (gconv_gru.py:119-170) on rows [N, .]: Chebyshev stack of [X, H] -> ONE MFMA GEMM for the
    update and reset gates with sigmoid / H*R in its epilogue -> Chebyshev stack of [X, H*R] -> GEMM with tanh / blend
    in its epilogue (the same gate-fused entry points DCRNN uses: pgt_gemm_gru_zr/h_f32).  Backward: gate kernels,
    feature-gradient GEMMs, the stack adjoint on the transposed operator, one weight-gradient GEMM per stack.
    Wzr [K*C, 2*O] / Wh [K*C, O] stack lins[k].weight^T of the x- and h-convolutions (C = in + out)."""

    @staticmethod
    def forward(ctx, X, H, Wzr, bzr, Wh, bh, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        check_tensor(lib, H, "H")
        Xc, Hc = X.contiguous(), H.contiguous()
        M, Fin = Xc.shape
        O = Wh.size(1)
        C = Fin + O
        N = g.N
        if M != N or Hc.shape != (M, O) or Wzr.shape != (K * C, 2 * O) or Wh.shape != (K * C, O):
            raise ValueError("ChebGRUCellFunction: inconsistent operand shapes")
        dev = Xc.device
        Wzr_c, Wh_c = Wzr.contiguous(), Wh.contiguous()
        TSzr = torch.empty(K, M, C, dtype=F32, device=dev)
        TSh = torch.empty(K, M, C, dtype=F32, device=dev)
        copy2d(TSzr[0][:, :Fin], Xc)
        copy2d(TSzr[0][:, Fin:], Hc)
        copy2d(TSh[0][:, :Fin], Xc)
        _cheb_stack_fwd(g, TSzr, K, N)
        ZR = torch.empty(M, 2 * O, dtype=F32, device=dev)
        HT = torch.empty(M, O, dtype=F32, device=dev)
        Hout = torch.empty(M, O, dtype=F32, device=dev)
        if FUSE_GATE_EPILOGUES and O % 4 == 0:
            gemm_gru_zr(TSzr, C, M * C, K, C, Wzr_c, 2 * O, 1, bzr, ZR, Hc, TSh[0], Fin)
            _cheb_stack_fwd(g, TSh, K, N)
            gemm_gru_h(TSh, C, M * C, K, C, Wh_c, O, 1, bh, HT, ZR, Hc, Hout, None)
        else:
            gemm(TSzr, C, M * C, K, C, Wzr_c, 2 * O, 1, ZR, 2 * O, 0, 2 * O, bzr, M, 2 * O)
            _gru_zr(ZR, Hc, TSh[0], Fin)
            _cheb_stack_fwd(g, TSh, K, N)
            gemm(TSh, C, M * C, K, C, Wh_c, O, 1, HT, O, 0, O, bh, M, O)
            _gru_h(HT, ZR, Hc, Hout, None)
        ctx.g, ctx.K, ctx.Fin = g, K, Fin
        ctx.has_bias = (bzr is not None, bh is not None)
        ctx.save_for_backward(TSzr, TSh, ZR, HT, Hc, Wzr_c, Wh_c)
        return Hout

    @staticmethod
    def backward(ctx, dHout):
        TSzr, TSh, ZR, HT, Hc, Wzr_c, Wh_c = ctx.saved_tensors
        g, K, Fin = ctx.g, ctx.K, ctx.Fin
        _, M, C = TSzr.shape
        O = HT.size(1)
        N = g.N
        dev = dHout.device
        dHout = dHout.contiguous()
        dPzr = torch.empty(M, 2 * O, dtype=F32, device=dev)
        dPh = torch.empty(M, O, dtype=F32, device=dev)
        dH = torch.zeros(M, O, dtype=F32, device=dev)
        _gru_h_bwd(dHout, ZR, Hc, HT, dPh, dPzr, dH, accumulate=False)          # dH = dH' * Z ; dPh ; the Z half of dPzr
        G = torch.empty(K, M, C, dtype=F32, device=dev)
        gemm(dPh, O, 0, 1, O, Wh_c, 1, O, G, C, M * C, C, None, M, K * C)       # candidate conv: G_k = dPh W_k^T
        _cheb_stack_bwd(g, G, K, N)
        _gru_zr_bwd(G[0], Fin, ZR, Hc, dPzr, dH)                                # the R half of dPzr ; dH += dXHR_H * R
        need_x = ctx.needs_input_grad[0]
        dX = G[0][:, :Fin].clone() if need_x else None
        gemm(dPzr, 2 * O, 0, 1, 2 * O, Wzr_c, 1, 2 * O, G, C, M * C, C, None, M, K * C)
        _cheb_stack_bwd(g, G, K, N)
        add2d(dH, G[0][:, Fin:])
        if need_x:
            add2d(dX, G[0][:, :Fin])
        dWzr = dbzr = dWh = dbh = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dWzr = torch.zeros_like(Wzr_c)
            dbzr = torch.zeros(2 * O, dtype=F32, device=dev) if ctx.has_bias[0] else None
            gemm_tn_acc(TSzr, C, M * C, K, C, dPzr, 2 * O, dWzr, 2 * O, dbzr, M, 2 * O)
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            dWh = torch.zeros_like(Wh_c)
            dbh = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias[1] else None
            gemm_tn_acc(TSh, C, M * C, K, C, dPh, O, dWh, O, dbh, M, O)
        return dX, (dH if ctx.needs_input_grad[1] else None), dWzr, dbzr, dWh, dbh, None, None


class ChebConvFunction(torch.autograd.Function):
    """PyG ChebConv.forward on node-major rows [N*Bt, C] (any number of independent graphs-in-batch folded into the
    feature dimension): Tx_0 = X, Tx_1 = L X, Tx_k = 2 L Tx_{k-1} - Tx_{k-2}; out = sum_k Tx_k @ W_k^T + bias.
    Wst [K*C, O] stacks lins[k].weight^T; g is the scaled-Laplacian SymGraph (pgt_cheb_prep, variant 0)."""

    @staticmethod
    def forward(ctx, X, Wst, bias, g, K, Bt):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        Xc = X.contiguous()
        M, C = Xc.shape
        N = g.N
        if M != N * Bt:
            raise ValueError(f"ChebConv: X has {M} rows, expected num_nodes*B = {N * Bt}")
        O = Wst.size(1)
        TS = torch.empty(K, M, C, dtype=F32, device=Xc.device)
        copy2d(TS[0], Xc)
        _cheb_stack_fwd(g, TS, K, N)
        Wc = Wst.contiguous()
        out = torch.empty(M, O, dtype=F32, device=Xc.device)
        gemm(TS, C, M * C, K, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K = g, K
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc)
        return out

    @staticmethod
    def backward(ctx, dOut):
        TS, Wc = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        _, M, C = TS.shape
        O = Wc.size(1)
        N = g.N
        dOut = dOut.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.zeros_like(Wc)
            db = torch.zeros(O, dtype=F32, device=dOut.device) if ctx.has_bias else None
            gemm_tn_acc(TS, C, M * C, K, C, dOut, O, dW, O, db, M, O)
        if ctx.needs_input_grad[0]:
            G = torch.empty(K, M, C, dtype=F32, device=dOut.device)
            gemm(dOut, O, 0, 1, O, Wc, 1, O, G, C, M * C, C, None, M, K * C)   # G_k = dOut @ W_k^T
            _cheb_stack_bwd(g, G, K, N)
            dX = G[0]
        return dX, dW, db, None, None, None


# --------------------------------------------------------------------------------------------- dense attention scores

class AttentionScoresFunction(torch.autograd.Function):
    """S = softmax_dim1( V . sigmoid( L R + bias ) ) for a batch of score matrices (ASTGCN's SpatialAttention,
    astgcn.py:226-262, and TemporalAttention, :291-328): L [B, n, m], R [B, m, n], bias [1, n, n] or [n, n], V [n, n]
    -> S [B, n, n].  Three launches forward (fused L R + bias + sigmoid; ONE MFMA GEMM for the whole batch on the
    [i][b][j] layout; softmax over dim 1) instead of five torch ops with [B, n, n] temporaries; hand-written backward
    (softmax and sigmoid adjoints, two GEMMs); the two small products with L and R that close the chain run on
    pgt_bmm_f32."""

    @staticmethod
    def forward(ctx, L, R, bias, V):
        lib = _lib.get_lib()
        for t, nm in ((L, "L"), (R, "R"), (bias, "bias"), (V, "V")):
            check_tensor(lib, t, nm)
        Lc, Rc, Vc = L.contiguous(), R.contiguous(), V.contiguous()
        bc = bias.contiguous()
        B, n, m = Lc.shape
        if Rc.shape != (B, m, n) or Vc.shape != (n, n) or bc.numel() != n * n:
            raise ValueError("AttentionScoresFunction: inconsistent operand shapes")
        dev, st = Lc.device, stream_of(lib, Lc)
        sig = torch.empty(n, B, n, dtype=F32, device=dev)
        lib.call("pgt_att_sigmoid_scores_f32", ptr(Lc), ptr(Rc), ptr(bc), B, n, m, ptr(sig), st)
        C = torch.empty(n, B * n, dtype=F32, device=dev)
        gemm(Vc, n, 0, 1, n, sig, B * n, 1, C, B * n, 0, B * n, None, n, B * n)
        S = torch.empty(B, n, n, dtype=F32, device=dev)
        lib.call("pgt_att_softmax_rows_f32", ptr(C), B, n, ptr(S), st)
        ctx.save_for_backward(Lc, Rc, Vc, sig, S)
        ctx.bias_shape = bias.shape
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.get_lib()
        Lc, Rc, Vc, sig, S = ctx.saved_tensors
        B, n, m = Lc.shape
        dev, st = dS.device, stream_of(lib, dS)
        dS = dS.contiguous()
        dC = torch.empty(n, B * n, dtype=F32, device=dev)
        lib.call("pgt_att_softmax_rows_bwd_f32", ptr(S), ptr(dS), B, n, ptr(dC), st)
        dV = None
        if ctx.needs_input_grad[3]:
            dV = torch.empty(n, n, dtype=F32, device=dev)
            gemm(dC, B * n, 0, 1, B * n, sig, 1, B * n, dV, n, 0, n, None, n, n)          # dV = dC sig^T
        dL = dR = dbias = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            Vt = Vc.t().contiguous()
            dsig = torch.empty(n, B * n, dtype=F32, device=dev)
            gemm(Vt, n, 0, 1, n, dC, B * n, 1, dsig, B * n, 0, B * n, None, n, B * n)    # dsig = V^T dC
            dP = torch.empty(B, n, n, dtype=F32, device=dev)
            db = torch.empty(n, n, dtype=F32, device=dev)
            lib.call("pgt_att_sigmoid_bwd_f32", ptr(sig), ptr(dsig), B, n, ptr(dP), ptr(db), st)
            if ctx.needs_input_grad[2]:
                dbias = db.view(ctx.bias_shape)
            if ctx.needs_input_grad[0]:
                dL = _bmm_raw(dP, Rc.transpose(1, 2), torch.empty(B, n, m, dtype=F32, device=dev))     # [B, n, m]
            if ctx.needs_input_grad[1]:
                dR = _bmm_raw(Lc.transpose(1, 2), dP, torch.empty(B, m, n, dtype=F32, device=dev))     # [B, m, n]
        return dL, dR, dbias, dV


# --------------------------------------------------------------------------------------------- EvolveGCN weight evolution

class EvolveWeightFunction(torch.autograd.Function):
    """W_t = GRU(summary(X_t), W_{t-1}) (EvolveGCN-H, evolvegcnh.py:93-100) or GRU(W_{t-1}, W_{t-1}) (EvolveGCN-O,
    evolvegcno.py:185-187) in ONE launch forward and ONE backward (pgt_evolve_weight(_bwd)_f32): top-k scoring / selection,
    the GRU cell on the 8 x 8 state and every gradient.  X may be None (O variant)."""

    @staticmethod
    def forward(ctx, X, p, Wih, Whh, bih, bhh, Wprev, k):
        lib = _lib.get_lib()
        pool = X is not None
        for t, n in ((Wih, "weight_ih"), (Whh, "weight_hh"), (Wprev, "weight")) + (((X, "X"), (p, "select.weight")) if pool else ()):
            check_tensor(lib, t, n)
        F_ = Wprev.size(-1)
        k = int(k)
        Wp = Wprev.reshape(-1, F_).contiguous()
        if Wp.size(0) != k or Wih.shape != (3 * F_, F_) or Whh.shape != (3 * F_, F_):
            raise ValueError("EvolveWeightFunction: the GRU's batch must be the k pooled rows (k == in_channels in the reference)")
        dev = Wp.device
        Xc = X.contiguous() if pool else None
        pc = p.reshape(-1).contiguous() if pool else None
        Wihc, Whhc = Wih.contiguous(), Whh.contiguous()
        Wnew = torch.empty(k, F_, dtype=F32, device=dev)
        perm = torch.empty(k, dtype=I32, device=dev)
        score = torch.empty(k, dtype=F32, device=dev)
        gates = torch.empty(4, k, F_, dtype=F32, device=dev)
        xt = torch.empty(k, F_, dtype=F32, device=dev)
        lib.call("pgt_evolve_weight_f32", ptr(Xc), Xc.stride(0) if pool else 0, Xc.size(0) if pool else 0, ptr(pc), ptr(Wihc),
                 ptr(Whhc), ptr(bih.contiguous() if bih is not None else None), ptr(bhh.contiguous() if bhh is not None else None),
                 ptr(Wp), F_, k, int(pool), ptr(Wnew), ptr(perm), ptr(score), ptr(gates), ptr(xt), stream_of(lib, Wp))
        ctx.save_for_backward(Xc, pc, Wihc, Whhc, Wp, perm, score, gates, xt)
        ctx.pool, ctx.has_bias, ctx.prev_shape, ctx.p_shape = pool, bih is not None, Wprev.shape, (p.shape if pool else None)
        return Wnew

    @staticmethod
    def backward(ctx, dWnew):
        lib = _lib.get_lib()
        Xc, pc, Wihc, Whhc, Wp, perm, score, gates, xt = ctx.saved_tensors
        k, F_ = Wp.shape
        dev = dWnew.device
        dWih, dWhh = torch.empty_like(Wihc), torch.empty_like(Whhc)
        dbih = torch.empty(3 * F_, dtype=F32, device=dev) if ctx.has_bias else None
        dbhh = torch.empty(3 * F_, dtype=F32, device=dev) if ctx.has_bias else None
        dWprev = torch.empty(k, F_, dtype=F32, device=dev)
        dX = torch.zeros_like(Xc) if ctx.pool else None
        dp = torch.empty(F_, dtype=F32, device=dev) if ctx.pool else None
        lib.call("pgt_evolve_weight_bwd_f32", ptr(dWnew.contiguous()), ptr(Xc), Xc.stride(0) if ctx.pool else 0,
                 Xc.size(0) if ctx.pool else 0, ptr(pc), ptr(Wihc), ptr(Whhc), ptr(Wp), ptr(perm), ptr(score), ptr(gates), ptr(xt),
                 F_, k, int(ctx.pool), ptr(dX), dX.stride(0) if ctx.pool else 0, ptr(dp), ptr(dWih), ptr(dWhh), ptr(dbih),
                 ptr(dbhh), ptr(dWprev), stream_of(lib, dWprev))
        return (dX, dp.view(ctx.p_shape) if ctx.pool else None, dWih, dWhh, dbih, dbhh, dWprev.view(ctx.prev_shape), None)


# --------------------------------------------------------------------------------------------- small batched products

def _bmm_raw(A, B, C, accumulate=False):
    """pgt_bmm_f32 on 3-D views [nb, M, K] x [nb, K, N] -> [nb, M, N]; strides are taken as they are (an expanded
    dimension has stride 0, a transposed view swapped strides): no copies."""
    lib = _lib.get_lib()
    for t, n in ((A, "A"), (B, "B"), (C, "C")):
        check_tensor(lib, t, n)
    nb, M, K = A.shape
    N = B.size(2)
    if B.shape != (nb, K, N) or C.shape != (nb, M, N):
        raise ValueError(f"bmm: shapes {tuple(A.shape)} x {tuple(B.shape)} -> {tuple(C.shape)}")
    lib.call("pgt_bmm_f32", ptr(A), *A.stride(), ptr(B), *B.stride(), ptr(C), *C.stride(), nb, M, N, K,
             int(bool(accumulate)), stream_of(lib, C))
    return C


class BmmFunction(torch.autograd.Function):
    """C[b] = A[b] B[b] for small matrices on pgt_bmm_f32 (the embeddings around ASTGCN's attention: astgcn.py:252-256,
    :318-322, :437).  Operands are 3-D views with any strides; gradients come back in the operands' (possibly expanded)
    shapes, so a matrix shared by the batch gets its sum over the batch from `expand`'s own adjoint."""

    @staticmethod
    def forward(ctx, A, B):
        C = torch.empty(A.size(0), A.size(1), B.size(2), dtype=F32, device=A.device)
        _bmm_raw(A, B, C)
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty(A.shape, dtype=F32, device=dC.device)
            _bmm_raw(dC, B.transpose(1, 2), dA)
        if ctx.needs_input_grad[1]:
            dB = torch.empty(B.shape, dtype=F32, device=dC.device)
            _bmm_raw(A.transpose(1, 2), dC, dB)
        return dA, dB


def bmm(A, B):
    """[nb, M, K] x [nb, K, N] (2-D operands are shared by the batch) -> [nb, M, N]."""
    nb = A.size(0) if A.dim() == 3 else B.size(0)
    if A.dim() == 2:
        A = A.unsqueeze(0).expand(nb, -1, -1)
    if B.dim() == 2:
        B = B.unsqueeze(0).expand(nb, -1, -1)
    return BmmFunction.apply(A, B)


class TimeConvResidualNormFunction(torch.autograd.Function):
    """The tail of an ASTGCN block (astgcn.py:463-478): time convolution Conv2d(O -> Ft, (1, 3), stride (1, s), padding
    (0, 1)) of the graph convolution's output + residual Conv2d(Fin -> Ft, (1, 1), stride (1, s)) of the block input ->
    relu -> LayerNorm(Ft), on channels-last rows (b, n, t).

    Xh [B, N, T, O] (already relu'd), Xcl [B, N, T, Fin] -> [B, N, T_out, Ft].  The three taps of the time convolution are
    ONE product on pgt_gemm_f32: the rows live in a buffer with a zero row before and after every (b, n) series, and the
    K-segmented operand takes segment j = the same buffer shifted by j rows (segment stride = one row), so no im2col copy
    exists; the residual convolution accumulates into the same output; relu + LayerNorm pick the strided rows and skip the
    padding rows (pgt_relu_layernorm_f32).  Backward: LayerNorm / relu adjoint, two weight-gradient products on
    pgt_gemm_tn_acc_f32 with the same shifted segments, three accumulating products for the taps' input gradient."""

    @staticmethod
    def forward(ctx, Xh, Xcl, Wt, bt, Wr, br, gamma, beta, stride, eps):
        lib = _lib.get_lib()
        for t, n in ((Xh, "Xh"), (Xcl, "Xcl"), (Wt, "Wt"), (Wr, "Wr"), (gamma, "gamma"), (beta, "beta")):
            check_tensor(lib, t, n)
        B, N, T, O = Xh.shape
        Fin, Ft = Xcl.size(3), Wt.size(0)
        if Xcl.shape != (B, N, T, Fin) or Wt.shape != (Ft, O, 1, 3) or Wr.shape != (Ft, Fin, 1, 1):
            raise ValueError("TimeConvResidualNormFunction: inconsistent operand shapes")
        dev = Xh.device
        BN, Tp, s = B * N, T + 2, int(stride)
        M = BN * Tp
        P = torch.zeros(M + 2, O, dtype=F32, device=dev)          # row (bn, tp) = Xh at t = tp - 1; zero rows at tp = 0, T + 1
        P[:M].view(BN, Tp, O)[:, 1:T + 1].copy_(Xh.reshape(BN, T, O))
        Q = torch.zeros(M, Fin, dtype=F32, device=dev)            # row (bn, tp) = X at t = tp
        Q.view(BN, Tp, Fin)[:, :T].copy_(Xcl.reshape(BN, T, Fin))
        W3 = Wt[:, :, 0, :].permute(2, 1, 0).reshape(3 * O, Ft).contiguous()        # W3[dt * O + o, c] = Wt[c, o, 0, dt]
        WrT = Wr[:, :, 0, 0].t().contiguous()                                        # [Fin, Ft]
        bias = None
        if bt is not None or br is not None:
            bias = (bt if bt is not None else 0) + (br if br is not None else 0)
            bias = bias.contiguous()
        Z = torch.empty(M, Ft, dtype=F32, device=dev)
        gemm(P, O, O, 3, O, W3, Ft, 1, Z, Ft, 0, Ft, bias, M, Ft)                     # segment j = the buffer shifted by j rows
        gemm(Q, Fin, 0, 1, Fin, WrT, Ft, 1, Z, Ft, 0, Ft, None, M, Ft, accumulate=True)
        T_out = (T - 1) // s + 1
        rows = BN * T_out
        Y = torch.empty(rows, Ft, dtype=F32, device=dev)
        stats = torch.empty(rows, 2, dtype=F32, device=dev)
        gc, bc = gamma.contiguous(), beta.contiguous()
        lib.call("pgt_relu_layernorm_f32", ptr(Z), T_out, Tp, s, ptr(gc), ptr(bc), float(eps), rows, Ft, ptr(Y), ptr(stats),
                 stream_of(lib, Z))
        ctx.save_for_backward(P, Q, W3, WrT, Z, stats, gc)
        ctx.dims = (B, N, T, O, Fin, Ft, s, T_out)
        ctx.has_bias = (bt is not None, br is not None)
        return Y.view(B, N, T_out, Ft)

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        P, Q, W3, WrT, Z, stats, gc = ctx.saved_tensors
        B, N, T, O, Fin, Ft, s, T_out = ctx.dims
        dev = dY.device
        BN, Tp = B * N, T + 2
        M, rows = BN * Tp, BN * T_out
        dYc = dY.contiguous().view(rows, Ft)
        dZ = torch.zeros(M, Ft, dtype=F32, device=dev)            # padding / skipped rows carry no gradient
        dgamma, dbeta = torch.zeros(Ft, dtype=F32, device=dev), torch.zeros(Ft, dtype=F32, device=dev)
        lib.call("pgt_relu_layernorm_bwd_f32", ptr(Z), T_out, Tp, s, ptr(gc), ptr(stats), ptr(dYc), rows, Ft, ptr(dZ),
                 ptr(dgamma), ptr(dbeta), stream_of(lib, dZ))
        dW3 = torch.zeros(3 * O, Ft, dtype=F32, device=dev)
        db = torch.zeros(Ft, dtype=F32, device=dev)
        gemm_tn_acc(P, O, O, 3, O, dZ, Ft, dW3, Ft, db, M, Ft)
        dWt = dW3.view(3, O, Ft).permute(2, 1, 0).unsqueeze(2).contiguous()          # [Ft, O, 1, 3]
        dWrT = torch.zeros(Fin, Ft, dtype=F32, device=dev)
        gemm_tn_acc(Q, Fin, 0, 1, Fin, dZ, Ft, dWrT, Ft, None, M, Ft)
        dWr = dWrT.t().contiguous().view(Ft, Fin, 1, 1)
        dXh = dXcl = None
        if ctx.needs_input_grad[0]:
            dP = torch.zeros(M + 2, O, dtype=F32, device=dev)
            for dt in range(3):          # dP[m + dt] += dZ[m] W3[dt]^T : B(k = c, n = o) = W3[dt * O + o, c]
                gemm(dZ, Ft, 0, 1, Ft, W3[dt * O:(dt + 1) * O], 1, Ft, dP[dt:], O, 0, O, None, M, O, accumulate=True)
            dXh = dP[:M].view(BN, Tp, O)[:, 1:T + 1].reshape(B, N, T, O)
        if ctx.needs_input_grad[1]:
            dQ = torch.empty(M, Fin, dtype=F32, device=dev)
            gemm(dZ, Ft, 0, 1, Ft, WrT, 1, Ft, dQ, Fin, 0, Fin, None, M, Fin)          # B(k = c, n = f) = WrT[f, c]
            dXcl = dQ.view(BN, Tp, Fin)[:, :T].reshape(B, N, T, Fin)
        return (dXh, dXcl, dWt, db if ctx.has_bias[0] else None, dWr, db if ctx.has_bias[1] else None, dgamma, dbeta,
                None, None)


# --------------------------------------------------------------------------------------------- attention Chebyshev conv

def spmm_att(csr, S, X3, transpose_s=False):
    """Y[i,b,:] = sum_q val[q] * S[b,i,col[q]] * X3[col[q],b,:]   (S[b,col[q],i] when transpose_s) on [N,B,C]."""
    lib = _lib.get_lib()
    check_tensor(lib, S, "S")
    check_tensor(lib, X3, "X")
    N, B, C = X3.shape
    Y = torch.empty_like(X3)
    lib.call("pgt_spmm_csr_att_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), ptr(S), N, B, C, ptr(X3), ptr(Y),
             int(bool(transpose_s)), stream_of(lib, X3))
    return Y


def sddmm_att(csr, G3, X3, dS):
    """dS[b,i,col[q]] += val[q] * <G3[i,b,:], X3[col[q],b,:]>  (pgt_sddmm_att_f32)."""
    lib = _lib.get_lib()
    N, B, C = X3.shape
    lib.call("pgt_sddmm_att_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), N, B, C, ptr(G3), ptr(X3), ptr(dS),
             stream_of(lib, X3))
    return dS


class ChebConvAttentionFunction(torch.autograd.Function):
    """ChebConvAttention.forward (astgcn.py:112-183) on x [B,N,Fin] — or [B,N,Tt,Fin]: Tt independent calls that share
    the attention (ASTGCNBlock calls the layer once per time step with the same S, astgcn.py:442-452), folded into
    one call — S [B,N,N], W [K,Fin,Fout]:

        T_0 = diag(S[b]) x[b]                         (the reference builds it through a dense eye(N)*S bmm, :159-164)
        T_1 = sum_e norm_e S[b,row_e,col_e] T_0[col_e]  (propagate on the transposed list with Att_norm, :157,169-171)
        T_k = 2 L T_{k-1} - T_{k-2}, k >= 2            (plain norm, :173-177)
        out = sum_k T_k W[k] + bias
    g = SymGraph from pgt_cheb_prep variant 1 (the in-tree __norm__, :82-110)."""

    @staticmethod
    def forward(ctx, x, S, W, bias, g, K):
        lib = _lib.get_lib()
        check_tensor(lib, x, "x")
        check_tensor(lib, S, "spatial_attention")
        squeeze = x.dim() == 3
        if squeeze:
            x = x.unsqueeze(2)
        B, N, Tt, C = x.shape
        if S.shape != (B, N, N) or g.N != N:
            raise ValueError(f"ChebConvAttention: x {tuple(x.shape)}, spatial_attention {tuple(S.shape)}, graph N={g.N}")
        O = W.size(2)
        CC = Tt * C                                                # channels seen by the aggregation
        M = N * B * Tt                                             # rows seen by the feature transform
        Sc = S.contiguous()
        Xnm = swap01(x.contiguous().view(B, N, CC), B, N, CC)      # [N, B, Tt*C]
        d = torch.diagonal(Sc, dim1=1, dim2=2).t().contiguous()    # [N, B]   S[b, i, i]
        TS = torch.empty(K, N, B, CC, dtype=F32, device=x.device)
        torch.mul(Xnm, d.unsqueeze(-1), out=TS[0])
        if K > 1:
            TS[1].copy_(spmm_att(g.fwd, Sc, TS[0]))
        for k in range(2, K):
            spmm(g.fwd, TS[k - 1].view(N, B * CC), TS[k].view(N, B * CC), T=TS[k - 2].view(N, B * CC), alpha=2.0, beta=-1.0)
        Wc = W.contiguous().view(K * C, O)
        out = torch.empty(M, O, dtype=F32, device=x.device)
        gemm(TS, C, M * C, K, C, Wc, O, 1, out, O, 0, O, bias, M, O)
        ctx.g, ctx.K, ctx.dims, ctx.squeeze = g, K, (B, N, Tt, C, O), squeeze
        ctx.has_bias = bias is not None
        ctx.save_for_backward(TS, Wc, Sc, Xnm, d)
        res = swap01(out.view(N, B, Tt * O), N, B, Tt * O).view(B, N, Tt, O)
        return res[:, :, 0] if squeeze else res

    @staticmethod
    def backward(ctx, dOut):
        TS, Wc, Sc, Xnm, d = ctx.saved_tensors
        g, K = ctx.g, ctx.K
        B, N, Tt, C, O = ctx.dims
        CC, M = Tt * C, N * B * Tt
        dev = dOut.device
        if ctx.squeeze:
            dOut = dOut.unsqueeze(2)
        dO = swap01(dOut.contiguous().view(B, N, Tt * O), B, N, Tt * O).view(M, O)   # node-major rows (n, b, t)
        dW = torch.zeros_like(Wc)
        db = torch.zeros(O, dtype=F32, device=dev) if ctx.has_bias else None
        gemm_tn_acc(TS, C, M * C, K, C, dO, O, dW, O, db, M, O)
        G = torch.empty(K, N, B, CC, dtype=F32, device=dev)
        gemm(dO, O, 0, 1, O, Wc, 1, O, G, C, M * C, C, None, M, K * C)
        for k in range(K - 1, 1, -1):                               # adjoint of T_k = 2 L T_{k-1} - T_{k-2}
            Gk, Gp = G[k].view(N, B * CC), G[k - 1].view(N, B * CC)
            spmm(g.bwd, Gk, Gp, T=Gp, alpha=2.0, beta=1.0)
            axpby2d(G[k - 2].view(N * B, CC), G[k].view(N * B, CC), -1.0, G[k - 2].view(N * B, CC), 1.0)
        dS = torch.zeros(B, N, N, dtype=F32, device=dev)
        if K > 1:
            sddmm_att(g.fwd, G[1], TS[0], dS)                       # d/dS of the attention-weighted hop
            G[0].add_(spmm_att(g.bwd, Sc, G[1], transpose_s=True))  # d/dT_0 through the same hop
        # T_0 = d * X : d/dX and the diagonal of d/dS
        dXnm = G[0] * d.unsqueeze(-1)
        dd = (G[0] * Xnm).sum(dim=-1)                               # [N, B]
        torch.diagonal(dS, dim1=1, dim2=2).add_(dd.t())
        dx = swap01(dXnm.contiguous(), N, B, CC).view(B, N, Tt, C)
        if ctx.squeeze:
            dx = dx[:, :, 0]
        return dx, dS, dW.view(K, C, O), db, None, None


# --------------------------------------------------------------------------------------------- LSTM gates

class LSTMGatesFunction(torch.autograd.Function):
    """(H', C') = peephole-LSTM gates of GConvLSTM / GCLSTM from the gate pre-activations P [M, 4*O] = i | f | c | o
    (pgt_lstm_gates_f32 / pgt_lstm_gates_bwd_f32).  w_ci / w_cf / w_co are [1, O] peephole weights or None."""

    @staticmethod
    def forward(ctx, P, C, w_ci, w_cf, w_co):
        lib = _lib.get_lib()
        check_tensor(lib, P, "P")
        check_tensor(lib, C, "C")
        M, O4 = P.shape
        O = O4 // 4
        gates = P.contiguous().clone()          # activated in place by the kernel; P itself stays untouched
        Cc = C.contiguous()
        ws = [None if w is None else w.contiguous().view(-1) for w in (w_ci, w_cf, w_co)]
        Hn = torch.empty(M, O, dtype=F32, device=P.device)
        Cn = torch.empty(M, O, dtype=F32, device=P.device)
        lib.call("pgt_lstm_gates_f32", ptr(gates), ptr(Cc), O, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(Hn), O, ptr(Cn),
                 O, M, O, stream_of(lib, P))
        ctx.save_for_backward(gates, Cc, Cn, *[w for w in ws if w is not None])
        ctx.has_w = tuple(w is not None for w in ws)
        ctx.w_shapes = tuple(None if w is None else tuple(w.shape) for w in (w_ci, w_cf, w_co))
        return Hn, Cn

    @staticmethod
    def backward(ctx, dH, dCn):
        lib = _lib.get_lib()
        saved = list(ctx.saved_tensors)
        gates, Cc, Cn = saved[:3]
        rest = saved[3:]
        ws = []
        for has in ctx.has_w:
            ws.append(rest.pop(0) if has else None)
        M, O = Cc.shape
        dev = gates.device
        dHc = (dH if dH is not None else torch.zeros_like(Cn)).contiguous()
        dCc = None if dCn is None else dCn.contiguous()
        dP = torch.empty_like(gates)
        dC = torch.empty(M, O, dtype=F32, device=dev)
        dw = torch.zeros(3, O, dtype=F32, device=dev) if any(ctx.has_w) else None
        lib.call("pgt_lstm_gates_bwd_f32", ptr(gates), ptr(Cc), O, ptr(Cn), O, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(dHc),
                 O, ptr(dCc), O, ptr(dP), ptr(dC), O, ptr(dw), M, O, stream_of(lib, gates))
        grads = [dw[i].view(ctx.w_shapes[i]) if ctx.has_w[i] else None for i in range(3)]
        return dP, dC, grads[0], grads[1], grads[2]


# --------------------------------------------------------------------------------------------- ST-Conv block: dense halves

class TemporalConvFunction(torch.autograd.Function):
    """TemporalConv.forward (stgcn.py:27-44): H = relu(conv_1(X) * sigmoid(conv_2(X)) + conv_3(X)), three Conv2d(Cin -> Cout,
    (1, k)) over time, on X [B, T, N, Cin] -> [B, T - k + 1, N, Cout] without the reference's permutes: ONE launch forward
    (pgt_tconv_glu_f32: the three convolutions as one row-shifted K-segmented product on the matrix cores, the gate on the
    accumulators).  Backward: the gate's adjoint as one streaming pass (pgt_tconv_glu_bwd_f32) that lays dP | dQ | dR out as
    the operand of ONE weight-gradient product (pgt_gemm_tn_acc_f32, X's taps as K segments) and ONE input-gradient
    product (pgt_gemm_f32, dZ's taps as K segments against the taps in reverse order)."""

    @staticmethod
    def forward(ctx, X, W1, W2, W3, b1, b2, b3):
        lib = _lib.get_lib()
        for t, n in ((X, "X"), (W1, "conv_1.weight"), (W2, "conv_2.weight"), (W3, "conv_3.weight")):
            check_tensor(lib, t, n)
        if X.dim() != 4:
            raise ValueError(f"TemporalConv: X must be [batch, time, nodes, channels], got {tuple(X.shape)}")
        B, T, N, Cin = X.shape
        Cout, k = W1.size(0), W1.size(3)
        if W1.shape != (Cout, Cin, 1, k) or W2.shape != W1.shape or W3.shape != W1.shape:
            raise ValueError("TemporalConv: the three convolutions must be Conv2d(in, out, (1, k)) of one shape")
        if T < k:
            raise RuntimeError(f"TemporalConv: kernel size {k} can't be greater than the {T} time steps")
        Xc = X.contiguous()
        dev = X.device
        # Wp[dt * Cin + ci, g * Cout + c] = conv_{g+1}.weight[c, ci, 0, dt]
        Wp = torch.stack((W1, W2, W3), 0)[:, :, :, 0, :].permute(3, 2, 0, 1).reshape(k * Cin, 3 * Cout).contiguous()
        has_bias = b1 is not None
        bias3 = torch.cat((b1, b2, b3)).contiguous() if has_bias else None
        Tp = T - k + 1
        H = torch.empty(B, Tp, N, Cout, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad)
        P = torch.empty_like(H) if need else None
        S = torch.empty_like(H) if need else None
        work = 4.0 * (B * T * N * Cin + (3 if need else 1) * H.numel() + Wp.numel()) if KERNEL_TIMER else 0
        _timed("tconv", work, lambda: lib.call(
            "pgt_tconv_glu_f32", ptr(Xc), Cin, B, T, N, Cin, Cout, k, ptr(Wp), ptr(bias3), ptr(H), ptr(P), ptr(S),
            stream_of(lib, H)), tag=(B * Tp * N, Cin, Cout, k))
        if need:
            ctx.save_for_backward(Xc, Wp, H, P, S)
        ctx.dims = (B, T, N, Cin, Cout, k)
        ctx.has_bias = has_bias
        return H

    @staticmethod
    def backward(ctx, dH):
        lib = _lib.get_lib()
        Xc, Wp, H, P, S = ctx.saved_tensors
        B, T, N, Cin, Cout, k = ctx.dims
        dev = dH.device
        dHc = dH.contiguous()
        pad, M = (k - 1) * N, B * T * N
        C3 = 3 * Cout
        dZ = torch.empty(pad + M, C3, dtype=F32, device=dev)
        _timed("tconv_bwd", 4.0 * (4 * H.numel() + dZ.numel()) if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_tconv_glu_bwd_f32", ptr(dHc), ptr(H), ptr(P), ptr(S), B, T, N, Cout, k, ptr(dZ), stream_of(lib, dZ)))
        # weight / bias gradients: dWp[dt Cin + ci, :] = sum_m X[m + dt N, ci] dZ[m, :] over the rows that have outputs
        Mw = (B * T - (k - 1)) * N
        dWp = torch.zeros(k * Cin, C3, dtype=F32, device=dev)
        db3 = torch.zeros(C3, dtype=F32, device=dev) if ctx.has_bias else None
        G = dZ[pad:]
        gemm_tn_acc(Xc.view(M, Cin), Cin, N * Cin, k, Cin, G, C3, dWp, C3, db3, Mw, C3)
        dW = dWp.view(k, Cin, 3, Cout).permute(2, 3, 1, 0).unsqueeze(3)                    # [3, Cout, Cin, 1, k]
        dX = None
        if ctx.needs_input_grad[0]:
            # dX[m] = sum_j dZpad[m + j N] Wp[(k - 1 - j) Cin : (k - j) Cin, :]^T   (dZpad = dZ behind (k - 1) N zero rows)
            Wb = Wp.view(k, Cin, C3).flip(0).permute(0, 2, 1).reshape(k * C3, Cin).contiguous()
            dX = torch.empty(B, T, N, Cin, dtype=F32, device=dev)
            gemm(dZ, C3, N * C3, k, C3, Wb, Cin, 1, dX, Cin, 0, Cin, None, M, Cin)
        db = (db3[:Cout], db3[Cout:2 * Cout], db3[2 * Cout:]) if ctx.has_bias else (None, None, None)
        return (dX, dW[0].contiguous(), dW[1].contiguous(), dW[2].contiguous()) + db


def temporal_conv(X, conv_1, conv_2, conv_3):
    """The gate of a TemporalConv module from its three nn.Conv2d parameter holders."""
    return TemporalConvFunction.apply(X, conv_1.weight, conv_2.weight, conv_3.weight, conv_1.bias, conv_2.bias, conv_3.bias)


class BatchNormNodesFunction(torch.autograd.Function):
    """BatchNorm2d(num_nodes) of STConv (stgcn.py:129, :156-159) on [B, T', N, C] in place of permute -> BatchNorm2d ->
    permute: per-node statistics over (batch, time, channel), running statistics updated by the same launch
    (pgt_batchnorm_nodes_f32 / pgt_batchnorm_nodes_bwd_f32: one workgroup per node, deterministic)."""

    @staticmethod
    def forward(ctx, X, gamma, beta, running_mean, running_var, momentum, eps, training):
        lib = _lib.get_lib()
        check_tensor(lib, X, "X")
        B, Tp, N, C = X.shape
        Xc = X.contiguous()
        dev = X.device
        Y = torch.empty_like(Xc)
        stats = torch.empty(N, 2, dtype=F32, device=dev)
        use_batch = bool(training) or running_mean is None
        work = 8.0 * Xc.numel() if KERNEL_TIMER else 0
        _timed("batchnorm", work, lambda: lib.call(
            "pgt_batchnorm_nodes_f32", ptr(Xc), B * Tp, N, C, ptr(gamma), ptr(beta),
            ptr(running_mean) if training else ptr(running_mean if not use_batch else None),
            ptr(running_var) if training else ptr(running_var if not use_batch else None),
            float(momentum), float(eps), int(use_batch), ptr(Y), ptr(stats), stream_of(lib, Y)))
        ctx.save_for_backward(Xc, stats, gamma)
        ctx.use_batch = use_batch
        ctx.mark_non_differentiable(*(t for t in (running_mean, running_var) if t is not None))
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.get_lib()
        Xc, stats, gamma = ctx.saved_tensors
        B, Tp, N, C = Xc.shape
        dev = dY.device
        dYc = dY.contiguous()
        dX = torch.empty_like(Xc) if ctx.needs_input_grad[0] else None
        dg = torch.empty(N, dtype=F32, device=dev) if gamma is not None and ctx.needs_input_grad[1] else None
        dbt = torch.empty(N, dtype=F32, device=dev) if ctx.needs_input_grad[2] else None
        _timed("batchnorm_bwd", 12.0 * Xc.numel() if KERNEL_TIMER else 0, lambda: lib.call(
            "pgt_batchnorm_nodes_bwd_f32", ptr(dYc), ptr(Xc), ptr(stats), ptr(gamma), B * Tp, N, C, int(ctx.use_batch),
            ptr(dX), ptr(dg), ptr(dbt), stream_of(lib, dYc)))
        return dX, dg, dbt, None, None, None, None, None


def batch_norm_nodes(X, bn, training):
    """X [B, T', N, C] through the parameters / buffers of an nn.BatchNorm2d(num_nodes) holder, torch's bookkeeping
    (num_batches_tracked, momentum = None -> cumulative average) on the host."""
    if X.size(2) != bn.num_features:
        raise ValueError(f"STConv: the input has {X.size(2)} nodes, the batch norm was built for {bn.num_features}")
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    if training and X.size(0) * X.size(1) * X.size(3) <= 1:
        raise ValueError("Expected more than 1 value per channel when training")
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return BatchNormNodesFunction.apply(X, bn.weight, bn.bias, rm, rv, momentum, bn.eps, training)
