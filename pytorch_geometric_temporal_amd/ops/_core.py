"""Kernel timer, raw kernel calls on torch-owned device memory (aggregation, products, element-wise), and the schedule switches.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""
import ctypes

import os

import torch

from .. import _lib
from .._lib import RowMapStruct, check_tensor, ptr, stream_of

F32 = torch.float32

from ._graphs import Ellw, LONG_ROW, RenumberedEllw, USE_ELLW, _window_kernel_covers, ellw_of


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
        if op is not None and op.left_out and X.size(1) > 256:
            op = None                  # (node-major batches: the hub kernel covers 64 lanes x 4 floats; the wide CSR kernel spreads a row anyway)
    work = spmm_algorithmic_bytes(csr.n_rows, csr.col.numel(), X.size(1), T is not None) if KERNEL_TIMER else 0
    if op is not None:
        es = op.struct()
        lay = op.csr or csr
        hubs = csr.left_rows if op.left_out and not (op.hub_col is not None and X.size(1) == 64) else None    # (folded at F = 64)

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
    plan_len = csr.short_len if getattr(csr, "left_rows", None) is not None else csr.max_len     # left-out rows (hubs)
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
