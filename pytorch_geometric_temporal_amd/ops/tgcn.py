"""TGCN / TGCN2 / A3TGCN: packed weights with the shared weight-gradient deposit, the fused cell Function.
(One family of `pytorch_geometric_temporal_amd.ops`; the package re-exports every name and forwards writes to its switches.)
"""

import os

import torch

from .. import _lib
from .._lib import check_tensor, ptr, stream_of

F32 = torch.float32

from ._core import (FUSE_GATE_EPILOGUES, KERNEL_TIMER, _det_workspace, _gru_h, _gru_h_bwd, _gru_zr, _gru_zr_bwd, _rows, _timed, add2d, axpby2d, copy2d, gemm, gemm_gru_h, gemm_gru_zr, gemm_tn_acc, spmm, swap01)
from ._graphs import tensor_version


# widest hidden state the one-workgroup-per-sample sequence kernels are used for (beyond it the in-kernel scalar products lose to
# the MFMA path even where the state would still fit the LDS); PGT_SEQ_SMALL=0 turns them off (A/B)
# T-GCN cell at hidden width 32: the fused one-launch forward / adjoint (csrc/tgcn_cell.hip); PGT_TGCN_FUSED=0 = the two
# fused-epilogue products + gate kernels (A/B, and the path of every other width)
USE_TGCN_FUSED = os.environ.get("PGT_TGCN_FUSED", "1") != "0"


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
