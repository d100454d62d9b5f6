"""DConv / DCRNN / BatchedDConv / BatchedDCRNN — drop-in mirrors of
torch_geometric_temporal/nn/recurrent/dcrnn.py (reference file:line cited per member), with the whole
message-passing path on hand-written gfx950 kernels behind the C ABI (include/pgt_hip.h).

Same class names, constructor arguments, parameter names/shapes (state_dict compatible:
`conv_x_{z,r,h}.weight [2, K, in+out, out]`, `.bias [out]`), forward signatures and return shapes.
"""
import torch

from ... import ops


class DConv(torch.nn.Module):
    r"""Diffusion convolution (reference: dcrnn.py:7-111).

    Args mirror the reference: in_channels, out_channels, K, bias=True.
    The reference's quirks are reproduced: edge weights only enter through the degrees,
    `norm_in` is indexed by `row` and applied positionally to the re-sorted reverse edge list, `Tx_0` is never
    advanced.  Deliberate divergence: `bias=False` works (the reference crashes in `__reset_parameters`,
    dcrnn.py:37).
    """

    def __init__(self, in_channels, out_channels, K, bias=True):
        super().__init__()
        assert K > 0
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.weight = torch.nn.Parameter(torch.empty(2, K, in_channels, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._reset_parameters()

    def _reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    # dense-adjacency semantics of dcrnn.py:59-77 (duplicates / zero weights make the reference fail)
    _strict_dense = True

    def forward(self, X, edge_index, edge_weight=None):
        """X [num_nodes, in_channels], edge_index [2,E] int64, edge_weight [E] or None -> [num_nodes, out_channels]
        (reference: dcrnn.py:42-111)."""
        K = self.weight.size(1)
        # K = 1 never touches the reversed edge list in the reference (dcrnn.py:85 onwards is skipped), so duplicate
        # edges / zero weights are accepted there
        g = ops.dconv_graph(edge_index, edge_weight, X.size(0), strict_dense=self._strict_dense and K > 1)
        return ops.DConvFunction.apply(X.contiguous(), ops.stack_weight(self.weight), self.bias, g, K, 1)


class BatchedDConv(DConv):
    r"""Reference: dcrnn.py:222-325.  Takes the (already replicated) block-diagonal graph; degrees by
    scatter-add, reverse list sorted by `col * num_nodes + row` — the same operators `DConv` builds, without the
    dense-path restrictions.  `cached_idx` is accepted for signature compatibility: graph preparation is cached
    by tensor identity on every call path (ops.GRAPH_CACHE)."""

    _strict_dense = False

    def forward(self, X, edge_index, edge_weight, cached_idx=False):
        return super().forward(X, edge_index, edge_weight)


class _StatesTensor(torch.Tensor):
    """What `BatchedDCRNN.forward` returns: a plain tensor in every respect but one — the reference's examples feed the
    `[B, T, N, out]` states to a per-node read-out `torch.nn.Linear(out, 1 … 4)` (examples/indexBatching/DCRNN/*_main.py,
    examples/recurrent/dcrnn_example.py:24-31), and for 1 – 4 output features over millions of rows the BLAS library
    behind `F.linear` picks a pathological tile (≈ 5 ms per call at 2.5 M rows against 0.14 ms for one streaming pass:
    bench.py `variants.dropin_default`).  `F.linear(states, weight, bias)` with a skinny fp32 weight is therefore routed to
    this package's streaming kernels (same parameters, same arithmetic type, ordinary autograd) — also when a `relu` stands
    between the states and the read-out, as in the reference's own models; every other operation — and `F.linear` with any
    other operand, or under autocast / tracing / torch.compile — runs as usual and returns plain tensors.  Pickling stores a
    plain tensor.
    `BatchedDCRNN.readout_interception = False` hands out plain tensors instead."""

    _RELUS = (torch.nn.functional.relu, torch.relu, torch.Tensor.relu)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear and not kwargs and 2 <= len(args) <= 3 and _interception_allowed(args[0]):
            x, w = args[0], args[1]
            b = args[2] if len(args) == 3 else None
            if (type(x) is _StatesTensor and type(w) in (torch.Tensor, torch.nn.Parameter) and w.dim() == 2 and
                    1 <= w.size(0) <= 4 and w.size(1) == x.size(-1) and x.dtype == w.dtype == torch.float32 and
                    x.device == w.device and (b is None or (type(b) in (torch.Tensor, torch.nn.Parameter) and
                                                            b.dtype == torch.float32 and b.device == w.device))):
                from ..conv import _rows_in_memory_order
                x2, restore = _rows_in_memory_order(x.as_subclass(torch.Tensor))
                return restore(ops.linear(x2, w.t(), b))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        # the reference's own models put a relu between the recurrent layer and the read-out (examples/recurrent/
        # dcrnn_example.py:27-28, examples/indexBatching/tgcn/metr_la_main.py:43-44): relu(states) is still "the states" for the
        # one purpose of this class, so the read-out that follows is routed as well
        if func in cls._RELUS and not kwargs.get("inplace", False) and isinstance(out, torch.Tensor) and \
                type(args[0]) is _StatesTensor:
            return out.as_subclass(_StatesTensor)
        return _plain(out)

    def __reduce_ex__(self, proto):
        # torch.save / pickle of a result stores a plain tensor: the subclass is a routing hint, not data
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


def _interception_allowed(x):
    """The routed read-out returns fp32 from the package's kernels: under autocast stock torch would return the autocast dtype,
    and a tracer / compiler should see the stock op — in those contexts the call is left alone."""
    dev = x.device.type if isinstance(x, torch.Tensor) else "cuda"
    if torch.is_autocast_enabled(dev) or torch.jit.is_tracing():
        return False
    comp = getattr(torch, "compiler", None)
    return not (comp is not None and comp.is_compiling())


def _plain(out):
    if type(out) is _StatesTensor:
        return out.as_subclass(torch.Tensor)
    if type(out) in (tuple, list):               # (torch.Size and other non-tensor results go back untouched)
        return type(out)(_plain(o) for o in out)
    return out


def _cell_weights(conv_z, conv_r, conv_h):
    """The stacked operands of the two gate products from the three convolutions' parameters: one launch
    (ops.CellWeightsFunction) instead of three weight re-stackings and two concatenations."""
    Wzr, bzr, Wh = ops.CellWeightsFunction.apply(conv_z.weight, conv_r.weight, conv_h.weight, conv_z.bias, conv_r.bias)
    return Wzr, bzr, Wh, conv_h.bias


class DCRNN(torch.nn.Module):
    r"""Diffusion convolutional GRU cell (reference: dcrnn.py:114-219).  One call = one fused cell step:
    both gate convolutions share one aggregation of [X, H] (the reference aggregates it twice)."""

    _conv_cls = DConv

    def __init__(self, in_channels: int, out_channels: int, K: int, bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.bias = bias
        self._create_parameters_and_layers()

    def _create_parameters_and_layers(self):
        c = self.in_channels + self.out_channels
        self.conv_x_z = self._conv_cls(c, self.out_channels, self.K, self.bias)
        self.conv_x_r = self._conv_cls(c, self.out_channels, self.K, self.bias)
        self.conv_x_h = self._conv_cls(c, self.out_channels, self.K, self.bias)

    def _set_hidden_state(self, X, H):
        if H is None:
            H = torch.zeros(X.shape[0], self.out_channels, device=X.device, dtype=X.dtype)
        return H

    def forward(self, X, edge_index, edge_weight=None, H=None):
        """X [N, in], edge_index [2,E], edge_weight [E]|None, H [N, out]|None -> H' [N, out] (dcrnn.py:194-219)."""
        if self.K == 1 and X.dim() == 2 and ops.cell_k1_fits(X.size(0), self.in_channels, self.out_channels):
            # no diffusion: the reference's graph preparation feeds nothing (dcrnn.py:79-82) and the cell is dense -- one
            # launch, H = None as a null pointer (csrc/small_cell.hip; BASELINE configs[0])
            cz, cr, ch = self.conv_x_z, self.conv_x_r, self.conv_x_h
            return ops.DCRNNCellK1Function.apply(X, H, cz.weight, cr.weight, ch.weight, cz.bias, cr.bias, ch.bias)
        g = ops.dconv_graph(edge_index, edge_weight, X.size(0), strict_dense=self.K > 1)
        Wzr, bzr, Wh, bh = _cell_weights(self.conv_x_z, self.conv_x_r, self.conv_x_h)
        if ops.USE_SEQ_SMALL and X.dim() == 2 and self._one_workgroup(g):
            # a small graph with K > 1 (test/recurrent_test.py:274-315, Chickenpox at K = 2, 3): stack, both gate products and the
            # blend in ONE workgroup, one launch each way (csrc/seq_small.hip) instead of ~10 launches per snapshot
            out = ops.DCRNNSeqSmallFunction.apply(X.view(1, 1, X.size(0), X.size(1)), None if H is None else H.unsqueeze(0),
                                                  Wzr, bzr, Wh, bh, g, self.K)
            return out[0, 0]
        H = self._set_hidden_state(X, H)
        out = ops.DCRNNSeqFunction.apply(X.contiguous().unsqueeze(0), H, Wzr, bzr, Wh, bh, g, self.K, 1)
        return out[0]

    def _one_workgroup(self, g):
        """One cell step of one sample: worth a single workgroup while its scalar products stay small (N S C 3O multiply-adds)."""
        C = self.in_channels + self.out_channels
        work = g.N * (2 * self.K - 1) * C * 3 * self.out_channels
        return work <= 4_000_000 and ops.seq_small_fits(g, self.in_channels, self.out_channels, self.K)


class BatchedDCRNN(torch.nn.Module):
    r"""Batched seq-to-seq DCRNN (reference: dcrnn.py:328-475): X [B, T, N, F] -> [B, T, N, out], hidden state
    starts at zero every forward.  The B copies of the graph are never materialised (the reference builds a
    B-times replicated edge list in a Python loop, dcrnn.py:363-369): rows are laid out node-major [N][B][C] so one
    aggregation launch covers the whole batch with B*C-float coalesced neighbour reads."""

    # a skinny `torch.nn.Linear` read-out applied to the result runs on this package's streaming kernels (_StatesTensor)
    readout_interception = True

    def __init__(self, in_channels: int, out_channels: int, K: int, bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.bias = bias
        c = in_channels + out_channels
        self.conv_x_z = BatchedDConv(c, out_channels, K, bias)
        self.conv_x_r = BatchedDConv(c, out_channels, K, bias)
        self.conv_x_h = BatchedDConv(c, out_channels, K, bias)
        # The result is the reference's contiguous [B, T, N, O] tensor (torch.stack(outputs, dim=1), dcrnn.py:463-475): the
        # candidate-gate epilogue of every step stores its H_t straight into out[:, t] (ops.DCRNNSeqFunction, btno) and the
        # backward pass reads the incoming gradient in that layout -- no transposition pass in either direction.

    def forward(self, X, edge_index, edge_weight):
        B, T, N, Fin = X.shape
        if Fin != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input features, got {Fin}")
        g = ops.dconv_graph(edge_index, edge_weight, N, strict_dense=False)
        Wzr, bzr, Wh, bh = _cell_weights(self.conv_x_z, self.conv_x_r, self.conv_x_h)
        O = self.out_channels
        if ops.USE_SEQ_SMALL and O <= ops.SEQ_SMALL_MAX_O and ops.seq_small_fits(g, Fin, O, self.K):
            # narrow states on a small graph (the reference's own BatchedDCRNN(2, 2, K = 3), pems_ddp.py:80): the whole 12-step
            # sequence of a sample in one workgroup, ONE launch forward and one backward (csrc/seq_small.hip)
            return self._states(ops.DCRNNSeqSmallFunction.apply(X, None, Wzr, bzr, Wh, bh, g, self.K))
        if ops.slab_fits(g, Fin + O, self.K):
            # small graph: batch-major rows m = b*N + n; every diffusion stack is ONE LDS-resident launch.
            # [B][T][N*F] -> [T][B][N*F]
            Xbm = ops.Swap01.apply(X.contiguous().view(B, T, N * Fin), B, T, N * Fin).view(T, B * N, Fin)
            H0 = torch.zeros(B * N, O, device=X.device, dtype=X.dtype)
            return self._states(ops.DCRNNSeqFunction.apply(Xbm, H0, Wzr, bzr, Wh, bh, g, self.K, B, True, True))   # [B, T, N, O]
        # [B][T*N][F] -> [T*N][B][F]  (node-major rows m = n*B + b per step: one aggregation launch per hop)
        Xnm = ops.Swap01.apply(X.contiguous().view(B, T * N, Fin), B, T * N, Fin).view(T, N * B, Fin)
        H0 = torch.zeros(N * B, O, device=X.device, dtype=X.dtype)
        return self._states(ops.DCRNNSeqFunction.apply(Xnm, H0, Wzr, bzr, Wh, bh, g, self.K, B, False, True))   # [B, T, N, O]

    def _states(self, out):
        return out.as_subclass(_StatesTensor) if self.readout_interception else out
