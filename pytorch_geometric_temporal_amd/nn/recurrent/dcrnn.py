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


from .._states import _StatesTensor, _plain, _interception_allowed, packed_once  # noqa: E402,F401  (the routed read-out: nn/_states.py)


def _cell_weights(conv_z, conv_r, conv_h):
    """The stacked operands of the two gate products from the three convolutions' parameters: one launch
    (ops.CellWeightsFunction) instead of three weight re-stackings and two concatenations."""
    params = (conv_z.weight, conv_r.weight, conv_h.weight, conv_z.bias, conv_r.bias)
    # (a per-snapshot loop calls the cell with the same parameters every time: packed once per training step, nn/_states.py)
    Wzr, bzr, Wh = packed_once(conv_z, params, lambda: ops.CellWeightsFunction.apply(*params), ops.CellWeightsFunction.repack)
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
        if ops.USE_SEQ64 and B >= ops.SEQ64_MIN_BATCH and X.dtype == torch.float32 and ops.seq64_fits(g, Fin, O, self.K):
            # hidden width 64 on a graph whose block fits a CU's LDS (the benchmarked BatchedDCRNN(2, 64, K = 3) on METR-LA's 207
            # sensors): all T steps of every sample in ONE launch, the diffusion terms and the products never leave the CU
            # (csrc/seq64.hip); the backward pass is the general path's BPTT on what the launch saved
            return self._states(ops.DCRNNSeq64Function.apply(X, None, Wzr, bzr, Wh, bh, g, self.K))
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
