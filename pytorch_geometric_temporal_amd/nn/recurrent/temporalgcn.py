"""TGCN / TGCN2 — drop-in mirrors of torch_geometric_temporal/nn/recurrent/temporalgcn.py.

Same constructor arguments, parameter names (`conv_{z,r,h}.lin.weight [out,in]`, `conv_{z,r,h}.bias`,
`linear_{z,r,h}.{weight [out, 2*out], bias}`), forward signatures and outputs; the cell is ONE fused call
(ops.TGCNCellFunction): one aggregation of X at the INPUT width shared by the three gates, `lin` and `linear` folded into
one product per gate pair with the gate chains in the GEMM epilogues (ops.TGCNWeightsFunction builds the folded operands
from the module's parameters in one launch), hand-written backward on the transposed operator.
"""
import torch

from ... import ops
from .._states import _StatesTensor, packed_once
from ..conv import GCNConv


def _cell(mod, X2, H2, g, Bt, batch_major=False):
    """One cell step on [num_nodes * Bt, .] rows (node-major, or batch-major as TGCN2 holds them)."""
    cz, cr, ch = mod.conv_z, mod.conv_r, mod.conv_h
    params = (cz.lin.weight, cr.lin.weight, ch.lin.weight, cz.bias, cr.bias, ch.bias,
              mod.linear_z.weight, mod.linear_r.weight, mod.linear_h.weight, mod.linear_z.bias, mod.linear_r.bias, mod.linear_h.bias)
    # (one packing per training step, not per time step of the caller's loop: nn/_states.py packed_once)
    Wzr, bzr, Wh, bh = packed_once(mod, params, lambda: ops.TGCNWeightsFunction.apply(*params), ops.TGCNWeightsFunction.repack)
    return ops.TGCNCellFunction.apply(X2, H2, Wzr, bzr, Wh, bh, g, Bt, batch_major)


class TGCN(torch.nn.Module):
    r"""Temporal Graph Convolutional GRU cell (reference: temporalgcn.py:5-130).

    Args: in_channels, out_channels, improved=False, cached=False, add_self_loops=True."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self._create_parameters_and_layers()

    def _create_parameters_and_layers(self):
        for gate in ("z", "r", "h"):      # temporalgcn.py:38-70
            setattr(self, "conv_" + gate, GCNConv(self.in_channels, self.out_channels, improved=self.improved,
                                                  cached=self.cached, add_self_loops=self.add_self_loops))
            setattr(self, "linear_" + gate, torch.nn.Linear(2 * self.out_channels, self.out_channels))

    def _graph(self, edge_index, edge_weight, num_nodes):
        # the three convs share one normalisation (same arguments); `cached` freezes it on the first call (PyG)
        return self.conv_z.graph(edge_index, edge_weight, num_nodes)

    def _set_hidden_state(self, X, H):
        if H is None:
            H = torch.zeros(X.shape[0], self.out_channels, device=X.device, dtype=X.dtype)
        return H

    def forward(self, X, edge_index, edge_weight=None, H=None):
        """X [N, in], edge_index [2,E], edge_weight [E]|None, H [N, out]|None -> H' [N, out] (temporalgcn.py:104-130)."""
        H = self._set_hidden_state(X, H)
        g = self._graph(edge_index, edge_weight, X.size(0))
        return _cell(self, X, H, g, 1)


class TGCN2(TGCN):
    r"""Batched T-GCN cell (reference: temporalgcn.py:133-233): X [B, N, in], H [B, N, out] -> [B, N, out].
    `batch_size` is kept for signature compatibility (the reference ignores it too, :148).
    The result is the states tensor of nn/_states.py: a plain `[B, N, out]` tensor whose skinny `torch.nn.Linear` read-out —
    directly or behind a relu, as in the reference's index-batching example (examples/indexBatching/tgcn/metr_la_main.py:43-45) —
    runs on the package's streaming kernels; `TGCN2.readout_interception = False` hands out plain tensors."""

    readout_interception = True

    def __init__(self, in_channels: int, out_channels: int, batch_size: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True):
        self.batch_size = batch_size
        super().__init__(in_channels, out_channels, improved, cached, add_self_loops)

    def _set_hidden_state(self, X, H):
        if H is None:
            H = torch.zeros(X.shape[0], X.shape[1], self.out_channels, device=X.device, dtype=X.dtype)
        return H

    def forward(self, X, edge_index, edge_weight=None, H=None):
        H = self._set_hidden_state(X, H)
        B, N, Fin = X.shape
        O = self.out_channels
        g = self._graph(edge_index, edge_weight, N)
        # rows stay batch-major (m = b*N + n), as the caller holds X and H: only the `in_channels` input columns are taken to
        # the node-major layout of the aggregation (one launch for the whole batch) and back; H is never transposed
        out = _cell(self, X.reshape(B * N, Fin), H.reshape(B * N, O), g, B, batch_major=True).view(B, N, O)
        return out.as_subclass(_StatesTensor) if self.readout_interception else out
