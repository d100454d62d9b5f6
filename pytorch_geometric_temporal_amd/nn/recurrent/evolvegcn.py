"""EvolveGCN-H / EvolveGCN-O and GCNConv_Fixed_W — drop-in mirrors of
torch_geometric_temporal/nn/recurrent/evolvegcnh.py and evolvegcno.py.

The per-snapshot work that scales with the graph — gcn_norm of the (changing) edge list and the aggregation
A_hat (X W_t) — runs on the HIP kernels (graph prep on the device, one aggregation launch, MFMA GEMM for X W_t).
The weight evolution W_t = GRU(., W_{t-1}) acts on an F x F matrix (8 x 8 in the reference's example): the modules stay a
TopKPooling / torch.nn.GRU pair so that `pooling_layer.*` / `recurrent_layer.*` keep the reference's parameter names and
initialisation, but the arithmetic — scoring, top-k, the GRU cell and every gradient — is ONE launch each way
(csrc/evolve.hip) instead of ~25 / ~40 torch and MIOpen launches per snapshot.
"""
import torch

from ... import ops
from ..conv import TopKPooling, glorot_


_POOLED_ROWS = {}


def _pooled_rows(ratio, n, dtype):
    """k of TopKPooling: an int ratio as it is, else ceil(ratio * N) computed in the score dtype (PyG); remembered per
    (ratio, N, dtype) — the tensor arithmetic costs more host time than the weight evolution's launch."""
    if isinstance(ratio, int):
        return min(ratio, n)
    key = (float(ratio), int(n), dtype)
    k = _POOLED_ROWS.get(key)
    if k is None:
        k = _POOLED_ROWS[key] = int((float(ratio) * torch.tensor(n).to(dtype)).ceil().to(torch.long))
    return k


def _fused_evolution_applies(X, in_channels, k, pooled=True):
    """The one-launch weight evolution covers the reference's shapes: fp32, the GRU's batch = the k pooled rows = in_channels
    <= 64, at most 4096 nodes; anything else keeps the module path (TopKPooling + torch.nn.GRU)."""
    return (X.dtype == torch.float32 and k == in_channels and 1 <= in_channels <= 64 and X.dim() == 2 and
            (not pooled or (k <= X.size(0) <= 4096)))


class GCNConv_Fixed_W(torch.nn.Module):
    r"""GCN convolution with an externally supplied weight (reference: evolvegcno.py:13-101):
    out = A_hat (x W),  A_hat = gcn_norm(edge_index, edge_weight).  The reference's "cache" is dead code
    (`_cached_edge_index` is never set, :85-90), so the normalisation is recomputed whenever the graph tensors change —
    here through the identity-keyed graph cache."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = True, normalize: bool = True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self.normalize = normalize

    def reset_parameters(self):
        pass

    def forward(self, W, x, edge_index, edge_weight=None):
        if (x.dim() == 2 and W.dim() == 2 and x.dtype == torch.float32 and edge_index.dim() == 2 and
                ops.gcn_small_fits(x.size(0), edge_index.size(1), W.size(0), W.size(1))):
            # a graph that fits one workgroup's LDS: normalisation, lists, product and aggregation from the raw edge list in
            # ONE launch (csrc/small_gcn.hip) -- a new edge list per snapshot costs no device preparation
            return ops.gcn_small(x, W, edge_index, edge_weight, self.improved, self.add_self_loops, self.normalize)
        if self.normalize:
            g = ops.gcn_graph(edge_index, edge_weight, x.size(-2), self.improved, self.add_self_loops)
        else:                                      # the edge list as it is: no gcn_norm, no self-loops (:83-90 skipped)
            g = ops.raw_graph(edge_index, edge_weight, x.size(-2))
        h = ops.linear(x, W, None)                 # x @ W  (evolvegcno.py:92)
        return ops.propagate(g, h)


class EvolveGCNH(torch.nn.Module):
    r"""EvolveGCN-H (reference: evolvegcnh.py:8-102): top-k summary of X_t drives a GRU that evolves the GCN weight.
    Hidden state `self.weight` is carried across calls; `reinitialize_weight()` resets it (:56)."""

    def __init__(self, num_of_nodes: int, in_channels: int, improved: bool = False, cached: bool = False,
                 normalize: bool = True, add_self_loops: bool = True):
        super().__init__()
        self.num_of_nodes = num_of_nodes
        self.in_channels = in_channels
        self.improved = improved
        self.cached = cached
        self.normalize = normalize
        self.add_self_loops = add_self_loops
        self.weight = None
        self.initial_weight = torch.nn.Parameter(torch.empty(1, in_channels, in_channels))
        self._create_layers()
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.initial_weight)

    def reinitialize_weight(self):
        self.weight = None

    def _create_layers(self):
        self.ratio = self.in_channels / self.num_of_nodes
        self.pooling_layer = TopKPooling(self.in_channels, self.ratio)
        self.recurrent_layer = torch.nn.GRU(input_size=self.in_channels, hidden_size=self.in_channels, num_layers=1)
        self.conv_layer = GCNConv_Fixed_W(self.in_channels, self.in_channels, improved=self.improved,
                                          cached=self.cached, normalize=self.normalize,
                                          add_self_loops=self.add_self_loops)

    def forward(self, X, edge_index, edge_weight=None):
        h0 = self.initial_weight if self.weight is None else self.weight
        k = _pooled_rows(self.pooling_layer.ratio, X.size(0), X.dtype)
        if _fused_evolution_applies(X, self.in_channels, k):
            # top-k summary -> GRU cell -> W_t, and every gradient, in one launch each way (ops.EvolveWeightFunction); the
            # modules keep the reference's parameters (pooling_layer.select.weight, recurrent_layer.*_l0)
            rl = self.recurrent_layer
            self.weight = ops.EvolveWeightFunction.apply(X, self.pooling_layer.select.weight, rl.weight_ih_l0, rl.weight_hh_l0,
                                                         getattr(rl, "bias_ih_l0", None), getattr(rl, "bias_hh_l0", None),
                                                         h0, k).unsqueeze(0)
        else:
            X_tilde = self.pooling_layer(X, edge_index)
            X_tilde = X_tilde[0][None, :, :]
            _, self.weight = self.recurrent_layer(X_tilde, h0)
        return self.conv_layer(self.weight.squeeze(dim=0), X, edge_index, edge_weight)


class EvolveGCNO(torch.nn.Module):
    r"""EvolveGCN-O (reference: evolvegcno.py:105-191): the GCN weight is both input and hidden state of the GRU."""

    def __init__(self, in_channels: int, improved: bool = False, cached: bool = False, normalize: bool = True,
                 add_self_loops: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.improved = improved
        self.cached = cached
        self.normalize = normalize
        self.add_self_loops = add_self_loops
        self.initial_weight = torch.nn.Parameter(torch.empty(1, in_channels, in_channels))
        self.weight = None
        self._create_layers()
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.initial_weight)

    def reinitialize_weight(self):
        self.weight = None

    def _create_layers(self):
        self.recurrent_layer = torch.nn.GRU(input_size=self.in_channels, hidden_size=self.in_channels, num_layers=1)
        self.conv_layer = GCNConv_Fixed_W(self.in_channels, self.in_channels, improved=self.improved,
                                          cached=self.cached, normalize=self.normalize,
                                          add_self_loops=self.add_self_loops)

    def forward(self, X, edge_index, edge_weight=None):
        w = self.initial_weight if self.weight is None else self.weight
        if _fused_evolution_applies(X, self.in_channels, self.in_channels, pooled=False):
            rl = self.recurrent_layer
            self.weight = ops.EvolveWeightFunction.apply(None, None, rl.weight_ih_l0, rl.weight_hh_l0,
                                                         getattr(rl, "bias_ih_l0", None), getattr(rl, "bias_hh_l0", None),
                                                         w, self.in_channels).unsqueeze(0)
        else:
            _, self.weight = self.recurrent_layer(w, w)
        return self.conv_layer(self.weight.squeeze(dim=0), X, edge_index, edge_weight)
