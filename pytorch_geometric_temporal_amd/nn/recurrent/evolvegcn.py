"""EvolveGCN-H / EvolveGCN-O and GCNConv_Fixed_W — drop-in mirrors of
torch_geometric_temporal/nn/recurrent/evolvegcnh.py and evolvegcno.py.

The per-snapshot work that scales with the graph — gcn_norm of the (changing) edge list and the aggregation
A_hat (X W_t) — runs on the HIP kernels (graph prep on the device, one aggregation launch, MFMA GEMM for X W_t).
The weight evolution W_t = GRU(., W_{t-1}) acts on an F x F matrix (8 x 8 in the reference's example): it stays a
torch.nn.GRU so that `recurrent_layer.*` keeps the reference's parameter names and initialisation.
"""
import torch

from ... import ops
from ..conv import TopKPooling, glorot_


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
        X_tilde = self.pooling_layer(X, edge_index)
        X_tilde = X_tilde[0][None, :, :]
        h0 = self.initial_weight if self.weight is None else self.weight
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
        _, self.weight = self.recurrent_layer(w, w)
        return self.conv_layer(self.weight.squeeze(dim=0), X, edge_index, edge_weight)
