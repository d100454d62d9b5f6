"""A3TGCN / A3TGCN2 — drop-in mirrors of torch_geometric_temporal/nn/recurrent/attentiontemporalgcn.py.

Every period uses the SAME hidden state H (attentiontemporalgcn.py:75-79, :151-157), so the `periods` T-GCN calls of
the reference are independent: they are folded into the batch dimension and run as ONE fused cell call (one
aggregation launch at width B*periods*in instead of 3*periods launches at width out), followed by the softmax-weighted
sum over periods.
"""
import torch

from ... import ops
from .temporalgcn import TGCN, TGCN2, _cell


class A3TGCN(torch.nn.Module):
    r"""Attention Temporal GCN (reference: attentiontemporalgcn.py:7-80).  X [N, in, periods] -> [N, out].
    Deliberate divergence: `_attention` is created on the default device and follows `.to()`, instead of being
    pinned to cuda at construction (:48-49)."""

    def __init__(self, in_channels: int, out_channels: int, periods: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.periods = periods
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self._setup_layers()

    def _setup_layers(self):
        self._base_tgcn = TGCN(self.in_channels, self.out_channels, improved=self.improved, cached=self.cached,
                               add_self_loops=self.add_self_loops)
        self._attention = torch.nn.Parameter(torch.empty(self.periods))
        torch.nn.init.uniform_(self._attention)

    def forward(self, X, edge_index, edge_weight=None, H=None):
        N, Fin, P = X.shape
        O = self.out_channels
        base = self._base_tgcn
        if H is None:
            H = torch.zeros(N, O, device=X.device, dtype=X.dtype)
        g = base._graph(edge_index, edge_weight, N)
        probs = torch.nn.functional.softmax(self._attention, dim=0)
        Xnm = X.permute(0, 2, 1).reshape(N * P, Fin)                  # rows m = n*P + p
        Hnm = H.unsqueeze(1).expand(N, P, O).reshape(N * P, O)
        Hn = _cell(base, Xnm, Hnm, g, P).view(N, P, O)
        return (Hn * probs.view(1, P, 1)).sum(dim=1)


class A3TGCN2(torch.nn.Module):
    r"""Batched A3T-GCN (reference: attentiontemporalgcn.py:83-157).  X [B, N, in, periods] -> [B, N, out]."""

    def __init__(self, in_channels: int, out_channels: int, periods: int, batch_size: int, improved: bool = False,
                 cached: bool = False, add_self_loops: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.periods = periods
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self.batch_size = batch_size
        self._setup_layers()

    def _setup_layers(self):
        self._base_tgcn = TGCN2(self.in_channels, self.out_channels, self.batch_size, improved=self.improved,
                                cached=self.cached, add_self_loops=self.add_self_loops)
        self._attention = torch.nn.Parameter(torch.empty(self.periods))
        torch.nn.init.uniform_(self._attention)

    def forward(self, X, edge_index, edge_weight=None, H=None):
        B, N, Fin, P = X.shape
        O = self.out_channels
        base = self._base_tgcn
        if H is None:
            H = torch.zeros(B, N, O, device=X.device, dtype=X.dtype)
        g = base._graph(edge_index, edge_weight, N)
        probs = torch.nn.functional.softmax(self._attention, dim=0)
        Xnm = X.permute(1, 0, 3, 2).reshape(N * B * P, Fin)           # rows m = (n*B + b)*P + p
        Hnm = H.permute(1, 0, 2).unsqueeze(2).expand(N, B, P, O).reshape(N * B * P, O)
        Hn = _cell(base, Xnm, Hnm, g, B * P).view(N, B, P, O)
        return (Hn * probs.view(1, 1, P, 1)).sum(dim=2).permute(1, 0, 2).contiguous()
