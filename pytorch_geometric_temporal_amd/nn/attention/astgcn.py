"""ChebConvAttention — drop-in mirror of torch_geometric_temporal/nn/attention/astgcn.py:16-199 (the graph convolution
of ASTGCN).  Same constructor, parameter names (`_weight [K, in, out]`, `_bias [out]`), forward signature, errors and
`__repr__`; the Laplacian normalisation runs on the device (pgt_cheb_prep, variant 1 = the in-tree `__norm__`), the
attention-weighted hop gathers S[b, row, col] at the edges instead of materialising a [B, E] temporary, the diagonal
scaling replaces the reference's dense eye(N) * S batched matmul, and the gradient w.r.t. the attention is a sampled
dense-dense product kernel.
"""
from typing import Optional

import torch

from ... import ops


class ChebConvAttention(torch.nn.Module):
    r"""Chebyshev spectral graph convolution with spatial attention (reference: astgcn.py:16-199).

    Args: in_channels, out_channels, K, normalization (None | "sym" | "rw"), bias."""

    def __init__(self, in_channels: int, out_channels: int, K: int, normalization: Optional[str] = None,
                 bias: bool = True, **kwargs):
        super().__init__()
        assert K > 0
        assert normalization in [None, "sym", "rw"], "Invalid normalization"
        self._in_channels = in_channels
        self._out_channels = out_channels
        self._normalization = normalization
        self._weight = torch.nn.Parameter(torch.empty(K, in_channels, out_channels))
        if bias:
            self._bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("_bias", None)
        self._reset_parameters()

    def _reset_parameters(self):
        torch.nn.init.xavier_uniform_(self._weight)
        if self._bias is not None:
            torch.nn.init.uniform_(self._bias)

    def forward(self, x, edge_index, spatial_attention, edge_weight=None, batch=None, lambda_max=None):
        """x [B, N, F_in], edge_index [2, E], spatial_attention [B, N, N] -> [B, N, F_out] (astgcn.py:112-183)."""
        if self._normalization != "sym" and lambda_max is None:
            raise ValueError(
                "You need to pass `lambda_max` to `forward() in`"
                "case the normalization is non-symmetric."
            )
        if lambda_max is None:
            lam = 2.0
        elif isinstance(lambda_max, torch.Tensor):
            if lambda_max.numel() > 1:
                # per-graph lambda_max with a `batch` vector (astgcn.py:97-98) only matters for PyG-style disjoint
                # batches; on the [B, N, F] layout every batch entry shares the graph, so all entries must agree
                if not bool((lambda_max == lambda_max.flatten()[0]).all()):
                    raise NotImplementedError("ChebConvAttention: per-graph lambda_max values differ")
            lam = float(lambda_max.flatten()[0])
        else:
            lam = float(lambda_max)
        if isinstance(edge_index, (list, tuple)):       # the reference accepts a list for edge_index (attention_test.py)
            edge_index = torch.as_tensor(edge_index, device=x.device)
        g = ops.cheb_graph(edge_index, edge_weight, x.size(1), self._normalization, lam, variant=1)
        return ops.ChebConvAttentionFunction.apply(x, spatial_attention, self._weight, self._bias, g,
                                                   self._weight.size(0))

    def __repr__(self):
        return "{}({}, {}, K={}, normalization={})".format(
            self.__class__.__name__, self._in_channels, self._out_channels, self._weight.size(0), self._normalization)
