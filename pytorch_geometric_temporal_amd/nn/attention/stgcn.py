"""TemporalConv / STConv — drop-in mirrors of torch_geometric_temporal/nn/attention/stgcn.py.

The graph convolution of the ST-Conv block is the message-passing part: the reference calls ChebConv once per
(batch, time) slice in a Python double loop (stgcn.py:151-153), recomputing the Laplacian normalisation each time.
Here all B*T' slices are folded into the feature dimension of ONE Chebyshev stack (K-1 aggregation launches + one MFMA
GEMM in total).  The gated temporal convolutions (three Conv2d + the gate: ONE launch on the matrix cores, csrc/tconv.hip)
and the node-wise batch norm run on the reference's own [B, T, N, C] layout — the reference's four permutes per block do
not exist here; `conv_1/2/3` and `_batch_norm` remain the torch modules that HOLD the parameters and buffers, so reference
checkpoints load with strict=True.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..conv import ChebConv


class TemporalConv(nn.Module):
    r"""Gated temporal convolution (reference: stgcn.py:8-44).  X [B, T, N, in] -> [B, T-(k-1), N, out]."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3):
        super().__init__()
        self.conv_1 = nn.Conv2d(in_channels, out_channels, (1, kernel_size))
        self.conv_2 = nn.Conv2d(in_channels, out_channels, (1, kernel_size))
        self.conv_3 = nn.Conv2d(in_channels, out_channels, (1, kernel_size))

    def forward(self, X):
        if X.dtype != torch.float32 or self.conv_1.weight.dtype != torch.float32:
            # a double / half module (`.double()`, `.half()`): the kernels are fp32 — torch's own convolutions on the SAME device,
            # in the order of stgcn.py:36-44
            Xp = X.permute(0, 3, 2, 1)
            H = F.relu(self.conv_1(Xp) * torch.sigmoid(self.conv_2(Xp)) + self.conv_3(Xp))
            return H.permute(0, 3, 2, 1)
        return ops.temporal_conv(X, self.conv_1, self.conv_2, self.conv_3)     # relu(P * sigmoid(Q) + R), stgcn.py:36-44


class STConv(nn.Module):
    r"""Spatio-temporal convolution block (reference: stgcn.py:47-160).
    X [B, T, N, in] -> [B, T - 2(kernel_size-1), N, out]."""

    def __init__(self, num_nodes: int, in_channels: int, hidden_channels: int, out_channels: int, kernel_size: int,
                 K: int, normalization: str = "sym", bias: bool = True):
        super().__init__()
        self.num_nodes = num_nodes
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.K = K
        self.normalization = normalization
        self.bias = bias
        self._temporal_conv1 = TemporalConv(in_channels, hidden_channels, kernel_size)
        self._graph_conv = ChebConv(hidden_channels, hidden_channels, K=K, normalization=normalization, bias=bias)
        self._temporal_conv2 = TemporalConv(hidden_channels, out_channels, kernel_size)
        self._batch_norm = nn.BatchNorm2d(num_nodes)

    def forward(self, X, edge_index, edge_weight=None):
        T_0 = self._temporal_conv1(X)                                  # [B, T', N, hidden]
        T = self._graph_conv(T_0, edge_index, edge_weight)             # every (b, t) slice in one Chebyshev stack
        T = F.relu(T)
        T = self._temporal_conv2(T)
        bn = self._batch_norm
        if T.dtype != torch.float32 or (bn.weight is not None and bn.weight.dtype != torch.float32):
            return bn(T.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)         # non-fp32 module: torch's batch norm as stgcn.py:156-159 calls it
        # BatchNorm2d over the node axis, stgcn.py:156-159; batch statistics follow the BATCH NORM's own mode (a frozen
        # `model._batch_norm.eval()` inside a training block uses its running statistics, as in the reference)
        return ops.batch_norm_nodes(T, bn, bn.training)
