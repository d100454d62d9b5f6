"""GConvGRU / GConvLSTM / GCLSTM — drop-in mirrors of torch_geometric_temporal/nn/recurrent/gconv_gru.py,
gconv_lstm.py and gc_lstm.py (the Chebyshev-convolution cells).

Same constructors, parameter names (`conv_x_z.lins.{k}.weight`, `conv_h_z.bias`, `w_c_i`, `b_i`, `W_i`, ...), forward
signatures and outputs.  Every gate of a cell convolves the SAME inputs, so instead of 6 / 8 / 4 independent ChebConv
calls (each with its own Laplacian normalisation and K-1 propagates) a cell runs ONE Chebyshev stack of [X, H] (one
aggregation launch per hop at width in+out) and ONE MFMA GEMM that produces all gate pre-activations; GConvGRU's
candidate needs a second stack of [X, H*R].  The LSTM gate chain (peepholes included) is one fused kernel
(pgt_lstm_gates_f32) with a hand-written backward; GConvGRU's gate chain runs in the epilogues of its two GEMMs
(pgt_gemm_gru_zr/h_f32, the entry points DCRNN uses) inside one autograd node (ops.ChebGRUCellFunction).
"""
from typing import Tuple

import torch

from ... import ops
from ..conv import ChebConv, glorot_


def _gate_weights(x_convs, h_convs):
    """Stack lins[k].weight^T of several ChebConvs acting on X (x_convs, may be empty) and on H (h_convs) into the
    [K*(in+out), G*out] operand of ops.ChebConvFunction, plus the summed biases [G*out] (None if no conv has one)."""
    K = len(h_convs[0].lins)
    rows = []
    for k in range(K):
        hs = torch.cat([c.lins[k].weight.t() for c in h_convs], dim=1)
        if x_convs:
            xs = torch.cat([c.lins[k].weight.t() for c in x_convs], dim=1)
            rows.append(torch.cat([xs, hs], dim=0))
        else:
            rows.append(hs)
    W = torch.cat(rows, dim=0)
    bs = []
    any_bias = False
    for g, hc in enumerate(h_convs):
        b = hc.bias
        if x_convs and x_convs[g].bias is not None:
            b = x_convs[g].bias if b is None else b + x_convs[g].bias
        any_bias = any_bias or b is not None
        bs.append(b)
    if not any_bias:
        return W, None
    out = h_convs[0].out_channels
    bs = [b if b is not None else torch.zeros(out, device=W.device, dtype=W.dtype) for b in bs]
    return W, torch.cat(bs)


def _graph(conv, edge_index, edge_weight, n, lambda_max):
    lam = None if lambda_max is None else float(lambda_max)
    return ops.cheb_graph(edge_index, edge_weight, n, conv.normalization, lam, variant=0)


class GConvGRU(torch.nn.Module):
    r"""Chebyshev graph convolutional GRU cell (reference: gconv_gru.py:5-170)."""

    def __init__(self, in_channels: int, out_channels: int, K: int, normalization: str = "sym", bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.normalization = normalization
        self.bias = bias
        for gate in ("z", "r", "h"):
            setattr(self, f"conv_x_{gate}", ChebConv(in_channels, out_channels, K, normalization, bias))
            setattr(self, f"conv_h_{gate}", ChebConv(out_channels, out_channels, K, normalization, bias))

    def _set_hidden_state(self, X, H):
        if H is None:
            H = torch.zeros(X.shape[0], self.out_channels, device=X.device, dtype=X.dtype)
        return H

    def forward(self, X, edge_index, edge_weight=None, H=None, lambda_max=None):
        H = self._set_hidden_state(X, H)
        O = self.out_channels
        g = _graph(self.conv_x_z, edge_index, edge_weight, X.size(0), lambda_max)
        Wzr, bzr = _gate_weights([self.conv_x_z, self.conv_x_r], [self.conv_h_z, self.conv_h_r])
        Wh, bh = _gate_weights([self.conv_x_h], [self.conv_h_h])
        # both Chebyshev stacks, both gate GEMMs (sigmoid / H*R and tanh / blend in their epilogues) and the
        # hand-written backward in one autograd node
        return ops.ChebGRUCellFunction.apply(X, H, Wzr, bzr, Wh, bh, g, self.K)


class GConvLSTM(torch.nn.Module):
    r"""Chebyshev graph convolutional LSTM cell with peepholes (reference: gconv_lstm.py:9-238)."""

    def __init__(self, in_channels: int, out_channels: int, K: int, normalization: str = "sym", bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.normalization = normalization
        self.bias = bias
        for gate in ("i", "f", "c", "o"):
            setattr(self, f"conv_x_{gate}", ChebConv(in_channels, out_channels, K, normalization, bias))
            setattr(self, f"conv_h_{gate}", ChebConv(out_channels, out_channels, K, normalization, bias))
            if gate != "c":
                setattr(self, f"w_c_{gate}", torch.nn.Parameter(torch.empty(1, out_channels)))
            setattr(self, f"b_{gate}", torch.nn.Parameter(torch.empty(1, out_channels)))
        self._set_parameters()

    def _set_parameters(self):
        for gate in ("i", "f", "o"):
            glorot_(getattr(self, f"w_c_{gate}"))
        for gate in ("i", "f", "c", "o"):
            torch.nn.init.zeros_(getattr(self, f"b_{gate}"))

    def forward(self, X, edge_index, edge_weight=None, H=None, C=None, lambda_max=None) -> Tuple[torch.Tensor, torch.Tensor]:
        O = self.out_channels
        if H is None:
            H = torch.zeros(X.shape[0], O, device=X.device, dtype=X.dtype)
        if C is None:
            C = torch.zeros(X.shape[0], O, device=X.device, dtype=X.dtype)
        g = _graph(self.conv_x_i, edge_index, edge_weight, X.size(0), lambda_max)
        W, b = _gate_weights([self.conv_x_i, self.conv_x_f, self.conv_x_c, self.conv_x_o],
                             [self.conv_h_i, self.conv_h_f, self.conv_h_c, self.conv_h_o])
        # the cell's own biases b_i .. b_o join the convolution biases inside the GEMM
        bb = torch.cat([self.b_i, self.b_f, self.b_c, self.b_o], dim=1).view(-1)
        b = bb if b is None else b + bb
        P = ops.ChebConvFunction.apply(torch.cat([X, H], dim=1), W, b, g, self.K, 1)     # [N, 4*O]: i | f | c | o
        return ops.LSTMGatesFunction.apply(P, C, self.w_c_i, self.w_c_f, self.w_c_o)


class GCLSTM(torch.nn.Module):
    r"""Integrated graph convolutional LSTM cell (reference: gc_lstm.py:9-205): dense input weights W_*, Chebyshev
    convolution of the hidden state only."""

    def __init__(self, in_channels: int, out_channels: int, K: int, normalization: str = "sym", bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.K = K
        self.normalization = normalization
        self.bias = bias
        for gate in ("i", "f", "c", "o"):
            setattr(self, f"conv_{gate}", ChebConv(out_channels, out_channels, K, normalization, bias))
            setattr(self, f"W_{gate}", torch.nn.Parameter(torch.empty(in_channels, out_channels)))
            setattr(self, f"b_{gate}", torch.nn.Parameter(torch.empty(1, out_channels)))
        self._set_parameters()

    def _set_parameters(self):
        for gate in ("i", "f", "c", "o"):
            glorot_(getattr(self, f"W_{gate}"))
            torch.nn.init.zeros_(getattr(self, f"b_{gate}"))

    def forward(self, X, edge_index, edge_weight=None, H=None, C=None, lambda_max=None) -> Tuple[torch.Tensor, torch.Tensor]:
        O = self.out_channels
        if H is None:
            H = torch.zeros(X.shape[0], O, device=X.device, dtype=X.dtype)
        if C is None:
            C = torch.zeros(X.shape[0], O, device=X.device, dtype=X.dtype)
        g = _graph(self.conv_i, edge_index, edge_weight, X.size(0), lambda_max)
        W, b = _gate_weights([], [self.conv_i, self.conv_f, self.conv_c, self.conv_o])
        Wx = torch.cat([self.W_i, self.W_f, self.W_c, self.W_o], dim=1)                  # [in, 4*O]
        bx = torch.cat([self.b_i, self.b_f, self.b_c, self.b_o], dim=1).view(-1)
        P = ops.linear(X, Wx, bx) + ops.ChebConvFunction.apply(H, W, b, g, self.K, 1)
        return ops.LSTMGatesFunction.apply(P, C, None, None, None)
