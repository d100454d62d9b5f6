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
        lam, lam_graphs = ops.cheb_lambda(lambda_max, batch)     # one lambda per graph through `batch` (astgcn.py:97-98)
        if lam is None and lam_graphs is None:
            lam = 2.0
        if isinstance(edge_index, (list, tuple)):       # the reference accepts a list for edge_index (attention_test.py)
            edge_index = torch.as_tensor(edge_index, device=x.device)
        if lam_graphs is not None:
            g = ops.cheb_graph(edge_index, edge_weight, x.size(1), self._normalization, lam_graphs, variant=1, batch=batch)
        else:
            g = ops.cheb_graph(edge_index, edge_weight, x.size(1), self._normalization, lam, variant=1)
        return ops.ChebConvAttentionFunction.apply(x, spatial_attention, self._weight, self._bias, g,
                                                   self._weight.size(0))

    def __repr__(self):
        return "{}({}, {}, K={}, normalization={})".format(
            self.__class__.__name__, self._in_channels, self._out_channels, self._weight.size(0), self._normalization)


def _init_like_reference(module):
    """xavier_uniform for matrices, uniform(0, 1) for vectors (astgcn.py:220-225 and the other _reset_parameters)."""
    for p in module.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_uniform_(p)
        else:
            torch.nn.init.uniform_(p)


class SpatialAttention(torch.nn.Module):
    r"""Spatial attention of ASTGCN (reference: astgcn.py:201-262): X [B, N, F, T] -> S [B, N, N],
    S = softmax_dim1( Vs . sigmoid( (X W1 W2) (W3 X)^T + bs ) ).  The embeddings are strided small products on pgt_bmm_f32
    (ops.bmm: no transposed copies); the [B, N, N] part (product, bias, sigmoid, Vs product, softmax) is
    ops.AttentionScoresFunction (csrc/attention.hip)."""

    def __init__(self, in_channels: int, num_of_vertices: int, num_of_timesteps: int):
        super().__init__()
        self._W1 = torch.nn.Parameter(torch.empty(num_of_timesteps))
        self._W2 = torch.nn.Parameter(torch.empty(in_channels, num_of_timesteps))
        self._W3 = torch.nn.Parameter(torch.empty(in_channels))
        self._bs = torch.nn.Parameter(torch.empty(1, num_of_vertices, num_of_vertices))
        self._Vs = torch.nn.Parameter(torch.empty(num_of_vertices, num_of_vertices))
        _init_like_reference(self)

    def forward(self, X):
        B, N, F_, T = X.shape
        Xc = X.contiguous()
        # (X W1) W2: [B N F, T] x [T, 1], then [B N, F] x [F, T]  (astgcn.py:252)
        a = ops.bmm(Xc.view(1, B * N * F_, T), self._W1.view(1, T, 1)).view(1, B * N, F_)
        lhs = ops.bmm(a, self._W2.unsqueeze(0)).view(B, N, T)                                        # [B, N, T]
        # W3 X: the contraction runs over F inside every [F, T] block — a [1, F] row vector shared by the B N blocks (:256)
        rhs = ops.bmm(self._W3.view(1, F_), Xc.view(B * N, F_, T)).view(B, N, T).transpose(-1, -2)   # [B, T, N]
        # V . sigmoid(lhs rhs + b) and the softmax over dim 1: fused score kernel, one MFMA GEMM for the batch, softmax
        return ops.AttentionScoresFunction.apply(lhs, rhs, self._bs, self._Vs)


class TemporalAttention(torch.nn.Module):
    r"""Temporal attention of ASTGCN (reference: astgcn.py:265-328): X [B, N, F, T] -> E [B, T, T]."""

    def __init__(self, in_channels: int, num_of_vertices: int, num_of_timesteps: int):
        super().__init__()
        self._U1 = torch.nn.Parameter(torch.empty(num_of_vertices))
        self._U2 = torch.nn.Parameter(torch.empty(in_channels, num_of_vertices))
        self._U3 = torch.nn.Parameter(torch.empty(in_channels))
        self._be = torch.nn.Parameter(torch.empty(1, num_of_timesteps, num_of_timesteps))
        self._Ve = torch.nn.Parameter(torch.empty(num_of_timesteps, num_of_timesteps))
        _init_like_reference(self)

    def forward(self, X):
        B, N, F_, T = X.shape
        Xc = X.contiguous()
        # (X^T U1) U2: the contraction over the nodes is a [1, N] row vector times the [N, F T] view of every batch entry,
        # then a[b] viewed [T, F] (strides 1, T) times U2 [F, N]  (astgcn.py:318)
        a = ops.bmm(self._U1.view(1, N), Xc.view(B, N, F_ * T)).view(B, F_, T)
        lhs = ops.bmm(a.transpose(1, 2), self._U2)                                                    # [B, T, N]
        rhs = ops.bmm(self._U3.view(1, F_), Xc.view(B * N, F_, T)).view(B, N, T)                      # [B, N, T]  (:322)
        return ops.AttentionScoresFunction.apply(lhs, rhs, self._be, self._Ve)


class ASTGCNBlock(torch.nn.Module):
    r"""One ASTGCN block (reference: astgcn.py:330-481): temporal attention -> spatial attention -> Chebyshev
    convolution with that attention on every time step -> time convolution + residual -> LayerNorm.
    The reference calls `ChebConvAttention` once per time step in a Python loop with the SAME attention (:442-452);
    here the T steps are one call (time folded next to the channels: one aggregation launch per hop, one GEMM)."""

    def __init__(self, in_channels: int, K: int, nb_chev_filter: int, nb_time_filter: int, time_strides: int,
                 num_of_vertices: int, num_of_timesteps: int, normalization: Optional[str] = None, bias: bool = True):
        super().__init__()
        self._temporal_attention = TemporalAttention(in_channels, num_of_vertices, num_of_timesteps)
        self._spatial_attention = SpatialAttention(in_channels, num_of_vertices, num_of_timesteps)
        self._chebconv_attention = ChebConvAttention(in_channels, nb_chev_filter, K, normalization, bias)
        self._time_convolution = torch.nn.Conv2d(nb_chev_filter, nb_time_filter, kernel_size=(1, 3),
                                                 stride=(1, time_strides), padding=(0, 1))
        self._residual_convolution = torch.nn.Conv2d(in_channels, nb_time_filter, kernel_size=(1, 1),
                                                     stride=(1, time_strides))
        self._layer_norm = torch.nn.LayerNorm(nb_time_filter)
        self._normalization = normalization
        _init_like_reference(self)

    def _lambda_max(self, edge_index, n):
        if self._normalization == "sym":
            return None
        from ..conv import laplacian_lambda_max
        # LaplacianLambdaMax() with its default normalization=None (astgcn.py:438, :460): the largest eigenvalue of the
        # UNNORMALISED Laplacian, whatever the block's normalization is
        return laplacian_lambda_max(edge_index, n, None)

    def forward(self, X, edge_index):
        B, N, Fin, T = X.shape
        E = self._temporal_attention(X)                                               # [B, T, T]
        X_tilde = ops.bmm(X.contiguous().view(B, N * Fin, T), E).view(B, N, Fin, T)   # X E per batch entry (astgcn.py:437)
        S = self._spatial_attention(X_tilde)                                          # [B, N, N]
        conv = self._chebconv_attention
        Xcl = X.permute(0, 1, 3, 2).contiguous()                                      # [B, N, T, Fin] channels last
        if not isinstance(edge_index, list):
            lam = self._lambda_max(edge_index, N)
            if conv._normalization != "sym" and lam is None:
                raise ValueError("You need to pass `lambda_max` to `forward() in`case the normalization is non-symmetric.")
            g = ops.cheb_graph(edge_index, None, N, conv._normalization, 2.0 if lam is None else lam, variant=1)
            out = ops.ChebConvAttentionFunction.apply(Xcl, S, conv._weight, conv._bias, g, conv._weight.size(0))   # [B, N, T, O]
        else:                                                                         # one graph per time step
            steps = [conv(X[:, :, :, t], edge_index[t], S, lambda_max=self._lambda_max(edge_index[t], N)).unsqueeze(2)
                     for t in range(T)]
            out = torch.cat(steps, dim=2)                                             # [B, N, T, O]
        # time convolution + residual convolution + relu + LayerNorm on channels-last rows: one segmented GEMM for the three
        # taps, the residual accumulated into it, relu + LayerNorm in one pass (ops.TimeConvResidualNormFunction)
        tc, rc, ln = self._time_convolution, self._residual_convolution, self._layer_norm
        y = ops.TimeConvResidualNormFunction.apply(torch.relu(out), Xcl, tc.weight, tc.bias, rc.weight, rc.bias, ln.weight,
                                                   ln.bias, tc.stride[1], ln.eps)    # [B, N, T_out, Ft]
        return y.permute(0, 1, 3, 2)                                                  # [B, N, Ft, T_out]


class ASTGCN(torch.nn.Module):
    r"""Attention-based spatial-temporal GCN (reference: astgcn.py:484-640).
    X [B, N, F_in, T_in], edge_index -> [B, N, T_out]."""

    def __init__(self, nb_block: int, in_channels: int, K: int, nb_chev_filter: int, nb_time_filter: int,
                 time_strides: int, num_for_predict: int, len_input: int, num_of_vertices: int,
                 normalization: Optional[str] = None, bias: bool = True):
        super().__init__()
        blocks = [ASTGCNBlock(in_channels, K, nb_chev_filter, nb_time_filter, time_strides, num_of_vertices, len_input,
                              normalization, bias)]
        blocks += [ASTGCNBlock(nb_time_filter, K, nb_chev_filter, nb_time_filter, 1, num_of_vertices,
                               len_input // time_strides, normalization, bias) for _ in range(nb_block - 1)]
        self._blocklist = torch.nn.ModuleList(blocks)
        self._final_conv = torch.nn.Conv2d(int(len_input / time_strides), num_for_predict, kernel_size=(1, nb_time_filter))
        _init_like_reference(self)

    def forward(self, X, edge_index):
        for block in self._blocklist:
            X = block(X, edge_index)
        return self._final_conv(X.permute(0, 3, 1, 2))[:, :, :, -1].permute(0, 2, 1)
