"""MSTGCNBlock / MSTGCN — drop-in mirrors of torch_geometric_temporal/nn/attention/mstgcn.py:
ASTGCN without the attention.  The Chebyshev convolution (normalization=None, lambda_max from the Laplacian's largest
eigenvalue) runs on the HIP kernels with every (batch, time) slice folded into one stack; time convolution, residual
and LayerNorm are dense torch modules as in the reference.
"""
import torch

from ..conv import ChebConv, laplacian_lambda_max
from .astgcn import _init_like_reference


class MSTGCNBlock(torch.nn.Module):
    r"""Reference: mstgcn.py:9-122.  X [B, N, F_in, T] -> [B, N, nb_time_filter, T / time_strides]."""

    def __init__(self, in_channels: int, K: int, nb_chev_filter: int, nb_time_filter: int, time_strides: int):
        super().__init__()
        self._cheb_conv = ChebConv(in_channels, nb_chev_filter, K, normalization=None)
        self._time_conv = torch.nn.Conv2d(nb_chev_filter, nb_time_filter, kernel_size=(1, 3), stride=(1, time_strides),
                                          padding=(0, 1))
        self._residual_conv = torch.nn.Conv2d(in_channels, nb_time_filter, kernel_size=(1, 1), stride=(1, time_strides))
        self._layer_norm = torch.nn.LayerNorm(nb_time_filter)
        self.nb_time_filter = nb_time_filter
        _init_like_reference(self)

    def forward(self, X, edge_index):
        B, N, Fin, T = X.shape
        if not isinstance(edge_index, list):
            lam = laplacian_lambda_max(edge_index, N, None)
            # The reference folds (batch, time) into ChebConv's leading dimension through a permute / reshape pair that
            # REINTERPRETS memory rather than transposing it (mstgcn.py:77-90); the same view arithmetic is reproduced
            # so that results match element for element.
            Xt = X.permute(2, 0, 1, 3).reshape(N, Fin, T * B).permute(2, 0, 1)
            Xt = torch.relu(self._cheb_conv(Xt, edge_index, lambda_max=lam))
            Xt = Xt.permute(1, 2, 0).reshape(N, self.nb_time_filter, B, T).permute(2, 0, 1, 3)
        else:
            steps = [self._cheb_conv(X[:, :, :, t], edge_index[t],
                                     lambda_max=laplacian_lambda_max(edge_index[t], N, None)).unsqueeze(-1)
                     for t in range(T)]
            Xt = torch.relu(torch.cat(steps, dim=-1))
        Xt = self._time_conv(Xt.permute(0, 2, 1, 3))
        Xr = self._residual_conv(X.permute(0, 2, 1, 3))
        out = self._layer_norm(torch.relu(Xr + Xt).permute(0, 3, 2, 1))
        return out.permute(0, 2, 3, 1)


class MSTGCN(torch.nn.Module):
    r"""Reference: mstgcn.py:125-198.  X [B, N, F_in, T_in] -> [B, N, T_out]."""

    def __init__(self, nb_block: int, in_channels: int, K: int, nb_chev_filter: int, nb_time_filter: int,
                 time_strides: int, num_for_predict: int, len_input: int):
        super().__init__()
        blocks = [MSTGCNBlock(in_channels, K, nb_chev_filter, nb_time_filter, time_strides)]
        blocks += [MSTGCNBlock(nb_time_filter, K, nb_chev_filter, nb_time_filter, 1) for _ in range(nb_block - 1)]
        self._blocklist = torch.nn.ModuleList(blocks)
        self._final_conv = torch.nn.Conv2d(int(len_input / time_strides), num_for_predict, kernel_size=(1, nb_time_filter))
        _init_like_reference(self)

    def forward(self, X, edge_index):
        for block in self._blocklist:
            X = block(X, edge_index)
        return self._final_conv(X.permute(0, 3, 1, 2))[:, :, :, -1].permute(0, 2, 1)
