"""The PyG operators the reference's hot path calls (torch_geometric.nn.GCNConv / ChebConv / TopKPooling), as
parameter-compatible modules whose message passing runs on the HIP kernels.  PyG itself is NOT a dependency: these
mirror the call sites of the reference (temporalgcn.py:38-70, stgcn.py:115-121, evolvegcnh.py:63, mpnn_lstm.py) —
same constructor arguments, same state_dict keys (`lin.weight [out,in]`, `bias`, `lins.{k}.weight`), same outputs.
"""
import math

import torch

from ... import ops


def glorot_(t):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


def _rows_in_memory_order(x):
    """(x as [rows, C] without a copy when possible, function restoring the leading shape of a [rows, C'] result).

    A product over the last dimension does not care in which order the rows are visited: when `x` is a permuted view
    of a contiguous tensor (e.g. a `[B, T, N, C]` view of states stored `[T, B, N, C]`), the rows are taken in MEMORY order and the result is handed back as the same permuted view."""
    lead = x.shape[:-1]
    if x.is_contiguous() or x.dim() < 3 or x.stride(-1) != 1:
        return x.reshape(-1, x.shape[-1]), (lambda y: y.view(*lead, y.shape[-1]))
    order = sorted(range(x.dim() - 1), key=lambda d: -x.stride(d))
    xp = x.permute(*order, x.dim() - 1)
    if not xp.is_contiguous():
        return x.reshape(-1, x.shape[-1]), (lambda y: y.view(*lead, y.shape[-1]))
    inv = [order.index(d) for d in range(x.dim() - 1)]
    plead = xp.shape[:-1]
    return xp.reshape(-1, x.shape[-1]), (lambda y: y.view(*plead, y.shape[-1]).permute(*inv, x.dim() - 1))


class Linear(torch.nn.Linear):
    """torch.nn.Linear (same parameters / initialisation / state_dict) whose product runs on the library's exact-fp32
    MFMA GEMM — the per-node readout that follows the recurrent cell in the reference's examples
    (examples/recurrent/dcrnn_example.py: `self.linear = torch.nn.Linear(filters, 1)`).  For out_features of 1-2 a
    library GEMM picks a pathological tile (4.5 ms per call at 2.5 M rows in the profile of round 1a); this one streams
    the rows once."""

    def forward(self, x):
        if getattr(x, "_pgt_pre", None) is not None:
            # relu(states) of a recurrent layer (nn/_states.py): relu and this product as one pass over the states
            return torch.nn.functional.linear(x, self.weight, self.bias)
        x2, restore = _rows_in_memory_order(x)
        return restore(ops.linear(x2, self.weight.t(), self.bias))


class _Lin(torch.nn.Module):
    """PyG's `Linear(in, out, bias=False, weight_initializer="glorot")`: a bare weight [out, in]."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels))
        glorot_(self.weight)

    def forward(self, x):
        x2, restore = _rows_in_memory_order(x)
        return restore(ops.linear(x2, self.weight.t(), None))


def _fold_batch(x):
    """[..., N, C] with any leading batch dims -> node-major rows [N, B*C] (+ the info to undo it)."""
    if x.dim() == 2:
        return x.contiguous(), None
    lead = x.shape[:-2]
    N, C = x.shape[-2], x.shape[-1]
    B = int(torch.Size(lead).numel())
    xb = x.reshape(B, N, C)
    return ops.Swap01.apply(xb, B, N, C).view(N, B * C), (lead, B, N)


def _unfold_batch(y2d, info, C_out):
    if info is None:
        return y2d
    lead, B, N = info
    return ops.Swap01.apply(y2d.view(N, B, C_out), N, B, C_out).view(*lead, N, C_out)


class GCNConv(torch.nn.Module):
    r"""torch_geometric.nn.GCNConv as the reference uses it: out = A_hat (x W^T) + b with
    A_hat = D^-1/2 (A + fill*I) D^-1/2 (gcn_norm; `improved` -> fill 2).  x is [N, in] or [B, N, in] (node_dim = -2:
    one edge list for every batch entry).  `cached=True` freezes the first graph's normalisation, as PyG does."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
                 normalize=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops, self.normalize = improved, cached, add_self_loops, normalize
        if not normalize:
            raise NotImplementedError("GCNConv(normalize=False) is not on the reference's hot path")
        self.lin = _Lin(in_channels, out_channels)
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self._cached_graph = None

    def reset_parameters(self):
        glorot_(self.lin.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)
        self._cached_graph = None

    def graph(self, edge_index, edge_weight, num_nodes):
        if self.cached and self._cached_graph is not None:
            return self._cached_graph
        g = ops.gcn_graph(edge_index, edge_weight, num_nodes, self.improved, self.add_self_loops)
        if self.cached:
            self._cached_graph = g
        return g

    def forward(self, x, edge_index, edge_weight=None):
        g = self.graph(edge_index, edge_weight, x.size(-2))
        # aggregate at the narrower width: A_hat (x W) == (A_hat x) W
        if self.in_channels <= self.out_channels:
            x2, info = _fold_batch(x)
            ax = _unfold_batch(ops.propagate(g, x2), info, self.in_channels)
            out = self.lin(ax)
        else:
            h2, info = _fold_batch(self.lin(x))
            out = _unfold_batch(ops.propagate(g, h2), info, self.out_channels)
        if self.bias is not None:
            out = out + self.bias
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"


class ChebConv(torch.nn.Module):
    r"""torch_geometric.nn.ChebConv (stgcn.py:115-121, gconv_gru.py:57-107): K-order Chebyshev filter on the scaled
    Laplacian 2L/lambda_max - I.  Parameters: `lins.{k}.weight [out, in]`, `bias [out]`."""

    def __init__(self, in_channels, out_channels, K, normalization="sym", bias=True):
        super().__init__()
        assert K > 0
        assert normalization in [None, "sym", "rw"], "Invalid normalization"
        self.in_channels, self.out_channels, self.normalization = in_channels, out_channels, normalization
        self.lins = torch.nn.ModuleList([_Lin(in_channels, out_channels) for _ in range(K)])
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def reset_parameters(self):
        for lin in self.lins:
            glorot_(lin.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight=None, batch=None, lambda_max=None):
        # lambda_max=None -> 2 * max(L), computed on the device (PyG ChebConv.__norm__); a tensor of several values = one per
        # graph of a disjoint batch, selected through `batch`
        lam, lam_graphs = ops.cheb_lambda(lambda_max, batch)
        N = x.size(-2)
        if lam_graphs is not None:
            g = ops.cheb_graph(edge_index, edge_weight, N, self.normalization, lam_graphs, variant=0, batch=batch)
        else:
            g = ops.cheb_graph(edge_index, edge_weight, N, self.normalization, lam, variant=0)
        K = len(self.lins)
        Wst = torch.cat([lin.weight.t() for lin in self.lins], dim=0)          # [K*in, out]
        if x.dim() == 2:
            return ops.ChebConvFunction.apply(x, Wst, self.bias, g, K, 1)
        lead = x.shape[:-2]
        B = int(torch.Size(lead).numel())
        C = x.size(-1)
        xnm = ops.Swap01.apply(x.reshape(B, N, C), B, N, C).view(N * B, C)     # node-major rows m = n*B + b
        out = ops.ChebConvFunction.apply(xnm, Wst, self.bias, g, K, B)
        return ops.Swap01.apply(out.view(N, B, self.out_channels), N, B, self.out_channels).view(
            *lead, N, self.out_channels)

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, K={len(self.lins)}, "
                f"normalization={self.normalization})")


class TopKPooling(torch.nn.Module):
    r"""torch_geometric.nn.TopKPooling as EvolveGCN-H uses it (evolvegcnh.py:61-63, 93-94): only output [0], the
    k = ceil(ratio * N) highest-scoring rows scaled by tanh(score); score = (x . p) / ||p||.  The projection is one
    [N, F] x [F] product on a 129 x 8 matrix followed by a top-k — torch ops, not a kernel of its own.
    Parameter name follows PyG >= 2.4: `select.weight [1, F]`."""

    class _Select(torch.nn.Module):
        def __init__(self, in_channels):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.empty(1, in_channels))
            bound = 1.0 / math.sqrt(in_channels)
            with torch.no_grad():
                self.weight.uniform_(-bound, bound)

    def __init__(self, in_channels, ratio=0.5):
        super().__init__()
        self.in_channels, self.ratio = in_channels, ratio
        self.select = TopKPooling._Select(in_channels)

    def forward(self, x, edge_index=None):
        w = self.select.weight
        score = torch.tanh((x * w).sum(dim=-1) / w.norm(p=2, dim=-1))
        n = x.size(0)
        if isinstance(self.ratio, int):
            k = min(self.ratio, n)
        else:   # PyG computes ceil(ratio * N) in the score dtype (fp32)
            k = int((float(self.ratio) * torch.tensor(n).to(score.dtype)).ceil().to(torch.long))
        perm = torch.sort(score.view(-1), descending=True).indices[:k]
        s = score[perm]
        return (x[perm] * s.view(-1, 1), None, None, None, perm, s)


_LAMBDA_CACHE = {}


def laplacian_lambda_max(edge_index, num_nodes, normalization=None, edge_weight=None, is_undirected=False):
    """Largest eigenvalue of the graph Laplacian — what torch_geometric.transforms.LaplacianLambdaMax computes for
    ASTGCN / MSTGCN every forward (astgcn.py:437-440, mstgcn.py:74-76).  Host-side (scipy ARPACK, as in PyG), once
    per edge list: the result is cached by tensor identity."""
    key = (edge_index.data_ptr(), ops.tensor_version(edge_index), tuple(edge_index.shape), str(edge_index.device), normalization,
           None if edge_weight is None else (edge_weight.data_ptr(), ops.tensor_version(edge_weight)), int(num_nodes))
    hit = _LAMBDA_CACHE.get(key)
    if hit is not None:
        return hit[0]
    import numpy as np
    import scipy.sparse as sp
    from scipy.sparse.linalg import eigs, eigsh
    ei = edge_index.detach().cpu()
    w = torch.ones(ei.size(1)) if edge_weight is None else edge_weight.detach().cpu().float()
    keep = ei[0] != ei[1]
    ei, w = ei[:, keep], w[keep]
    n = int(num_nodes)
    deg = torch.zeros(n).scatter_add_(0, ei[0], w)
    loops = torch.arange(n)
    if normalization is None:
        vals = torch.cat([-w, deg])
    elif normalization == "sym":
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        vals = torch.cat([-dis[ei[0]] * w * dis[ei[1]], torch.ones(n)])
    else:
        dinv = 1.0 / deg
        dinv[dinv == float("inf")] = 0
        vals = torch.cat([-dinv[ei[0]] * w, torch.ones(n)])
    rows, cols = torch.cat([ei[0], loops]).numpy(), torch.cat([ei[1], loops]).numpy()
    L = sp.coo_matrix((vals.numpy().astype(np.float64), (rows, cols)), shape=(n, n))
    fn = eigsh if (is_undirected and normalization != "rw") else eigs
    lam = float(fn(L, k=1, which="LM", return_eigenvectors=False).real[0])
    _LAMBDA_CACHE[key] = (lam, edge_index, edge_weight)       # keep the key tensors alive
    if len(_LAMBDA_CACHE) > 64:
        _LAMBDA_CACHE.pop(next(iter(_LAMBDA_CACHE)))
    return lam
