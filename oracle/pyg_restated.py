"""TEST INFRASTRUCTURE — CPU restatement of the PyTorch Geometric primitives the hot path calls.

The reference (torch_geometric_temporal) keeps its sparse arithmetic in the un-vendored, un-pinned third-party
dependency `torch_geometric` (setup.py:3-10; CI installs "latest" for torch 2.3.0, .github/workflows/main.yml:29-32,
i.e. the PyG 2.5/2.6 series), which is NOT present under /root/reference and cannot be installed here.  This file
restates, in plain PyTorch on the CPU, the published algorithm of exactly the PyG entry points the path uses
(call sites: nn/recurrent/dcrnn.py:3-4,59,77,86-99; temporalgcn.py:2,38-70; evolvegcno.py:6-10,88-101;
evolvegcnh.py:3,63; nn/attention/stgcn.py:5,115-121; astgcn.py:11-13,93-107,169-175).

PARITY STATUS: the PyG primitives below are *unpinned* (no PyG build is available to check them against; the
reference's own tests assert shapes only, test/recurrent_test.py:274-315).  They are cross-checked by dual
implementations and closed-form identities in tests/test_oracle_pyg.py.  The reference's *in-tree* modules are
pinned for real: oracle/ref_import.py executes the actual files under /root/reference on top of these primitives
and tests/golden/ holds their outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import inspect
import math

import torch
from torch import Tensor


# ----------------------------------------------------------------------------------------------- utils

def maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def scatter_sum(src, index, dim=0, dim_size=None):
    """torch_geometric.utils.scatter(..., reduce='sum'): zeros(dim_size).scatter_add_/index_add_ in edge order."""
    if dim < 0:
        dim = src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = list(src.shape)
    size[dim] = dim_size
    return src.new_zeros(size).index_add_(dim, index, src)


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    """[1, N, N] with N = edge_index.max()+1 (NOT X.size(0)); duplicate edges are summed; edge_attr None -> ones."""
    assert batch is None
    N = maybe_num_nodes(edge_index, max_num_nodes)
    if edge_attr is None:
        edge_attr = torch.ones(edge_index.size(1), device=edge_index.device)
    idx = edge_index[0] * N + edge_index[1]
    adj = scatter_sum(edge_attr, idx, 0, N * N)
    return adj.view(1, N, N)


def dense_to_sparse(adj):
    """2-D: row-major nonzero() order; exact zeros dropped."""
    assert adj.dim() == 2
    index = adj.nonzero().t().contiguous()
    return index, adj[index[0], index[1]]


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    return edge_index, (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    loop_index = torch.arange(0, N, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fv = 1.0 if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fv)
        edge_attr = torch.cat([edge_attr, loop_attr], dim=0)
    return torch.cat([edge_index, loop_index], dim=1), edge_attr


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    """Existing self-loops are removed from the list and re-appended as the last N entries (node order), keeping
    their weight (last one wins); nodes without one get `fill_value`."""
    N = maybe_num_nodes(edge_index, num_nodes)
    mask = edge_index[0] != edge_index[1]
    loop_index = torch.arange(0, N, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fv = 1.0 if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fv)
        inv_mask = ~mask
        loop_attr[edge_index[0][inv_mask]] = edge_attr[inv_mask]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    return edge_index, edge_attr


def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False, add_self_loops=True,
             flow="source_to_target", dtype=None):
    """PyG 2.5/2.6 order: self-loops first, THEN `None -> ones` (so `improved` has no effect without weights)."""
    fill_value = 2.0 if improved else 1.0
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    idx = col if flow == "source_to_target" else row
    deg = scatter_sum(edge_weight, idx, 0, num_nodes)
    deg_inv_sqrt = deg.pow_(-0.5)
    deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
    edge_weight = deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]
    return edge_index, edge_weight


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    assert normalization in (None, "sym", "rw")
    edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_sum(edge_weight, row, 0, num_nodes)
    if normalization is None:
        edge_index, _ = add_self_loops(edge_index, num_nodes=num_nodes)
        edge_weight = torch.cat([-edge_weight, deg], dim=0)
    elif normalization == "sym":
        deg_inv_sqrt = deg.pow_(-0.5)
        deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
        edge_weight = deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]
        edge_index, edge_weight = add_self_loops(edge_index, -edge_weight, fill_value=1.0, num_nodes=num_nodes)
    else:
        deg_inv = 1.0 / deg
        deg_inv.masked_fill_(deg_inv == float("inf"), 0)
        edge_weight = deg_inv[row] * edge_weight
        edge_index, edge_weight = add_self_loops(edge_index, -edge_weight, fill_value=1.0, num_nodes=num_nodes)
    return edge_index, edge_weight


def glorot(value):
    if isinstance(value, Tensor):
        stdv = math.sqrt(6.0 / (value.size(-2) + value.size(-1)))
        value.data.uniform_(-stdv, stdv)


def zeros(value):
    if isinstance(value, Tensor):
        value.data.fill_(0)


def uniform(size, value):
    if isinstance(value, Tensor):
        bound = 1.0 / math.sqrt(size)
        value.data.uniform_(-bound, bound)


# ----------------------------------------------------------------------------------------------- data

class Data:
    """Attribute bag standing in for torch_geometric.data.Data (only what the signal iterators use)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        self._extra = list(kwargs)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in ["x", "edge_index", "edge_attr", "y"] + self._extra if getattr(self, k) is not None]

    def to(self, device):
        for k in self.keys():
            v = getattr(self, k)
            if isinstance(v, Tensor):
                setattr(self, k, v.to(device))
        return self


# ----------------------------------------------------------------------------------------------- message passing

class MessagePassing(torch.nn.Module):
    """aggr="add" only.  propagate(): `*_j` kwargs are gathered at edge_index[0] (source), `*_i` at edge_index[1]
    (target) for flow="source_to_target"; edge-aligned kwargs are passed through positionally; the messages are
    summed into zeros(N) along node_dim with N = x.size(node_dim); update() is the identity."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert aggr == "add", "only aggr='add' is on the path"
        assert flow in ("source_to_target", "target_to_source")
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        assert size is None
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        params = list(inspect.signature(self.message).parameters)
        dim_size = None
        margs = {}
        for name in params:
            if name.endswith("_j") or name.endswith("_i"):
                data = kwargs[name[:-2]]
                idx = edge_index[j] if name.endswith("_j") else edge_index[i]
                dim_size = data.size(self.node_dim)
                margs[name] = data.index_select(self.node_dim, idx)
            else:
                margs[name] = kwargs.get(name)
        out = self.message(**margs)
        dim = self.node_dim if self.node_dim >= 0 else out.dim() + self.node_dim
        return scatter_sum(out, edge_index[i], dim, dim_size)

    def message(self, x_j):
        return x_j


class Linear(torch.nn.Module):
    """torch_geometric.nn.dense.linear.Linear: weight [out, in]; parameter names `weight`, `bias`."""

    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        if weight_initializer == "glorot":
            glorot(self.weight)
        else:
            torch.nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            zeros(self.bias)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class GCNConv(MessagePassing):
    """out = D^-1/2 (A + I) D^-1/2 (X W^T) + b  — gcn_norm -> lin -> propagate -> + bias."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
                 normalize=True, bias=True, **kwargs):
        kwargs.setdefault("aggr", "add")
        super().__init__(**kwargs)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops, self.normalize = improved, cached, add_self_loops, normalize
        self._cached_edge_index = None
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x, edge_index, edge_weight=None):
        if self.normalize:
            cache = self._cached_edge_index
            if cache is None:
                edge_index, edge_weight = gcn_norm(edge_index, edge_weight, x.size(self.node_dim), self.improved,
                                                   self.add_self_loops, self.flow, x.dtype)
                if self.cached:
                    self._cached_edge_index = (edge_index, edge_weight)
            else:
                edge_index, edge_weight = cache
        x = self.lin(x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight, size=None)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_weight):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j


class ChebConv(MessagePassing):
    """PyG ChebConv: scaled Laplacian 2L/lambda_max - I, T_0 = x, T_1 = L^ x, T_k = 2 L^ T_{k-1} - T_{k-2}."""

    def __init__(self, in_channels, out_channels, K, normalization="sym", bias=True, **kwargs):
        kwargs.setdefault("aggr", "add")
        super().__init__(**kwargs)
        assert K > 0
        assert normalization in (None, "sym", "rw")
        self.in_channels, self.out_channels, self.normalization = in_channels, out_channels, normalization
        self.lins = torch.nn.ModuleList(
            [Linear(in_channels, out_channels, bias=False, weight_initializer="glorot") for _ in range(K)])
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def __norm__(self, edge_index, num_nodes, edge_weight, normalization, lambda_max=None, dtype=None, batch=None):
        edge_index, edge_weight = get_laplacian(edge_index, edge_weight, normalization, dtype, num_nodes)
        if lambda_max is None:
            lambda_max = 2.0 * edge_weight.max()
        elif not isinstance(lambda_max, Tensor):
            lambda_max = torch.tensor(lambda_max, dtype=dtype, device=edge_index.device)
        if batch is not None and lambda_max.numel() > 1:
            lambda_max = lambda_max[batch[edge_index[0]]]
        edge_weight = (2.0 * edge_weight) / lambda_max
        edge_weight.masked_fill_(edge_weight == float("inf"), 0)
        loop_mask = edge_index[0] == edge_index[1]
        edge_weight[loop_mask] -= 1
        return edge_index, edge_weight

    def forward(self, x, edge_index, edge_weight=None, batch=None, lambda_max=None):
        edge_index, norm = self.__norm__(edge_index, x.size(self.node_dim), edge_weight, self.normalization,
                                         lambda_max, dtype=x.dtype, batch=batch)
        Tx_0 = x
        Tx_1 = x
        out = self.lins[0](Tx_0)
        if len(self.lins) > 1:
            Tx_1 = self.propagate(edge_index, x=x, norm=norm, size=None)
            out = out + self.lins[1](Tx_1)
        for lin in self.lins[2:]:
            Tx_2 = self.propagate(edge_index, x=Tx_1, norm=norm, size=None)
            Tx_2 = 2.0 * Tx_2 - Tx_0
            out = out + lin(Tx_2)
            Tx_0, Tx_1 = Tx_1, Tx_2
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, norm):
        return norm.view(-1, 1) * x_j


class _SelectTopK(torch.nn.Module):
    def __init__(self, in_channels, ratio):
        super().__init__()
        self.in_channels, self.ratio = in_channels, ratio
        self.weight = torch.nn.Parameter(torch.empty(1, in_channels))
        uniform(in_channels, self.weight)


class TopKPooling(torch.nn.Module):
    """score = tanh((x . p) / ||p||); perm = top-k(score), k = ceil(ratio * N); returns x[perm] * score[perm]
    first (the only output EvolveGCN-H consumes, evolvegcnh.py:95-96).  Parameter name: `select.weight` [1, F]."""

    def __init__(self, in_channels, ratio=0.5, min_score=None, multiplier=1.0, nonlinearity="tanh"):
        super().__init__()
        assert min_score is None and nonlinearity == "tanh"
        self.in_channels, self.ratio, self.multiplier = in_channels, ratio, multiplier
        self.select = _SelectTopK(in_channels, ratio)

    def forward(self, x, edge_index, edge_attr=None, batch=None, attn=None):
        attn = x if attn is None else attn
        w = self.select.weight
        score = (attn * w).sum(dim=-1)
        score = torch.tanh(score / w.norm(p=2, dim=-1))
        n = x.size(0)
        if isinstance(self.ratio, int):
            k = min(self.ratio, n)
        else:
            k = int((float(self.ratio) * torch.tensor(n).to(score.dtype)).ceil().to(torch.long))
        _, order = torch.sort(score.view(-1), descending=True)
        perm = order[:k]
        s = score[perm]
        xo = x[perm] * s.view(-1, 1)
        if self.multiplier != 1:
            xo = self.multiplier * xo
        mask = torch.zeros(n, dtype=torch.bool)
        mask[perm] = True
        emask = mask[edge_index[0]] & mask[edge_index[1]]
        remap = torch.full((n,), -1, dtype=torch.long)
        remap[perm] = torch.arange(k)
        ei = remap[edge_index[:, emask]]
        ea = None if edge_attr is None else edge_attr[emask]
        return xo, ei, ea, batch, perm, s
