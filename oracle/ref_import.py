"""TEST INFRASTRUCTURE — run the reference's OWN module files (read in place from /root/reference, never copied)
on top of the restated PyG primitives of oracle/pyg_restated.py.

`import torch_geometric_temporal` fails in this container because `torch_geometric` is absent.  This helper
registers a minimal stand-in `torch_geometric` package in sys.modules (only the names the hot-path files import),
then loads individual reference files such as nn/recurrent/dcrnn.py by path, bypassing the package __init__
(which would pull in every model).  Used to (1) generate tests/golden/*.npz (oracle/make_golden.py) and
(2) check that oracle/functional.py reproduces the in-tree code bit-for-bit.

/root/reference does not exist on the GPU box: nothing that runs there may import this file's loaders;
`reference_available()` tells the tests whether to skip.
"""
import importlib
import os
import sys
import types

from . import pyg_restated as P

REFERENCE_ROOT = os.environ.get("PGT_REFERENCE_ROOT", "/root/reference")
_PKG = "torch_geometric_temporal"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, _PKG, "nn", "recurrent"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _LaplacianLambdaMax:
    """torch_geometric.transforms.LaplacianLambdaMax: largest eigenvalue of the graph Laplacian (scipy eigs)."""

    def __init__(self, normalization=None, is_undirected=False):
        self.normalization, self.is_undirected = normalization, is_undirected

    def __call__(self, data):
        import numpy as np
        import scipy.sparse as sp
        from scipy.sparse.linalg import eigs, eigsh
        ew = data.edge_attr
        if ew is not None and ew.numel() != data.edge_index.size(1):
            ew = None
        n = data.num_nodes
        ei, w = P.get_laplacian(data.edge_index, ew, self.normalization, num_nodes=n)
        L = sp.coo_matrix((w.numpy(), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))
        fn = eigsh if (self.is_undirected and self.normalization != "rw") else eigs
        lam = fn(L, k=1, which="LM", return_eigenvectors=False)
        data.lambda_max = float(lam.real[0])
        return data


def install_pyg_stub():
    """Idempotent.  Refuses to shadow a real torch_geometric installation."""
    if "torch_geometric" in sys.modules:
        if getattr(sys.modules["torch_geometric"], "__pgt_stub__", False):
            return
        raise RuntimeError("a real torch_geometric is importable; the stub is only for containers without it")
    import torch
    from typing import Optional
    tg = _mod("torch_geometric", __pgt_stub__=True, __path__=[])
    tg.utils = _mod("torch_geometric.utils", to_dense_adj=P.to_dense_adj, dense_to_sparse=P.dense_to_sparse,
                    remove_self_loops=P.remove_self_loops, add_self_loops=P.add_self_loops,
                    get_laplacian=P.get_laplacian, add_remaining_self_loops=P.add_remaining_self_loops)
    tg.nn = _mod("torch_geometric.nn", GCNConv=P.GCNConv, ChebConv=P.ChebConv, TopKPooling=P.TopKPooling,
                 MessagePassing=P.MessagePassing, __path__=[])
    tg.nn.conv = _mod("torch_geometric.nn.conv", MessagePassing=P.MessagePassing, GCNConv=P.GCNConv,
                      ChebConv=P.ChebConv, __path__=[])
    tg.nn.conv.gcn_conv = _mod("torch_geometric.nn.conv.gcn_conv", gcn_norm=P.gcn_norm, GCNConv=P.GCNConv)
    tg.nn.inits = _mod("torch_geometric.nn.inits", glorot=P.glorot, zeros=P.zeros, uniform=P.uniform)
    tg.data = _mod("torch_geometric.data", Data=P.Data)
    tg.typing = _mod("torch_geometric.typing", OptTensor=Optional[torch.Tensor], Adj=torch.Tensor,
                     SparseTensor=type("SparseTensor", (), {}), PairTensor=tuple, OptPairTensor=tuple)
    tg.transforms = _mod("torch_geometric.transforms", LaplacianLambdaMax=_LaplacianLambdaMax)


def _ensure_pkg(name, path):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m
    return sys.modules[name]


def load(dotted):
    """load("nn.recurrent.dcrnn") -> the reference module object, executed from /root/reference in place."""
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not available (expected on the GPU box)")
    install_pyg_stub()
    parts = dotted.split(".")
    base = os.path.join(REFERENCE_ROOT, _PKG)
    _ensure_pkg(_PKG, base)
    name = _PKG
    for p in parts[:-1]:
        base = os.path.join(base, p)
        name = name + "." + p
        _ensure_pkg(name, base)
    return importlib.import_module(_PKG + "." + dotted)


def load_dataset(name):
    """load_dataset("chickenpox") -> the reference's dataset/<name>.py.  The dataset modules import the signal classes
    from the PACKAGE `..signal`, whose real __init__ pulls in the batch / heterogeneous iterators (torch_geometric
    Batch, HeteroData) and dask; the two plain iterators are loaded by path and published on a stand-in package."""
    static = load("signal.static_graph_temporal_signal")
    dynamic = load("signal.dynamic_graph_temporal_signal")
    pkg = sys.modules[_PKG + ".signal"]
    pkg.StaticGraphTemporalSignal = static.StaticGraphTemporalSignal
    pkg.DynamicGraphTemporalSignal = dynamic.DynamicGraphTemporalSignal
    return load("dataset." + name)
