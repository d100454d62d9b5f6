"""TEST INFRASTRUCTURE — generate tests/golden/*.npz by running the reference's ACTUAL module files
(/root/reference/torch_geometric_temporal/nn/..., loaded in place by oracle/ref_import.py on top of the restated
PyG primitives) on seeded inputs.  Run in the build container only (the GPU box has no /root/reference):

    python -m oracle.make_golden            # rewrites every fixture
    python -m oracle.make_golden dcrnn      # only cases whose name contains "dcrnn"

Each .npz holds:  in/<name> inputs, param/<state_dict key> weights, out/<name> reference outputs, meta/<name>.
The reference's own tests pin shapes only (test/recurrent_test.py:274-315, test/attention_test.py:140-307); the
call forms exercised there (no weight / weight / weight + hidden state; K = 2, 3; all normalisations) are the ones
frozen here with values.
"""
import json
import os
import sys

import numpy as np
import torch

from pytorch_geometric_temporal_amd.dataset import synthetic as syn

from . import ref_import as R

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _t(a, dtype=None):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def _randomise(module, seed):
    """Re-draw every parameter (including the zero-initialised biases) so that no term is trivially zero."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                bound = (6.0 / (p.size(-2) + p.size(-1))) ** 0.5
            else:
                bound = 0.5
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)


def _pack(inputs, module, outputs, meta=None):
    d = {}
    for k, v in inputs.items():
        if v is not None:
            d["in/" + k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    if module is not None:
        for k, v in module.state_dict().items():
            d["param/" + k] = v.detach().numpy()
    for k, v in outputs.items():
        d["out/" + k] = v.detach().numpy()
    for k, v in (meta or {}).items():
        d["meta/" + k] = np.asarray(v)
    return d


def chickenpox_graph():
    """Vendored dataset file of the reference (dataset/chickenpox.json): 20 nodes, 102 edges incl. 20 self-loops."""
    with open(os.path.join(R.REFERENCE_ROOT, "dataset", "chickenpox.json")) as f:
        d = json.load(f)
    ei = np.array(d["edges"], dtype=np.int64).T
    fx = np.array(d["FX"], dtype=np.float32)
    return ei, np.ones(ei.shape[1], dtype=np.float32), fx


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


# ------------------------------------------------------------------------------------------------ DCRNN family

def _dcrnn_case(ei, ew, n, fin, out, K, seed):
    m = R.load("nn.recurrent.dcrnn")
    layer = m.DCRNN(fin, out, K)
    _randomise(layer, seed)
    ei_t, ew_t = _t(ei), (None if ew is None else _t(ew))
    X = _rand((n, fin), seed + 1)
    H0 = _rand((n, out), seed + 2)
    with torch.no_grad():
        h_nw = layer(X, ei_t)                 # no weight  (test/recurrent_test.py:289)
        h_w = layer(X, ei_t, ew_t) if ew is not None else h_nw
        h_wh = layer(X, ei_t, ew_t, H0) if ew is not None else layer(X, ei_t, None, H0)
    return _pack({"X": X, "H0": H0, "edge_index": ei_t, "edge_weight": ew_t}, layer,
                 {"H_noweight": h_nw, "H_weight": h_w, "H_weight_hidden": h_wh}, {"K": K})


@case
def dcrnn_chickenpox_K1():
    ei, ew, _ = chickenpox_graph()
    return _dcrnn_case(ei, ew * 0 + _rand((ei.shape[1],), 5, 0.5, 1.5).numpy(), 20, 4, 32, 1, 10)


@case
def dcrnn_chickenpox_K2():
    ei, ew, _ = chickenpox_graph()
    return _dcrnn_case(ei, _rand((ei.shape[1],), 6, 0.5, 1.5).numpy(), 20, 4, 32, 2, 11)


@case
def dcrnn_chickenpox_K3():
    ei, ew, _ = chickenpox_graph()
    return _dcrnn_case(ei, _rand((ei.shape[1],), 7, 0.5, 1.5).numpy(), 20, 4, 32, 3, 12)


@case
def dcrnn_ws_directed_K3():
    # the reference tests' own mock graph: zero in-degree sources -> inf / nan in the output (kept in the fixture)
    ei = syn.watts_strogatz_directed(40, 6, 0.5, seed=3)
    ew = _rand((ei.shape[1],), 8, 0.05, 1.0).numpy()
    return _dcrnn_case(ei, ew, 40, 8, 16, 3, 13)


@case
def dcrnn_sensor_asym_K3():
    ei, ew = syn.sensor_graph(60, 420, seed=4, symmetric=False)
    return _dcrnn_case(ei, ew, 60, 2, 16, 3, 14)


@case
def dcrnn_sensor_sym_K2():
    ei, ew = syn.sensor_graph(60, 420, seed=5, symmetric=True)
    return _dcrnn_case(ei, ew, 60, 2, 16, 2, 15)


@case
def dconv_sensor_asym_K3():
    m = R.load("nn.recurrent.dcrnn")
    ei, ew = syn.sensor_graph(50, 330, seed=6, symmetric=False)
    layer = m.DConv(6, 10, 3)
    _randomise(layer, 16)
    X = _rand((50, 6), 17)
    with torch.no_grad():
        H = layer(X, _t(ei), _t(ew))
        Hb = m.BatchedDConv(6, 10, 3)
        Hb.load_state_dict(layer.state_dict())
        Hb_out = Hb(X, _t(ei), _t(ew))
    return _pack({"X": X, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, {"H": H, "H_batched": Hb_out},
                 {"K": 3})


@case
def batched_dcrnn_sensor_K3():
    m = R.load("nn.recurrent.dcrnn")
    ei, ew = syn.sensor_graph(40, 270, seed=7, symmetric=False)
    layer = m.BatchedDCRNN(2, 8, 3)
    _randomise(layer, 18)
    X = _rand((3, 5, 40, 2), 19)
    with torch.no_grad():
        out = layer(X, _t(ei), _t(ew))
    return _pack({"X": X, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, {"out": out}, {"K": 3})


@case
def batched_dcrnn_metrla_shape_K3():
    # the reference example's own model: BatchedDCRNN(2, 2, K=3) (examples/indexBatching/DCRNN/pems_bay_main.py:44)
    m = R.load("nn.recurrent.dcrnn")
    ei, ew = syn.sensor_graph(207, 1515, seed=0, symmetric=False)
    layer = m.BatchedDCRNN(2, 2, 3)
    _randomise(layer, 20)
    X = _rand((2, 12, 207, 2), 21)
    with torch.no_grad():
        out = layer(X, _t(ei), _t(ew))
    return _pack({"X": X, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, {"out": out}, {"K": 3})


# ------------------------------------------------------------------------------------------------ TGCN / A3TGCN

@case
def tgcn_sensor():
    m = R.load("nn.recurrent.temporalgcn")
    ei, ew = syn.sensor_graph(50, 330, seed=8, symmetric=False)
    layer = m.TGCN(4, 16)
    _randomise(layer, 22)
    X, H0 = _rand((50, 4), 23), _rand((50, 16), 24)
    with torch.no_grad():
        o1 = layer(X, _t(ei))
        o2 = layer(X, _t(ei), _t(ew))
        o3 = layer(X, _t(ei), _t(ew), H0)
        imp = m.TGCN(4, 16, improved=True)
        imp.load_state_dict(layer.state_dict())
        o4 = imp(X, _t(ei), _t(ew), H0)
    return _pack({"X": X, "H0": H0, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer,
                 {"H_noweight": o1, "H_weight": o2, "H_weight_hidden": o3, "H_improved": o4})


@case
def tgcn2_sensor():
    m = R.load("nn.recurrent.temporalgcn")
    ei, ew = syn.sensor_graph(40, 270, seed=9, symmetric=False)
    layer = m.TGCN2(2, 8, batch_size=3)
    _randomise(layer, 25)
    X, H0 = _rand((3, 40, 2), 26), _rand((3, 40, 8), 27)
    with torch.no_grad():
        o1 = layer(X, _t(ei), _t(ew))
        o2 = layer(X, _t(ei), _t(ew), H0)
    return _pack({"X": X, "H0": H0, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer,
                 {"H_weight": o1, "H_weight_hidden": o2})


def _fix_attention(layer, seed):
    # A3TGCN creates `_attention` on cuda when available (attentiontemporalgcn.py:48-49); here it is CPU
    _randomise(layer, seed)
    with torch.no_grad():
        layer._attention.copy_(_rand(layer._attention.shape, seed + 100, 0.0, 1.0))


@case
def a3tgcn_sensor():
    m = R.load("nn.recurrent.attentiontemporalgcn")
    ei, ew = syn.sensor_graph(50, 330, seed=10, symmetric=False)
    layer = m.A3TGCN(4, 16, periods=5)
    _fix_attention(layer, 28)
    X, H0 = _rand((50, 4, 5), 29), _rand((50, 16), 30)
    with torch.no_grad():
        o1 = layer(X, _t(ei), _t(ew))
        o2 = layer(X, _t(ei), _t(ew), H0)
    return _pack({"X": X, "H0": H0, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer,
                 {"H_weight": o1, "H_weight_hidden": o2}, {"periods": 5})


@case
def a3tgcn2_sensor():
    m = R.load("nn.recurrent.attentiontemporalgcn")
    ei, ew = syn.sensor_graph(40, 270, seed=11, symmetric=False)
    layer = m.A3TGCN2(2, 8, periods=4, batch_size=3)
    _fix_attention(layer, 31)
    X, H0 = _rand((3, 40, 2, 4), 32), _rand((3, 40, 8), 33)
    with torch.no_grad():
        o1 = layer(X, _t(ei), _t(ew))
        o2 = layer(X, _t(ei), _t(ew), H0)
    return _pack({"X": X, "H0": H0, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer,
                 {"H_weight": o1, "H_weight_hidden": o2}, {"periods": 4})


# ------------------------------------------------------------------------------------------------ Chebyshev family

@case
def stconv_sensor():
    m = R.load("nn.attention.stgcn")
    ei, ew = syn.sensor_graph(30, 200, seed=12, symmetric=False)
    outs, layer = {}, None
    X = _rand((2, 7, 30, 4), 34)
    for norm in ("sym", "rw"):
        layer_n = m.STConv(30, 4, 8, 6, kernel_size=3, K=3, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, 35)
        else:
            layer_n.load_state_dict(layer.state_dict())
        layer_n.eval()   # BatchNorm2d in inference mode: running stats (0, 1)
        with torch.no_grad():
            outs["out_" + norm] = layer_n(X, _t(ei), _t(ew))
    layer.train()
    with torch.no_grad():
        outs["out_sym_train"] = layer(X, _t(ei), _t(ew))
    return _pack({"X": X, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, outs, {"K": 3, "kernel_size": 3})


@case
def chebconvattention_sensor():
    m = R.load("nn.attention.astgcn")
    ei, ew = syn.sensor_graph(30, 200, seed=13, symmetric=False)
    X = _rand((3, 30, 4), 36)
    S = torch.softmax(_rand((3, 30, 30), 37, -2, 2), dim=1)
    outs, layer = {}, None
    for norm, lam in (("sym", None), ("rw", 2.3), (None, 3.1)):
        layer_n = m.ChebConvAttention(4, 8, 3, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, 38)
        else:
            layer_n.load_state_dict(layer.state_dict())
        with torch.no_grad():
            kw = {} if lam is None else {"lambda_max": torch.tensor(lam)}
            outs["out_" + str(norm)] = layer_n(X, _t(ei), S, _t(ew), **kw)
            outs["out_noweight_" + str(norm)] = layer_n(X, _t(ei), S, **kw)
    return _pack({"X": X, "S": S, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, outs,
                 {"K": 3, "lambda_rw": 2.3, "lambda_none": 3.1})


def _two_graph_batch(seed):
    """A disjoint batch of two sensor graphs (14 + 16 nodes) as one edge list with a `batch` label per node."""
    ei_a, ew_a = syn.sensor_graph(14, 60, seed=seed, symmetric=False)
    ei_b, ew_b = syn.sensor_graph(16, 80, seed=seed + 1, symmetric=False)
    ei = np.concatenate([ei_a, ei_b + 14], axis=1)
    ew = np.concatenate([ew_a, ew_b])
    batch = torch.tensor([0] * 14 + [1] * 16)
    return _t(ei), _t(ew), batch


@case
def chebconvattention_graphs():
    """ChebConvAttention with `batch` and one lambda_max per graph (astgcn.py:97-98; test/attention_test.py:205-217 pins the
    shapes only).  Also `batch` beside a single lambda_max / none: the labels change nothing there."""
    m = R.load("nn.attention.astgcn")
    ei, ew, batch = _two_graph_batch(140)
    X = _rand((3, 30, 4), 141)
    S = torch.softmax(_rand((3, 30, 30), 142, -2, 2), dim=1)
    lam = torch.tensor([2.0, 3.0])
    outs, layer = {}, None
    for norm in ("sym", "rw", None):
        layer_n = m.ChebConvAttention(4, 8, 3, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, 143)
        else:
            layer_n.load_state_dict(layer.state_dict())
        with torch.no_grad():
            outs["out_graphs_" + str(norm)] = layer_n(X, ei, S, ew, batch, lam)
            outs["out_graphs_noweight_" + str(norm)] = layer_n(X, ei, S, None, batch, lam)
            outs["out_batch_scalar_" + str(norm)] = layer_n(X, ei, S, ew, batch, torch.tensor(2.5))
    with torch.no_grad():
        outs["out_batch_nolambda_sym"] = layer(X, ei, S, ew, batch)
    # gradients of the per-graph form ("sym"): loss = sum(out * G)
    G = _rand((3, 30, 8), 144)
    Xg, Sg = X.clone().requires_grad_(True), S.clone().requires_grad_(True)
    layer.zero_grad()
    (layer(Xg, ei, Sg, ew, batch, lam) * G).sum().backward()
    outs.update({"grad_X": Xg.grad, "grad_S": Sg.grad, "grad__weight": layer._weight.grad, "grad__bias": layer._bias.grad})
    return _pack({"X": X, "S": S, "edge_index": ei, "edge_weight": ew, "batch": batch, "lambda_max": lam, "G": G}, layer, outs,
                 {"K": 3, "lambda_scalar": 2.5})


@case
def chebconv_graphs():
    """PyG ChebConv (restated, oracle/pyg_restated.py) with `batch` and one lambda_max per graph: the same selection rule
    (`lambda_max[batch[edge_index[0]]]` over the Laplacian's entries, diagonal included)."""
    from . import pyg_restated as P
    ei, ew, batch = _two_graph_batch(150)
    X = _rand((30, 5), 151)
    lam = torch.tensor([1.7, 2.6])
    outs, layer = {}, None
    for norm in ("sym", "rw", None):
        layer_n = P.ChebConv(5, 7, 3, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, 152)
        else:
            layer_n.load_state_dict(layer.state_dict())
        with torch.no_grad():
            outs["out_graphs_" + str(norm)] = layer_n(X, ei, ew, batch, lam)
            outs["out_batch_scalar_" + str(norm)] = layer_n(X, ei, ew, batch, 2.2)
    return _pack({"X": X, "edge_index": ei, "edge_weight": ew, "batch": batch, "lambda_max": lam}, layer, outs,
                 {"K": 3, "lambda_scalar": 2.2})


@case
def stconv_sensor_grads():
    """STConv (stgcn.py:86-168) in TRAINING mode with every gradient: loss = sum(out * G); dX, the parameter gradients and the
    updated BatchNorm running statistics, from the reference's own module file under torch autograd."""
    m = R.load("nn.attention.stgcn")
    ei, ew = syn.sensor_graph(30, 200, seed=12, symmetric=False)
    X = _rand((2, 7, 30, 4), 160)
    layer = m.STConv(30, 4, 8, 6, kernel_size=3, K=3, normalization="sym")
    _randomise(layer, 161)
    layer.train()
    Xg = X.clone().requires_grad_(True)
    out = layer(Xg, _t(ei), _t(ew))
    G = _rand(tuple(out.shape), 162)
    (out * G).sum().backward()
    outs = {"out_train": out.detach(), "grad_X": Xg.grad}
    for k, p in layer.named_parameters():
        outs["grad_" + k] = p.grad
    outs["running_mean"] = layer._batch_norm.running_mean.clone()
    outs["running_var"] = layer._batch_norm.running_var.clone()
    # parameters as they were when the forward ran (state_dict below holds the updated running statistics too)
    return _pack({"X": X, "edge_index": _t(ei), "edge_weight": _t(ew), "G": G}, layer, outs, {"K": 3, "kernel_size": 3})


# ------------------------------------------------------------------------------------------------ ASTGCN / MSTGCN

@case
def astgcn_sensor():
    m = R.load("nn.attention.astgcn")
    ei, _ = syn.sensor_graph(24, 150, seed=90, symmetric=False)
    X = _rand((3, 24, 2, 8), 91)
    outs, layer = {}, None
    for norm in ("sym", None, "rw"):   # "rw": lambda_max still comes from the UNNORMALISED Laplacian (astgcn.py:438)
        layer_n = m.ASTGCN(2, 2, 3, 6, 5, 2, 4, 8, 24, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, 92)
        else:
            layer_n.load_state_dict(layer.state_dict())
        with torch.no_grad():
            outs["out_" + str(norm)] = layer_n(X, _t(ei))
    return _pack({"X": X, "edge_index": _t(ei)}, layer, outs, {"args": [2, 2, 3, 6, 5, 2, 4, 8, 24]})


@case
def mstgcn_sensor():
    m = R.load("nn.attention.mstgcn")
    ei, _ = syn.sensor_graph(24, 150, seed=93, symmetric=False)
    X = _rand((3, 24, 2, 8), 94)
    layer = m.MSTGCN(2, 2, 3, 6, 6, 2, 4, 8)   # (the reference needs nb_chev_filter == nb_time_filter, mstgcn.py:86-88)
    _randomise(layer, 95)
    with torch.no_grad():
        out = layer(X, _t(ei))
    return _pack({"X": X, "edge_index": _t(ei)}, layer, {"out": out}, {"args": [2, 2, 3, 6, 6, 2, 4, 8]})


# ------------------------------------------------------------------------------------------------ ChebConv cells

def _cheb_cell_case(modname, clsname, seed, lstm):
    m = R.load("nn.recurrent." + modname)
    ei, ew = syn.sensor_graph(36, 240, seed=seed, symmetric=False)
    outs, layer = {}, None
    X, H0, C0 = _rand((36, 5), seed + 1), _rand((36, 7), seed + 2), _rand((36, 7), seed + 3)
    for norm, lam in (("sym", None), ("rw", 2.4)):
        layer_n = getattr(m, clsname)(5, 7, 3, normalization=norm)
        if layer is None:
            layer = layer_n
            _randomise(layer, seed + 4)
        else:
            layer_n.load_state_dict(layer.state_dict())
        kw = {} if lam is None else {"lambda_max": torch.tensor(lam)}
        with torch.no_grad():
            if lstm:
                h1, c1 = layer_n(X, _t(ei), _t(ew), **kw)
                h2, c2 = layer_n(X, _t(ei), _t(ew), H0, C0, **kw)
                outs.update({f"H_{norm}": h1, f"C_{norm}": c1, f"H_state_{norm}": h2, f"C_state_{norm}": c2})
            else:
                outs.update({f"H_{norm}": layer_n(X, _t(ei), _t(ew), **kw),
                             f"H_state_{norm}": layer_n(X, _t(ei), _t(ew), H0, **kw)})
    return _pack({"X": X, "H0": H0, "C0": C0, "edge_index": _t(ei), "edge_weight": _t(ew)}, layer, outs,
                 {"K": 3, "lambda_rw": 2.4})


@case
def gconvgru_sensor():
    return _cheb_cell_case("gconv_gru", "GConvGRU", 60, False)


@case
def gconvlstm_sensor():
    return _cheb_cell_case("gconv_lstm", "GConvLSTM", 70, True)


@case
def gclstm_sensor():
    return _cheb_cell_case("gc_lstm", "GCLSTM", 80, True)


# ------------------------------------------------------------------------------------------------ EvolveGCN

def _dynamic_graphs(n, steps, seed):
    rng = np.random.default_rng(seed)
    eis, ews = [], []
    for s in range(steps):
        e = int(rng.integers(4 * n, 8 * n))
        ei, ew = syn.sensor_graph(n, e + n, seed=seed * 100 + s, symmetric=False)
        eis.append(ei)
        ews.append((ew * rng.uniform(10, 1000)).astype(np.float32))   # covid-style large weights
    return eis, ews


@case
def evolvegcnh_dynamic():
    m = R.load("nn.recurrent.evolvegcnh")
    n, F, steps = 40, 8, 4
    eis, ews = _dynamic_graphs(n, steps, 14)
    layer = m.EvolveGCNH(n, F)
    _randomise(layer, 39)
    inputs, outs = {}, {}
    with torch.no_grad():
        for s in range(steps):
            X = _rand((n, F), 40 + s)
            inputs[f"X{s}"], inputs[f"edge_index{s}"], inputs[f"edge_weight{s}"] = X, _t(eis[s]), _t(ews[s])
            outs[f"out{s}"] = layer(X, _t(eis[s]), _t(ews[s]))
    return _pack(inputs, layer, outs, {"steps": steps, "num_nodes": n})


@case
def evolvegcno_dynamic():
    m = R.load("nn.recurrent.evolvegcno")
    n, F, steps = 40, 8, 4
    eis, ews = _dynamic_graphs(n, steps, 15)
    layer = m.EvolveGCNO(F)
    _randomise(layer, 50)
    inputs, outs = {}, {}
    with torch.no_grad():
        for s in range(steps):
            X = _rand((n, F), 51 + s)
            inputs[f"X{s}"], inputs[f"edge_index{s}"], inputs[f"edge_weight{s}"] = X, _t(eis[s]), _t(ews[s])
            outs[f"out{s}"] = layer(X, _t(eis[s]), _t(ews[s]))
    return _pack(inputs, layer, outs, {"steps": steps, "num_nodes": n})



# ------------------------------------------------------------------------------------------------ BASELINE.json shapes
# Inputs are rebuilt from seeds on both sides (oracle/baseline_cases.py); the fixture stores the parameters, a sample of
# the reference output and fp64 checksums per slice.

@case
def baseline_c2_batched_dcrnn64():
    """config 2 at the BENCHMARKED model: BatchedDCRNN(2, 64, K=3), 207 nodes, B = 64, 12 steps, on the 1 515-edge graph
    of BASELINE.json and the 1 722-edge variant (the reference's data)."""
    from . import baseline_cases as BC
    m = R.load("nn.recurrent.dcrnn")
    layer = m.BatchedDCRNN(2, 64, 3)
    BC.randomise(layer, 210, gain=0.25)
    outs = {}
    for E in (1515, 1722):
        ei, ew, X = BC.metrla(E)
        with torch.no_grad():
            out = layer(X, ei, ew)                                   # [64, 12, 207, 64]
        outs[f"sample_E{E}"] = out[list(BC.METRLA_SAMPLE_B)][:, list(BC.METRLA_SAMPLE_T)].contiguous()
        outs[f"sums_bt_E{E}"] = BC.slice_sums(out, (0, 1))
        outs[f"abs_sums_bt_E{E}"] = BC.slice_sums(out.abs(), (0, 1))
    return _pack({}, layer, outs, {"K": 3})


@case
def baseline_c3_a3tgcn2_pemsbay():
    """config 3: A3TGCN2(2, 32, periods=12, batch_size=64) on the PeMS-BAY-shaped graph (325 nodes / 2 694 edges)."""
    from . import baseline_cases as BC
    m = R.load("nn.recurrent.attentiontemporalgcn")
    ei, ew, X, H0 = BC.pemsbay()
    layer = m.A3TGCN2(2, 32, periods=12, batch_size=64)
    _fix_attention(layer, 310)
    with torch.no_grad():
        o1 = layer(X, ei, ew)
        o2 = layer(X, ei, ew, H0)
    sel = list(BC.PEMSBAY_SAMPLE_B)
    return _pack({}, layer, {"sample_weight": o1[sel].contiguous(), "sample_weight_hidden": o2[sel].contiguous(),
                             "sums_weight": BC.slice_sums(o1, (0, 1)), "sums_weight_hidden": BC.slice_sums(o2, (0, 1))},
                 {"periods": 12})


@case
def baseline_c4_tgcn2_50k():
    """config 4: TGCN2(2, 32, batch_size=8) on 50 000 nodes / 400 000 edges (locality-ordered and uniform-random)."""
    from . import baseline_cases as BC
    m = R.load("nn.recurrent.temporalgcn")
    layer = m.TGCN2(2, 32, batch_size=8)
    BC.randomise(layer, 410)
    nodes = BC.sample_nodes_50k()
    outs = {}
    for kind in ("local", "uniform"):
        ei, ew, X, H0 = BC.graph50k(kind)
        with torch.no_grad():
            out = layer(X, ei, ew, H0)                               # [8, 50 000, 32]
        outs[f"sample_{kind}"] = out[:, nodes].contiguous()
        # one checksum per batch entry and block of 500 consecutive nodes
        outs[f"sums_{kind}"] = out.double().view(8, 100, 500 * 32).sum(-1)
    return _pack({"sample_nodes": nodes}, layer, outs)


@case
def baseline_c5_evolvegcnh_covid():
    """config 5: the vendored dataset/england_covid.json (129 regions, a new directed graph with weights up to 9.6e5
    every day) through EvolveGCNH(129, 8), every snapshot the reference loader yields (lags = 8: 53 snapshots)."""
    from . import baseline_cases as BC
    m = R.load("nn.recurrent.evolvegcnh")
    ds = R.load_dataset("encovid").EnglandCovidDatasetLoader
    # the reference loader downloads its JSON; feed it the vendored copy instead
    with open(os.path.join(R.REFERENCE_ROOT, "dataset", "england_covid.json")) as f:
        raw = json.load(f)
    loader = ds.__new__(ds)
    loader._dataset = raw
    signal = loader.get_dataset(lags=8)
    layer = m.EvolveGCNH(129, 8)
    BC.randomise(layer, 510)
    outs, wmax = [], 0.0
    with torch.no_grad():
        for snap in signal:
            outs.append(layer(snap.x, snap.edge_index, snap.edge_attr))
            wmax = max(wmax, float(snap.edge_attr.max()))
    return _pack({}, layer, {"out": torch.stack(outs)}, {"snapshots": len(outs), "max_weight": wmax})


@case
def baseline_c1_chickenpox_epoch():
    """config 1: the reference example's loop (examples/recurrent/dcrnn_example.py:19-46) with the reference's own modules:
    ChickenpoxDatasetLoader (fed from the vendored JSON) -> temporal_signal_split(0.2) -> DCRNN(4, 32, 1) + Linear(32, 1),
    cost = mean over the 103 train snapshots of the MSE, one backward.  Stored: the cost, the first predictions, every
    parameter gradient."""
    from . import baseline_cases as BC
    m = R.load("nn.recurrent.dcrnn")
    ds_mod = R.load_dataset("chickenpox")
    loader = ds_mod.ChickenpoxDatasetLoader.__new__(ds_mod.ChickenpoxDatasetLoader)
    with open(os.path.join(R.REFERENCE_ROOT, "dataset", "chickenpox.json")) as f:
        loader._dataset = json.load(f)
    dataset = loader.get_dataset()
    # temporal_signal_split(dataset, train_ratio=0.2) (signal/train_test_split.py: int(ratio * snapshot_count) leading
    # snapshots; the module itself imports the batch iterators, which need torch_geometric.data.Batch)
    train = dataset[0:int(0.2 * dataset.snapshot_count)]

    class RecurrentGCN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.recurrent = m.DCRNN(4, 32, 1)
            self.linear = torch.nn.Linear(32, 1)

        def forward(self, x, edge_index, edge_weight):
            return self.linear(torch.relu(self.recurrent(x, edge_index, edge_weight)))

    model = RecurrentGCN()
    BC.randomise(model, 610, gain=0.5)
    cost, preds, n = 0, [], 0
    for snap in train:
        y_hat = model(snap.x, snap.edge_index, snap.edge_attr)
        if n < 3:
            preds.append(y_hat.detach())
        cost = cost + torch.mean((y_hat - snap.y) ** 2)
        n += 1
    cost = cost / n
    cost.backward()
    outs = {"cost": cost.detach(), "pred_head": torch.stack(preds)}
    for k, p in model.named_parameters():
        outs["grad/" + k] = p.grad.detach()
    return _pack({}, model, outs, {"snapshots": n})

# ------------------------------------------------------------------------------------------------ signal iterator

@case
def chickenpox_signal_head():
    """First snapshots of the Chickenpox signal as the reference's iterator yields them
    (signal/static_graph_temporal_signal.py:103-134 over dataset/chickenpox.py:57-81, lags = 4)."""
    sig = R.load("signal.static_graph_temporal_signal")
    ei, ew, fx = chickenpox_graph()
    lags = 4
    feats = [fx[i:i + lags, :].T for i in range(fx.shape[0] - lags)]
    targs = [fx[i + lags, :].T for i in range(fx.shape[0] - lags)]
    s = sig.StaticGraphTemporalSignal(ei, ew, feats, targs)
    out = {"edge_index": _t(ei), "edge_weight": _t(ew), "snapshot_count": torch.tensor(s.snapshot_count)}
    for t, snap in enumerate(s):
        if t >= 3:
            break
        out[f"x{t}"], out[f"y{t}"] = snap.x, snap.y
        assert torch.equal(snap.edge_index, _t(ei))
    return _pack({"FX_head": fx[:8]}, None, out, {"lags": lags})


def main(argv):
    if not R.reference_available():
        raise SystemExit(f"{R.REFERENCE_ROOT} not found: fixtures can only be generated where the reference is mounted")
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    pats = argv[1:]
    for name, fn in CASES.items():
        if pats and not any(p in name for p in pats):
            continue
        torch.manual_seed(0)
        d = fn()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main(sys.argv)
