"""Host-side wrappers over the C ABI (include/pgt_hip.h): graph handles, raw kernel calls on torch-owned device
memory, and the autograd Functions the nn.Module mirrors are built from.  PyTorch is plumbing here: it owns the
buffers and the stream; every arithmetic step on the path is a HIP kernel behind the C ABI.

One module per model family (round 5 kept all of this in one 2 800-line file):

    _graphs   Csr / Ellw / DConvGraph / SymGraph, graph preparation, the graph cache
    _core     kernel timer, raw kernel calls (aggregation, products, gates, element-wise), schedule switches
    dcrnn     DConv, DCRNN, BatchedDCRNN: stacks, cells, whole-sequence Functions
    tgcn      TGCN / TGCN2 / A3TGCN cell and its packed weights
    cheb      ChebConv, GConvGRU / GConvLSTM / GCLSTM cells, LSTM gates
    astgcn    ASTGCN / MSTGCN: attention scores, batched products, ChebConvAttention
    evolve    EvolveGCN-H / -O weight evolution, the small-graph GCN layer
    stconv    STConv: TemporalConv, node-wise batch norm
    generic   aggregation / linear / read-out Functions shared by the families

`ops.NAME` is the public spelling of every name below, as before.  The package is also where the SWITCHES are flipped
(`ops.USE_SEQ64 = False`, `ops.KERNEL_TIMER = timer`, `monkeypatch.setattr(ops, "slab_fits", ...)`): a write to an attribute of
the package is forwarded to every family module that holds the name — the module that defines it and the ones that
imported it — so the code that reads the switch sees it whichever module it lives in.
"""
import sys
import types

from . import _graphs, _core, dcrnn, generic, tgcn, cheb, astgcn, evolve, stconv

_FAMILIES = (_graphs, _core, dcrnn, generic, tgcn, cheb, astgcn, evolve, stconv)

for _m in _FAMILIES:
    for _k, _v in vars(_m).items():
        if not (_k.startswith("__") and _k.endswith("__")):
            globals()[_k] = _v
del _m, _k, _v


class _Ops(types.ModuleType):
    def __setattr__(self, name, value):
        for m in _FAMILIES:
            if name in m.__dict__:
                setattr(m, name, value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _Ops
