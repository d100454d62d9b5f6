"""The two T-GCN cell kernels alone at config 4's size (M = B * 50 000 rows, Fin = 2, hidden 32): launch time by graph replay
(pairs of buffer sets rotate: 259 / 310 MB per launch already exceed the Infinity Cache), algorithmic bytes / time against 8 TB/s.
    python scripts/tgcn_cell_probe.py [B=8] [launches=40]            PGT_TUNE=tgcn_rows=0 for the column-per-lane kernels"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd import _lib  # noqa: E402
from pytorch_geometric_temporal_amd.ops import ptr, stream_of  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
lib = _lib.get_lib()
M, Fin, O = B * 50_000, 2, 32
C = Fin + O
torch.manual_seed(0)
sets = []
for _ in range(2):
    sets.append(dict(AX=torch.randn(M, Fin, device=dev), H=torch.randn(M, O, device=dev), ZR=torch.empty(M, 2 * O, device=dev),
                     HT=torch.empty(M, O, device=dev), Hn=torch.empty(M, O, device=dev), dHn=torch.randn(M, O, device=dev),
                     dH=torch.empty(M, O, device=dev)))
Wzr, bzr = torch.randn(C, 2 * O, device=dev) * 0.2, torch.randn(2 * O, device=dev) * 0.1
Wh, bh = torch.randn(C, O, device=dev) * 0.2, torch.randn(O, device=dev) * 0.1
dW = torch.empty(C * 3 * O + 3 * O, device=dev)
dWzr, dWh = dW[:C * 2 * O], dW[C * 2 * O:C * 3 * O]
dbzr, dbh = dW[C * 3 * O:C * 3 * O + 2 * O], dW[C * 3 * O + 2 * O:]
nws = int(lib._pgt_tgcn_cell_bwd_ws_floats(Fin, O))
ws = torch.empty(nws, device=dev)


def fwd(s):
    lib.call("pgt_tgcn_cell_f32", ptr(s["AX"]), Fin, ptr(s["H"]), O, ptr(Wzr), ptr(bzr), ptr(Wh), ptr(bh), M, Fin, O, ptr(s["ZR"]),
             ptr(s["HT"]), ptr(s["Hn"]), O, stream_of(lib, s["H"]))


def bwd(s):
    lib.call("pgt_tgcn_cell_bwd_f32", ptr(s["dHn"]), O, ptr(s["AX"]), Fin, ptr(s["H"]), O, ptr(s["ZR"]), ptr(s["HT"]), ptr(Wzr), ptr(Wh),
             M, Fin, O, ptr(s["dH"]), O, ptr(dWzr), ptr(dbzr), ptr(dWh), ptr(dbh), ptr(ws), nws, stream_of(lib, s["H"]))


def timed(fn):
    for s in sets:
        fn(s)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(launches):
                fn(sets[i % 2])
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / launches)
    return sum(ts) / len(ts)


t_f = timed(fwd)
t_b = timed(bwd)                      # (includes the partial-sum reduction launch that follows the adjoint kernel)
bf, bb = 4.0 * M * (Fin + 5 * O), 4.0 * M * (Fin + 6 * O)
print(f"B = {B} (M = {M}): forward {t_f:.1f} us = {bf / t_f / 1e3 / 8000:.3f} of 8 TB/s ({bf / 1e6:.0f} MB); "
      f"adjoint + reduction {t_b:.1f} us = {bb / t_b / 1e3 / 8000:.3f} ({bb / 1e6:.0f} MB)  [PGT_TUNE={os.environ.get('PGT_TUNE', '')}]")
