"""Phase timeline of csrc/seq64.hip's forward kernel (lab/libseq64_lab.so: the same source with wall_clock64 marks): per cell step
and gate, the microseconds workgroup 0 spends up to each mark.  usage: seq64_trace.py [B = 256] [edges = 1515]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import _lib, ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN
from pytorch_geometric_temporal_amd.nn.recurrent.dcrnn import _cell_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1515
dev = torch.device("cuda:0")
lab = ctypes.CDLL(os.path.join(ROOT, "lab", f"libseq64_lab{sys.argv[3] if len(sys.argv) > 3 else ''}.so"))
print(f"== {os.path.basename(lab._name)}")
N, T, Fin, O, K = 207, 12, 2, 64, 3
S, C, M = 2 * K - 1, Fin + O, B * N
ei_np, ew_np = syn.sensor_graph(N, E, seed=0, symmetric=False)
g = ops.dconv_graph(torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev), N)
torch.manual_seed(0)
model = BatchedDCRNN(Fin, O, K).to(dev)
with torch.no_grad():
    Wzr, bzr, Wh, bh = [t.contiguous() for t in _cell_weights(model.conv_x_z, model.conv_x_r, model.conv_x_h)]
X = torch.randn(B, T, N, Fin, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
lab.pgt_dcrnn_seq64_pack_floats.restype = ctypes.c_int64
Wp = torch.empty(lab.pgt_dcrnn_seq64_pack_floats(ctypes.c_int64(K)), **f32)
TSzr, TSh = torch.empty(S, T, M, C, **f32), torch.empty(S, T, M, C, **f32)
ZR, HT, out = torch.empty(T, M, 2 * O, **f32), torch.empty(T, M, O, **f32), torch.empty(B, T, N, O, **f32)
P, I = ctypes.c_void_p, ctypes.c_int64
p = lambda t: P(t.data_ptr())
so, si = g.fwd_o.struct(), g.fwd_i.struct()
stream = P(torch.cuda.current_stream().cuda_stream)
rc = lab.pgt_dcrnn_seq64_pack_f32(p(Wzr), p(Wh), I(Fin), I(K), p(Wp), stream)
assert rc == 0, rc


def launch():
    rc = lab.pgt_dcrnn_seq64_f32(ctypes.byref(so), ctypes.byref(si), I(g.E), I(g.E), I(N), p(X), I(T * N * Fin), I(N * Fin), P(None), p(Wp),
                                 p(Wzr), p(bzr), p(Wh), p(bh), I(B), I(T), I(Fin), I(K), p(out), I(T * N * O), I(N * O), p(TSzr), p(TSh),
                                 I(T * M * C), I(M * C), p(ZR), p(HT), stream)
    assert rc == 0, ctypes.c_char_p(lab.sq_lab_last_error()).value


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    launch()
e1.record()
torch.cuda.synchronize()
print(f"B = {B}, {E} edges: {e0.elapsed_time(e1) * 100:.1f} us per launch = {e0.elapsed_time(e1) * 100 / T:.1f} us per cell step")
SLOTS = 9
tr = torch.zeros(T * 2 * SLOTS, dtype=torch.int64, device=dev)
assert lab.sq_lab_set_trace(p(tr)) == 0
launch()
torch.cuda.synchronize()
lab.sq_lab_set_trace(P(None))
tr = tr.cpu().view(T, 2, SLOTS).double() / 100.0       # us
names = ["T0 ready", "hop1", "mfma T0,T1o", "hop2o", "mfma T2o", "T1i->LDS", "hop2i", "mfma T1i,T2i", "gates"]
print("marks: " + " | ".join(names))
for t in (1, 5, 11):
    for G in (0, 1):
        prev = tr[t, G - 1, SLOTS - 1] if G == 1 else tr[t - 1, 1, SLOTS - 1]
        d = []
        for sl in range(SLOTS):
            d.append(float(tr[t, G, sl] - prev))
            prev = tr[t, G, sl]
        print(f"  step {t:2d} {'z|r ' if G == 0 else 'cand'}: " + " ".join(f"{x:6.2f}" for x in d) + f"   = {sum(d):6.2f} us")


# ---- the adjoint kernel: marks per (step, gate: 0 candidate, 1 z | r): 0 gate adjoint done | 1 G2o, G1o | 2 B += 2 Po^T A | 3 park |
# 4 G2i, G1i | 5 B += 2 Pi^T A | 6 G0 | 7 A += park + Pi^T B | 8 state gradient folded
dOut = torch.randn(B, T, N, O, **f32)
dPzr, dPh = torch.empty(T, M, 2 * O, **f32), torch.empty(T, M, O, **f32)
lab.pgt_dcrnn_seq64_bwd_ws_floats.restype = ctypes.c_int64
nws = lab.pgt_dcrnn_seq64_bwd_ws_floats(I(N), I(B))
ws = torch.empty(nws, **f32)
Wpb = torch.empty_like(Wp)
assert lab.pgt_dcrnn_seq64_pack_bwd_f32(p(Wzr), p(Wh), I(Fin), I(K), p(Wpb), stream) == 0
to, ti = g.bwd_o.struct(), g.bwd_i.struct()


def launch_bwd():
    rc = lab.pgt_dcrnn_seq64_bwd_f32(ctypes.byref(to), ctypes.byref(ti), I(g.E), I(g.E), I(N), p(dOut), I(T * N * O), I(N * O), p(out),
                                     I(T * N * O), I(N * O), P(None), p(ZR), p(HT), p(Wpb), I(B), I(T), I(Fin), I(K), p(dPzr), p(dPh),
                                     P(None), p(ws), I(nws), stream)
    assert rc == 0, ctypes.c_char_p(lab.sq_lab_last_error()).value


for _ in range(3):
    launch_bwd()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    launch_bwd()
e1.record()
torch.cuda.synchronize()
print(f"adjoint, B = {B}: {e0.elapsed_time(e1) * 100:.1f} us per launch = {e0.elapsed_time(e1) * 100 / T:.1f} us per cell step")
tr = torch.zeros(T * 2 * SLOTS, dtype=torch.int64, device=dev)
assert lab.sq_lab_set_trace(p(tr)) == 0
launch_bwd()
torch.cuda.synchronize()
lab.sq_lab_set_trace(P(None))
tr = tr.cpu().view(T, 2, SLOTS).double() / 100.0
print("marks: gate adj | G2o,G1o | B+=2PoA | park | G2i,G1i | B+=2PiA | G0 | A+=park+PiB | fold")
for t in (10, 5, 1):
    for G in (0, 1):
        prev = tr[t + 1, 1, 8] if G == 0 else tr[t, 0, 7]
        d = []
        for sl in range(SLOTS if G == 1 else 8):
            d.append(float(tr[t, G, sl] - prev))
            prev = tr[t, G, sl]
        print(f"  adjoint step {t:2d} {'cand' if G == 0 else 'z|r '}: " + " ".join(f"{x:6.2f}" for x in d) + f"   = {sum(d):6.2f} us")
