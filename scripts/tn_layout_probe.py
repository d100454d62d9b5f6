"""The weight-gradient product over the saved stacks (all T steps of B windows): fp32 atomics against the deterministic two-pass form,
and rows of 66 floats (what the forward saves) against rows padded to 68 (16-byte aligned).  usage: tn_layout_probe.py [B = 1024]"""
import sys
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
S, T, N, O, Fin = 5, 12, 207, 64, 2
M, C = B * N, Fin + O
for ld, det in ((66, False), (66, True), (68, False), (66, False), (66, True), (68, False)):
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = det
    TS = torch.randn(S, T, M, ld, device=dev)
    for NO in (128, 64):
        G = torch.randn(T, M, NO, device=dev)
        dW = torch.zeros(S * C, NO, device=dev)
        db = torch.zeros(NO, device=dev)
        for _ in range(2):
            ops.gemm_tn_acc(TS, ld, T * M * ld, S, C, G, NO, dW, NO, db, T * M, NO)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_tn_acc(TS, ld, T * M * ld, S, C, G, NO, dW, NO, db, T * M, NO)
        e1.record()
        torch.cuda.synchronize()
        print(f"B = {B}: rows of {ld} floats, N = {NO}, {'deterministic' if det else 'atomics'}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    del TS
