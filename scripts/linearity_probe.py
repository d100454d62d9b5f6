#!/usr/bin/env python
"""GPU diagnostic: run-to-run spread and linearity (in the loss weights) of the BatchedDCRNN parameter gradients at the
benchmark batch, with the split-bf16 kernels on / off and with atomics-free weight gradients."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import baseline_cases as BC  # noqa: E402
from pytorch_geometric_temporal_amd import _lib, ops  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    ei, ew, _ = BC.metrla(1515)
    X = BC.rand((B, 12, 207, 2), 1201).to(dev)
    w = BC.rand((B, 12, 207, 64), 1202).to(dev)
    torch.manual_seed(0)
    m = BatchedDCRNN(2, 64, 3)
    BC.randomise(m, 7, gain=0.25)
    m = m.to(dev)
    eid, ewd = ei.to(dev), ew.to(dev)

    def grads(wm):
        m.zero_grad()
        out = m(X, eid, ewd)
        (out * wm).sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in m.parameters()]

    lo, hi = w.clone(), w.clone()
    lo[B // 2:] = 0
    hi[:B // 2] = 0
    for label, bx, det in (("split-bf16 + atomics", 1, False), ("exact fp32 + atomics", 0, False), ("split-bf16 + deterministic", 1, True),
                           ("exact fp32 + deterministic", 0, True)):
        lib.tune("gemm_bx", bx)
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = det
        o1, g1 = grads(w)
        o2, g2 = grads(w)
        _, gl = grads(lo)
        _, gh = grads(hi)
        names = [n for n, _ in m.named_parameters()]
        print(f"== {label}: forward run-to-run max diff {float((o1 - o2).abs().max()):.3e}")
        for n, a, a2, b, c in zip(names, g1, g2, gl, gh):
            print(f"   {n:22s} |g|max {float(a.abs().max()):10.3f}  run-to-run {float((a - a2).abs().max()):.3e}  "
                  f"linearity {float((a - (b + c)).abs().max()):.3e}  |lo|max {float(b.abs().max()):.3f} |hi|max {float(c.abs().max()):.3f}")
    lib.tune("gemm_bx", 1)
    ops.DETERMINISTIC_WEIGHT_GRADIENTS = False


if __name__ == "__main__":
    main()
