"""GPU probe: the split-bf16 GEMM (csrc/gemm_bx.hip) against the fp32 tile kernels at the DCRNN training-step shapes."""
import time
import torch
from pytorch_geometric_temporal_amd import _lib, ops

lib = _lib.get_lib()
dev = torch.device("cuda:0")
M, S, C, O, T = 1024 * 207, 5, 66, 64, 3
g = torch.Generator().manual_seed(0)
TS = torch.randn(S, T, M, C, generator=g).to(dev)          # the diffusion stack of T time steps: segment stride T * M * C
A = TS[:, 1]
Wzr, bzr = (torch.randn(S * C, 2 * O, generator=g) / 18).to(dev), torch.randn(2 * O, generator=g).to(dev)
Wh, bh = (torch.randn(S * C, O, generator=g) / 18).to(dev), torch.randn(O, generator=g).to(dev)
H = torch.randn(M, O, generator=g).to(dev)
zr, xhr = torch.empty(M, 2 * O, device=dev), torch.zeros(M, C, device=dev)
ht, out0 = torch.empty(M, O, device=dev), torch.empty(M, O, device=dev)
dG = torch.randn(M, 2 * O, generator=g).to(dev)
dTS = torch.empty(4, M, O, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


cases = {
    "NN+zr 330->128": lambda: ops.gemm_gru_zr(A, C, T * M * C, S, C, Wzr, 2 * O, 1, bzr, zr, H, xhr, 2),
    "NN+h 330->64": lambda: ops.gemm_gru_h(A, C, T * M * C, S, C, Wh, O, 1, bh, ht, zr, H, out0, None),
    "NN 330->128": lambda: ops.gemm(A, C, T * M * C, S, C, Wzr, 2 * O, 1, zr, 2 * O, 0, 2 * O, bzr, M, 2 * O),
    "NT 128->256": lambda: ops.gemm(dG, 2 * O, 0, 1, 2 * O, Wzr[:4 * O], 1, 2 * O, dTS, O, M * O, O, None, M, 4 * O),
    "NT 64->256": lambda: ops.gemm(ht, O, 0, 1, O, Wh[:4 * O], 1, O, dTS, O, M * O, O, None, M, 4 * O),
    "NT 128->64": lambda: ops.gemm(dG, 2 * O, 0, 1, 2 * O, Wzr[:O], 1, 2 * O, out0, O, 0, O, None, M, O),
    "NT 64->64": lambda: ops.gemm(ht, O, 0, 1, O, Wh[:O], 1, O, out0, O, 0, O, None, M, O),
}
for name, fn in cases.items():
    res = {}
    for bx in (1, 0):
        lib.tune("gemm_bx", bx)
        res[bx] = timeit(fn)
    lib.tune("gemm_bx", 1)
    print(f"{name:16s} split-bf16 {res[1]:7.1f} us   fp32 MFMA {res[0]:7.1f} us")

# ---- does a split-bf16 launch slow the kernels that follow it (clock / power management)?  A fixed fp32 product
# (128 -> 64, never on the split-bf16 path) is timed with device events right after a 330 -> 128 product run either way.
ref_fn = cases["NT 128->64"]
big_fn = cases["NN 330->128"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for bx in (1, 0, 1, 0):
    lib.tune("gemm_bx", bx)
    tb, tr = [], []
    for it in range(30):
        ev[0].record(); big_fn(); ev[1].record(); ref_fn(); ev[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            tb.append(ev[0].elapsed_time(ev[1]) * 1e3); tr.append(ev[1].elapsed_time(ev[2]) * 1e3)
    print(f"gemm_bx={bx}: 330->128 {sum(tb) / len(tb):7.1f} us, then the fp32 128->64 product {sum(tr) / len(tr):6.1f} us")
lib.tune("gemm_bx", 1)
