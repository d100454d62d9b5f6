"""The skinny read-out torch's own F.linear runs behind BatchedDCRNN ([2.5 M, 64] x [64, 2], forward + backward): time under each BLAS
preference torch offers (torch.backends.cuda.preferred_blas_library) and with TunableOp, next to the package's streaming kernels."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd import ops

dev = torch.device("cuda:0")
M = 1024 * 12 * 207
x = torch.randn(M, 64, device=dev, requires_grad=True)
lin = torch.nn.Linear(64, 2).to(dev)


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def torch_fb():
    x.grad = None
    lin.zero_grad()
    torch.nn.functional.linear(x, lin.weight, lin.bias).sum().backward()


def ours_fb():
    x.grad = None
    lin.zero_grad()
    ops.linear(x, lin.weight.t(), lin.bias).sum().backward()


print(f"package streaming kernels (ops.linear): {bench(ours_fb):.3f} ms forward + backward")
for pref in ("default", "cublas", "cublaslt", "hipblaslt"):
    try:
        if pref != "default":
            torch.backends.cuda.preferred_blas_library(pref)
        print(f"torch F.linear, preferred_blas_library = {pref}: {bench(torch_fb):.3f} ms (now: {torch.backends.cuda.preferred_blas_library()})")
    except Exception as e:
        print(f"preferred_blas_library({pref}): {e!r}")
try:
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.set_max_tuning_duration(500)
    print(f"torch F.linear, TunableOp on: {bench(torch_fb, 5):.3f} ms")
    tunable.enable(False)
except Exception as e:
    print(f"TunableOp: {e!r}")
