"""Host-side cost of one product call on the GPU box: data_ptr(), ctypes argument objects, a ctypes call, torch.empty,
an autograd.Function round trip -- the terms that bound the eager small-graph configs."""
import ctypes
import gc
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.get_lib()
t = torch.empty(20, 32, device=dev)
v = t[:, 4:]


def per(fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e6 * (time.perf_counter() - t0) / n


print("data_ptr() on a cuda tensor      %.2f us" % per(lambda: t.data_ptr()))
print("data_ptr() on a cuda view        %.2f us" % per(lambda: v.data_ptr()))
print("_lib.ptr(t)                      %.2f us" % per(lambda: _lib.ptr(t)))
print("c_void_p(int)                    %.2f us" % per(lambda: ctypes.c_void_p(12345)))
print("torch.cuda.current_stream        %.2f us" % per(lambda: torch.cuda.current_stream(dev).cuda_stream))
print("_lib.stream_of                   %.2f us" % per(lambda: _lib.stream_of(lib, t)))
print("pgt_abi_version() ctypes call    %.2f us" % per(lambda: lib._pgt_abi_version()))
print("torch.empty(20, 32)              %.2f us" % per(lambda: torch.empty(20, 32, device=dev)))
print("t[:, 4:] view                    %.2f us" % per(lambda: t[:, 4:]))
gc.disable()
print("gc off: _lib.ptr(t)              %.2f us" % per(lambda: _lib.ptr(t)))
gc.enable()
torch.cuda.synchronize()
