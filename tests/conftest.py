import glob
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_geometric_temporal_amd import _lib  # noqa: E402
from pytorch_geometric_temporal_amd import ops  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
EMU_DIR = os.path.join(ROOT, "tests", "_emu")
EMU_PATH = os.path.join(EMU_DIR, "libpgt_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def build_emu_library():
    """The CPU test double: the SAME kernel sources compiled with g++ against tests/hipemu (fibers instead of lanes)."""
    if os.environ.get("PGT_EMU_LIB"):        # e.g. an AddressSanitizer build of the same sources (scripts/asan_audit.sh)
        return os.environ["PGT_EMU_LIB"]
    csrc = os.path.join(ROOT, "pytorch_geometric_temporal_amd", "csrc")
    srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
    deps = srcs + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(ROOT, "include", "pgt_hip.h"),
                                                          os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h")] + \
        glob.glob(os.path.join(ROOT, "tests", "hipemu", "*.h"))
    if os.path.exists(EMU_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_PATH) for d in deps):
        return EMU_PATH
    os.makedirs(EMU_DIR, exist_ok=True)
    tmp = f"{EMU_PATH}.{os.getpid()}.tmp"       # several pytest-xdist workers may build at once: each links its own file, the
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-x", "c++", "-DPGT_EMU",       # rename is atomic, nobody maps a half-written one
           "-I", os.path.join(ROOT, "tests", "hipemu"), "-I", os.path.join(ROOT, "include"), "-I", csrc] + srcs + \
          ["-o", tmp]
    subprocess.run(cmd, check=True)
    os.replace(tmp, EMU_PATH)
    return EMU_PATH


class Backend:
    def __init__(self, name, device):
        self.name, self.device = name, torch.device(device)

    def t(self, a, dtype=None):
        t = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a, dtype=dtype)
        return t.detach().clone().to(self.device)


_EMU_LIB = None


def get_emu_lib():
    """Built and mapped on first use only: a `-m gpu` process never requests it, so it maps libpgt_hip.so alone."""
    global _EMU_LIB
    if _EMU_LIB is None:
        _EMU_LIB = _lib.PgtLib(build_emu_library())
    return _EMU_LIB


@pytest.fixture(scope="session")
def emu_lib():
    return get_emu_lib()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """"emu": kernels on the CPU test double (runs everywhere).  "hip": the product library on cuda:0 (-m gpu)."""
    ops.GRAPH_CACHE.clear()
    if request.param == "emu":
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _lib._set_library_for_testing(get_emu_lib())
        yield Backend("emu", "cpu")
        _lib._set_library_for_testing(None)
    else:
        _lib._set_library_for_testing(None)
        if not torch.cuda.is_available():
            pytest.skip("gpu-marked case: no GPU on this host (run through gpurun with -m gpu)")
        lib = _lib.get_lib()   # raises loudly if libpgt_hip.so is missing
        assert lib.target == "gfx950"
        yield Backend("hip", "cuda:0")
    ops.GRAPH_CACHE.clear()


@pytest.fixture
def emu_backend(emu_lib):
    """The CPU test double only (property / fuzz tests: many small cases, no GPU leg)."""
    ops.GRAPH_CACHE.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _lib._set_library_for_testing(emu_lib)
    yield Backend("emu", "cpu")
    _lib._set_library_for_testing(None)
    ops.GRAPH_CACHE.clear()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {"in": {}, "param": {}, "out": {}, "meta": {}}
    for k in z.files:
        grp, key = k.split("/", 1)
        out[grp][key] = torch.from_numpy(z[k]) if grp != "meta" else z[k]
    return out


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def assert_close_with_nonfinite(a, b, atol, rtol, what=""):
    """allclose that also demands identical inf / nan placement (the reference's own mock graphs produce them)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    assert torch.equal(fa, fb), f"{what}: non-finite placement differs ({(~fa).sum()} vs {(~fb).sum()})"
    assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: nan placement differs"
    inf_mask = torch.isinf(a)
    assert torch.equal(a[inf_mask], b[inf_mask]), f"{what}: inf signs differ"
    d = (a[fa] - b[fa]).abs()
    tol = atol + rtol * b[fa].abs()
    assert bool((d <= tol).all()), f"{what}: max abs diff {float(d.max()) if d.numel() else 0:.3e} (atol {atol}, rtol {rtol})"
