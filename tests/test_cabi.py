"""The drop-in boundary: include/pgt_hip.h <-> ctypes prototypes <-> symbols exported by the built libraries."""
import os
import re
import subprocess

import pytest

from conftest import ROOT, build_emu_library
from pytorch_geometric_temporal_amd import _build, _lib

HEADER = os.path.join(ROOT, "include", "pgt_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(pgt_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def exported(path):
    txt = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {l.split()[-1] for l in txt.splitlines() if " T " in l and l.split()[-1].startswith("pgt_")}


def test_header_and_ctypes_prototypes_agree():
    hdr = header_functions()
    assert set(hdr) == set(_lib.PROTOTYPES), set(hdr) ^ set(_lib.PROTOTYPES)
    for name, nargs in hdr.items():
        assert len(_lib.PROTOTYPES[name][1]) == nargs, name


def test_hip_library_builds_for_gfx950_and_exports_every_declared_symbol():
    path = _build.build_hip_library()          # hipcc cross-compiles without a GPU
    assert exported(path) == set(header_functions())
    lib = _lib.PgtLib(path)                    # loads (no compute) and reports the right target / ABI
    assert lib.target == "gfx950"
    blob = open(path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob      # the fat binary carries gfx950 code objects


def test_emu_test_double_exports_the_same_abi():
    assert exported(build_emu_library()) == set(header_functions())


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.PgtLibraryMissing, match="no CPU fallback"):
        _lib.PgtLib(str(tmp_path / "libpgt_hip.so"))


def test_only_emu_library_can_be_injected():
    class Fake:
        target = "gfx950"
    with pytest.raises(_lib.PgtError):
        _lib._set_library_for_testing(Fake())
