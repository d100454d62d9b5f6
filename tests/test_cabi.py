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
    for m in re.finditer(r"\b(?:int|int64_t|size_t|const char\*)\s+(pgt_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
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


def test_integration_notes_name_the_current_abi_and_every_entry_point():
    """INTEGRATION.md is what a maintainer of the reference binds from: its ABI number is the header's, and every exported symbol
    is named in its entry-point table."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    abi = int(re.search(r"#define\s+PGT_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert abi == _lib.EXPECTED_ABI
    assert {int(n) for n in re.findall(r"PGT_ABI_VERSION (\d+)", doc)} == {abi}
    assert {int(n) for n in re.findall(r"pgt_abi_version\(\) == (\d+)", doc)} == {abi}
    missing = sorted(n for n in header_functions() if f"`{n}`" not in doc)
    assert missing == [], missing


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.PgtLibraryMissing, match="no CPU fallback"):
        _lib.PgtLib(str(tmp_path / "libpgt_hip.so"))


def test_only_emu_library_can_be_injected():
    class Fake:
        target = "gfx950"
    with pytest.raises(_lib.PgtError):
        _lib._set_library_for_testing(Fake())


def test_product_code_never_touches_the_oracle_or_the_test_double():
    """The oracle (and the CPU test double of the kernels) are test infrastructure: nothing in the package may import,
    load or shell out to them; bench.py only does so inside its cpu_baseline leg, __graft_entry__ only in smoke()."""
    pkg = os.path.join(ROOT, "pytorch_geometric_temporal_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            if re.search(r"^\s*(from|import)\s+(oracle|tests|conftest)\b", src, flags=re.M) \
                    or ("libpgt_emu" in src and f != "_lib.py"):      # _lib.py names it in the test hook's docstring
                offenders.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert offenders == []
    # bench.py / bench_tgcn.py: the oracle is imported only INSIDE the CPU-baseline functions (`def cpu_*`: the baseline proper, the
    # thread-sweep point a child process runs, config 4's CPU leg) — never at module level, never in a timed GPU path
    for fname in ("bench.py", "bench_tgcn.py"):
        bench = open(os.path.join(ROOT, fname)).read()
        for m in re.finditer(r"^[ \t]*(?:from|import)\s+oracle\b.*$", bench, flags=re.M):
            head = bench.rfind("\ndef ", 0, m.start())
            assert head >= 0, f"{fname}: module-level oracle import: {m.group(0)!r}"
            name = re.match(r"\ndef\s+(\w+)", bench[head:]).group(1)
            assert name.startswith("cpu_"), f"{fname}: oracle imported in {name}()"
            assert m.group(0).startswith((" ", "\t")), f"{fname}: oracle import at module level"


def test_tune_switches_from_the_environment_are_parsed(monkeypatch):
    calls = []

    class Dummy:
        target = "gfx950"

        def tune(self, k, v):
            calls.append((k, v))

    monkeypatch.setenv("PGT_TUNE", "gemm_dbp=1, slab_pairs=2")
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "_preload_hip_runtime", lambda: None)
    monkeypatch.setattr(_lib, "_check_not_stale", lambda: None)
    monkeypatch.setattr(_lib, "PgtLib", lambda path: Dummy())
    try:
        _lib.get_lib()
        assert calls == [("gemm_dbp", 1), ("slab_pairs", 2)]
    finally:
        _lib._LIB = None


def test_bad_arguments_return_error_codes_not_crashes():
    """The C ABI validates before it launches: null pointers, negative sizes, undersized scratch and unknown switches
    come back as PGT_ERR_* with a message in pgt_last_error(); zero-sized problems are PGT_OK no-ops (no launch)."""
    import ctypes
    import torch
    lib = _lib.PgtLib(build_emu_library())                  # same host-side entry code as the product library
    null = ctypes.c_void_p(0)
    x = torch.zeros(64)
    p = ctypes.c_void_p(x.data_ptr())
    bad = [
        ("pgt_gemm_f32", (p, 4, 0, 1, 4, p, 4, 1, null, 4, 0, 4, null, 4, 4, 0, null)),          # null output
        ("pgt_gemm_f32", (p, 4, 0, 1, 4, p, 4, 1, p, 4, 0, 4, null, -1, 4, 0, null)),            # negative M
        ("pgt_gemm_f32", (p, 4, 0, 1, 4, p, 4, 1, p, 4, 0, 0, null, 4, 4, 0, null)),             # c_seg_n = 0
        ("pgt_gemm_tn_acc_f32", (p, 4, 0, 1, 4, null, 4, p, 4, null, 4, 4, null)),               # null gradient
        ("pgt_spmm_csr_f32", (null, p, p, 4, p, 4, p, 4, null, 0, 1.0, 0.0, 4, null)),           # null rowptr
        ("pgt_spmm_csr_f32", (p, p, p, -2, p, 4, p, 4, null, 0, 1.0, 0.0, 4, null)),             # negative rows
        ("pgt_copy2d_f32", (null, 4, p, 4, 4, 4, null)),                                         # null destination
    ]
    for name, args in bad:
        rc = getattr(lib, "_" + name)(*args)
        assert rc < 0, (name, rc)
        assert lib.last_error(), name
        with pytest.raises(_lib.PgtError, match=name):
            lib.call(name, *args)
    with pytest.raises(_lib.PgtError, match="unknown key"):
        lib.tune("no_such_switch", 1)
    # every switch the header documents is accepted (a key dropped from pgt_tune would silently disable an A/B script)
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pgt_hip.h")).read()
    doc = header[header.index("Schedule switches for A/B measurements"):header.index("int pgt_tune(")]
    keys = set(re.findall(r'"([a-z][a-z0-9_]+)"', doc))
    assert {"gemm_bx", "gemm_bx_sym", "gemm_db", "spmm_ellw", "slab_pairs"} <= keys
    defaults = {"gemm_small_fill": 256, "gemm_small_tiles": 0, "spmm_tile_rows": 32, "spmm_unroll": 8, "spmm_ellw_rows": 0,
                "spmm_ellw_cus": 0, "spmm_ellw_cfg": 0, "slab_pairs": 2, "slab_wpc": 0, "slab_threads": 0, "slab_gu": 2}   # every other switch defaults to 1
    defaults["tgcn_wgs"] = 0
    for k in sorted(keys - {"tgcn_probe"}):
        lib.tune(k, defaults.get(k, 1))
    # ... except the one that selects kernels which compute wrong results on purpose: compiled only into lab builds
    with pytest.raises(_lib.PgtError, match="PGT_LAB_PROBES"):
        lib.tune("tgcn_probe", 1)
    assert lib.prep_workspace_bytes(2, 2) > 8
    ok = [
        ("pgt_gemm_f32", (p, 4, 0, 1, 4, p, 4, 1, p, 4, 0, 4, null, 0, 4, 0, null)),             # M = 0
        ("pgt_gemm_tn_acc_f32", (p, 4, 0, 1, 4, p, 4, p, 4, null, 0, 4, null)),                  # M = 0
        ("pgt_copy2d_f32", (p, 4, p, 4, 0, 4, null)),
    ]
    for name, args in ok:
        assert getattr(lib, "_" + name)(*args) == 0, name
