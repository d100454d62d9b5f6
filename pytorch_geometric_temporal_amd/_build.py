"""Builds libpgt_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a plain
C-ABI shared object (include/pgt_hip.h).  hipcc cross-compiles without a GPU, so this also runs on CPU-only hosts."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpgt_hip.so")
INCLUDE = os.path.join(ROOT, "include")

SOURCES = ["pgt_core.hip", "spmm.hip", "dconv_slab.hip", "gemm.hip", "gemm_bx.hip", "elementwise.hip",
           "graph_prep.hip", "attention.hip", "evolve.hip", "small_cell.hip", "small_gcn.hip", "tconv.hip", "tgcn.hip", "seq_small.hip", "tgcn_cell.hip", "tile_order.hip", "readout.hip", "seq64.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def source_digest():
    """sha256 over the kernel sources and headers: `lib/build_id.txt` records the digest the library was built from, and
    the loader refuses a library whose digest differs from the sources next to it (a stale .so silently benchmarks
    old kernels)."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "pgt_common.h"),
                                                       os.path.join(INCLUDE, "pgt_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


BUILD_ID_PATH = os.path.join(LIB_DIR, "build_id.txt")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link pytorch_geometric_temporal_amd/lib/libpgt_hip.so."""
    hipcc = find_hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, "pgt_common.h"), os.path.join(INCLUDE, "pgt_hip.h")]
    objs, relink, jobs = [], force, []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE, "-I", CSRC, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append(cmd)
            relink = True
        objs.append(o)
    if jobs:
        # one hipcc per translation unit, side by side (the unrolled GEMM kernels take minutes each)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(lambda c: subprocess.run(c, check=True), jobs))
    if relink or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    with open(BUILD_ID_PATH, "w") as fh:
        fh.write(source_digest() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip_library(force="--force" in sys.argv, verbose=True))
