"""One binary on-disk format for a temporal graph dataset.

The reference reads three different raw formats at start-up — a vendored JSON (`dataset/chickenpox.py:32-44`,
`dataset/encovid.py:22-25`), `adj_mat.npy` / `node_values.npy` out of a downloaded zip (`dataset/metr_la.py:55-96`,
`dataset/pems_bay.py:61-96`) — and converts them on every run (JSON parse, dense -> sparse adjacency, z-scoring,
transposes).  A `.pgtc` file holds the result of that conversion once, laid out for the device:

    bytes 0..7      magic  b"PGTCACHE"
    bytes 8..15     little-endian uint64: length L of the JSON header
    bytes 16..16+L  JSON header: {"version", "name", "meta": {...}, "arrays": {name: {"dtype", "shape", "offset"}}}
    then            the arrays, each starting at a 64-byte aligned offset (relative to the start of the file)

Arrays of a static-graph dataset: `series` [T, N, F] float32 (time-major: index batching slices it directly, and it
uploads to HBM in one copy), `edge_index` [2, E] int64 and `edge_weight` [E] float32 in the reference's order, and the
same graph as CSR by destination row (`csr_rowptr` int32 [N+1], `csr_col` int32 [E], `csr_val` float32 [E]; slots keep
the edge order inside a row, the order `pgt_*_prep` produces on the device).  Dynamic-graph datasets store the per-step
edge lists concatenated (`edge_index`, `edge_weight`) plus `edge_offset` int64 [steps + 1].  Optional: `means`,
`stds` (z-score statistics), anything else the writer passes.

`load_cache(path)` memory-maps the file: nothing is parsed or copied until an array is touched.
"""
import json
import os

import numpy as np

MAGIC = b"PGTCACHE"
VERSION = 1
_ALIGN = 64


def csr_by_destination(edge_index, edge_weight, num_nodes):
    """(rowptr int32 [N+1], col int32 [E], val float32 [E]): row = destination node edge_index[1], col = source node,
    slots inside a row in edge order (stable sort) — the layout of `struct pgt_csr` (include/pgt_hip.h)."""
    ei = np.asarray(edge_index)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, E], got {ei.shape}")
    src, dst = ei[0].astype(np.int64), ei[1].astype(np.int64)
    if src.size and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= num_nodes):
        raise ValueError("edge endpoint out of range")
    order = np.argsort(dst, kind="stable")
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    w = np.ones(src.size, dtype=np.float32) if edge_weight is None else np.asarray(edge_weight, dtype=np.float32)
    return rowptr.astype(np.int32), src[order].astype(np.int32), w[order]


def save_cache(path, name, arrays, meta=None):
    """Write `arrays` (name -> numpy array) and the JSON-serialisable `meta` dict as one .pgtc file (atomically)."""
    header = {"version": VERSION, "name": name, "meta": meta or {}, "arrays": {}}
    mats = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    # two passes: the header length moves the first offset, the offsets are part of the header
    offsets, hdr = {}, b""
    for _ in range(3):
        pos = 16 + len(hdr)
        for k, a in mats.items():
            pos = (pos + _ALIGN - 1) // _ALIGN * _ALIGN
            offsets[k] = pos
            pos += a.nbytes
        header["arrays"] = {k: {"dtype": a.dtype.str, "shape": list(a.shape), "offset": offsets[k]} for k, a in mats.items()}
        new = json.dumps(header, sort_keys=True).encode()
        new += b" " * (-len(new) % 8)
        if len(new) == len(hdr):
            hdr = new
            break
        hdr = new
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(hdr)).tobytes())
        f.write(hdr)
        for k, a in mats.items():
            f.write(b"\0" * (offsets[k] - f.tell()))
            f.write(a.tobytes())
    os.replace(tmp, path)
    return path


class TemporalGraphCache:
    """A loaded .pgtc file: `.name`, `.meta`, `.arrays` (numpy memmaps, read-only), attribute access to the arrays."""

    def __init__(self, name, meta, arrays, path):
        self.name, self.meta, self.arrays, self.path = name, meta, arrays, path

    def __getattr__(self, key):
        arrays = self.__dict__.get("arrays", {})
        if key in arrays:
            return arrays[key]
        raise AttributeError(key)

    def __contains__(self, key):
        return key in self.arrays

    @property
    def dynamic(self):
        return "edge_offset" in self.arrays

    def step_edges(self, t):
        """(edge_index [2, E_t], edge_weight [E_t]) of step t of a dynamic-graph dataset."""
        a, b = int(self.arrays["edge_offset"][t]), int(self.arrays["edge_offset"][t + 1])
        return self.arrays["edge_index"][:, a:b], self.arrays["edge_weight"][a:b]

    def to_torch(self, device=None, keys=None):
        """dict of torch tensors (one host -> device copy per array; int32 CSR arrays stay int32)."""
        import torch
        out = {}
        for k in (keys or self.arrays):
            t = torch.from_numpy(np.array(self.arrays[k]))   # own the memory: the map may outlive nothing
            out[k] = t.to(device) if device is not None else t
        return out


def load_cache(path):
    """Memory-map a .pgtc file.  Raises ValueError for a foreign / truncated / newer-version file."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise ValueError(f"{path}: not a PGTCACHE file")
        hlen = int(np.frombuffer(head[8:16], dtype=np.uint64)[0])
        if 16 + hlen > size:
            raise ValueError(f"{path}: truncated header")
        header = json.loads(f.read(hlen).decode())
    if header.get("version", 0) > VERSION:
        raise ValueError(f"{path}: format version {header.get('version')} is newer than this reader ({VERSION})")
    arrays = {}
    for k, d in header["arrays"].items():
        dt, shape, off = np.dtype(d["dtype"]), tuple(d["shape"]), int(d["offset"])
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        if off + nbytes > size:
            raise ValueError(f"{path}: array '{k}' runs past the end of the file")
        arrays[k] = np.memmap(path, dtype=dt, mode="r", offset=off, shape=shape) if nbytes else np.zeros(shape, dt)
    return TemporalGraphCache(header["name"], header["meta"], arrays, path)
