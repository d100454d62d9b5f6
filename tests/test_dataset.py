"""Dataset loaders + the .pgtc cache format: value parity with the reference's own loader code
(run in place from /root/reference when it exists), the reference's shape pins (test/dataset_test.py:304-312,
test/index_test.py:93-115), the committed Chickenpox fixture, and the file format itself."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_import
from pytorch_geometric_temporal_amd.dataset import (ChickenpoxDatasetLoader, EnglandCovidDatasetLoader,
                                                    METRLADatasetLoader, PemsBayDatasetLoader, csr_by_destination,
                                                    dense_to_sparse_numpy, load_cache, save_cache)

needs_reference = pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not present")


# ------------------------------------------------------------------------------------------------ file format

def test_cache_round_trip_alignment_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"series": rng.standard_normal((7, 5, 2)).astype(np.float32), "edge_index": rng.integers(0, 5, (2, 11)),
              "edge_weight": rng.random(11).astype(np.float32), "empty": np.zeros((0, 3), np.float32),
              "scalarish": np.array([3], dtype=np.int32)}
    p = save_cache(str(tmp_path / "a.pgtc"), "toy", arrays, {"nodes": 5, "note": "x" * 37})
    c = load_cache(p)
    assert c.name == "toy" and c.meta["nodes"] == 5 and not c.dynamic
    raw = open(p, "rb").read()
    assert raw[:8] == b"PGTCACHE"
    hdr = json.loads(raw[16:16 + int(np.frombuffer(raw[8:16], np.uint64)[0])])
    for k, a in arrays.items():
        assert np.array_equal(c.arrays[k], a) and c.arrays[k].dtype == a.dtype
        assert hdr["arrays"][k]["offset"] % 64 == 0
    assert np.array_equal(c.series, arrays["series"]) and "series" in c
    t = c.to_torch()
    assert t["edge_index"].dtype == torch.int64 and torch.equal(t["series"], torch.from_numpy(arrays["series"]))
    with pytest.raises(AttributeError):
        c.nope
    bad = tmp_path / "b.pgtc"
    bad.write_bytes(b"NOTACACHE" + raw[9:])
    with pytest.raises(ValueError, match="not a PGTCACHE"):
        load_cache(str(bad))
    bad.write_bytes(raw[:-40])
    with pytest.raises(ValueError, match="past the end"):
        load_cache(str(bad))
    newer = json.dumps(dict(hdr, version=99)).encode()
    bad.write_bytes(raw[:8] + np.uint64(len(newer)).tobytes() + newer)
    with pytest.raises(ValueError, match="newer"):
        load_cache(str(bad))


def test_csr_by_destination_is_stable_and_matches_dense():
    rng = np.random.default_rng(1)
    n, e = 9, 40
    ei = rng.integers(0, n, (2, e))
    w = rng.random(e).astype(np.float32)
    rp, col, val = csr_by_destination(ei, w, n)
    assert rp.dtype == np.int32 and col.dtype == np.int32 and val.dtype == np.float32 and rp[-1] == e
    dense = np.zeros((n, n), np.float64)
    np.add.at(dense, (ei[1], ei[0]), w)
    back = np.zeros((n, n), np.float64)
    for r in range(n):
        np.add.at(back[r], col[rp[r]:rp[r + 1]], val[rp[r]:rp[r + 1]])
        src_in_order = ei[0][ei[1] == r]                  # slots keep the edge order inside a row
        assert np.array_equal(col[rp[r]:rp[r + 1]], src_in_order)
    assert np.allclose(dense, back)
    with pytest.raises(ValueError):
        csr_by_destination(np.array([[0], [n]]), None, n)
    A = np.array([[0, 2.0, 0], [0, 0, 0], [1.5, 0, 3.0]])
    ei2, v2 = dense_to_sparse_numpy(A)
    assert ei2.tolist() == [[0, 2, 2], [1, 0, 2]] and v2.tolist() == [2.0, 1.5, 3.0]


# ------------------------------------------------------------------------------------------------ packaged datasets

def test_chickenpox_shapes_fixture_and_index_batches():
    loader = ChickenpoxDatasetLoader()
    dataset = loader.get_dataset()
    g = load_golden("chickenpox_signal_head")             # frozen from the reference's StaticGraphTemporalSignal
    assert dataset.snapshot_count == int(g["out"]["snapshot_count"]) == 517
    for epoch in range(2):                                # test/dataset_test.py:304-312
        n = 0
        for snapshot in dataset:
            assert snapshot.edge_index.shape == (2, 102) and snapshot.edge_attr.shape == (102,)
            assert snapshot.x.shape == (20, 4) and snapshot.y.shape == (20,)
            n += 1
        assert n == 517
    for t in range(3):
        assert torch.equal(dataset[t].x, g["out"][f"x{t}"]) and torch.equal(dataset[t].y, g["out"][f"y{t}"])
    assert torch.equal(dataset[0].edge_index, g["out"]["edge_index"])
    assert torch.equal(dataset[0].edge_attr, g["out"]["edge_weight"])
    # index batching == snapshot iterator (test/index_test.py:93-115)
    with pytest.raises(ValueError):
        loader.get_index_dataset()
    tr, va, te, edges, edge_weights = ChickenpoxDatasetLoader(index=True).get_index_dataset(batch_size=1, shuffle=False)
    assert edges.shape == (2, 102) and edge_weights.shape == (102,) and edges.dtype == torch.int64
    assert (len(tr.dataset), len(va.dataset), len(te.dataset)) == (360, 51, 103)
    for snapshot, (x, y) in zip(dataset, tr):
        assert torch.equal(snapshot.x, torch.squeeze(x).permute(1, 0).float())
        assert torch.equal(snapshot.y, torch.squeeze(y).float()[0, ...])
        assert torch.equal(snapshot.edge_index, edges) and torch.equal(snapshot.edge_attr, edge_weights)
    # the packaged cache carries the graph as CSR by destination too
    c = loader._cache
    rp, col, val = csr_by_destination(c.edge_index, c.edge_weight, 20)
    assert np.array_equal(c.csr_rowptr, rp) and np.array_equal(c.csr_col, col) and np.array_equal(c.csr_val, val)


def test_england_covid_dynamic_signal():
    ds = EnglandCovidDatasetLoader().get_dataset(lags=8)
    assert ds.snapshot_count == 53
    sizes = set()
    for t, snap in enumerate(ds):
        assert snap.x.shape == (129, 8) and snap.y.shape == (129,)
        assert snap.edge_index.shape[0] == 2 and snap.edge_index.shape[1] == snap.edge_attr.shape[0]
        assert int(snap.edge_index.max()) < 129
        sizes.add(snap.edge_index.shape[1])
    assert len(sizes) > 10 and min(sizes) >= 836 and max(sizes) <= 2158
    z = np.stack([np.asarray(f) for f in ds.features])
    assert abs(float(z.mean())) < 0.2


@needs_reference
def test_vendored_datasets_equal_the_reference_loaders():
    root = os.path.join(ref_import.REFERENCE_ROOT, "dataset")
    cp = ref_import.load_dataset("chickenpox")
    ref = object.__new__(cp.ChickenpoxDatasetLoader)      # skip __init__: it downloads
    ref._dataset = json.load(open(os.path.join(root, "chickenpox.json")))
    ref.index = False
    for lags in (4, 7):
        a, b = ref.get_dataset(lags), ChickenpoxDatasetLoader().get_dataset(lags)
        assert np.array_equal(a.edge_index, b.edge_index) and np.array_equal(a.edge_weight, b.edge_weight)
        assert len(a.features) == len(b.features)
        for fa, fb, ta, tb in zip(a.features, b.features, a.targets, b.targets):
            assert np.array_equal(fa, fb) and np.array_equal(ta, tb)
    # the JSON itself is accepted too
    c = ChickenpoxDatasetLoader(path=os.path.join(root, "chickenpox.json")).get_dataset()
    assert np.array_equal(np.stack(c.features), np.stack(ChickenpoxDatasetLoader().get_dataset().features))
    ec = ref_import.load_dataset("encovid")
    ref = object.__new__(ec.EnglandCovidDatasetLoader)
    ref._dataset = json.load(open(os.path.join(root, "england_covid.json")))
    a, b = ref.get_dataset(8), EnglandCovidDatasetLoader().get_dataset(8)
    assert len(a.edge_indices) == len(b.edge_indices) == 53
    for t in range(53):
        assert np.array_equal(a.edge_indices[t], b.edge_indices[t])
        assert np.array_equal(a.edge_weights[t], b.edge_weights[t])
        assert np.array_equal(a.features[t], b.features[t]) and np.array_equal(a.targets[t], b.targets[t])


# ------------------------------------------------------------------------------------------------ sensor networks

def _write_sensor_files(d, prefix, n=9, steps=60, feats=2, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.random((n, n)).astype(np.float32)
    A[A < 0.6] = 0.0
    np.fill_diagonal(A, 1.0)
    X = (50 + 10 * rng.standard_normal((steps, n, feats))).astype(np.float64)     # [T, N, F] on disk
    np.save(os.path.join(d, prefix + "adj_mat.npy"), A)
    np.save(os.path.join(d, prefix + "node_values.npy"), X)
    return A, X


@pytest.mark.parametrize("cls,prefix,zipname,mod", [(METRLADatasetLoader, "", "METR-LA.zip", "metr_la"),
                                                    (PemsBayDatasetLoader, "pems_", "PEMS-BAY.zip", "pems_bay")])
def test_sensor_network_loaders(tmp_path, cls, prefix, zipname, mod):
    d = str(tmp_path)
    with pytest.raises(FileNotFoundError, match="never downloads"):
        cls(raw_data_dir=d)
    A, X = _write_sensor_files(d, prefix)
    ds = cls(raw_data_dir=d).get_dataset(num_timesteps_in=4, num_timesteps_out=3)
    assert ds.snapshot_count == 60 - 7 + 1
    s0 = ds[0]
    assert s0.x.shape == (9, 2, 4) and s0.y.shape == ((9, 3) if cls is METRLADatasetLoader else (9, 2, 3))
    ei, ew = dense_to_sparse_numpy(A)
    assert torch.equal(s0.edge_index, torch.from_numpy(ei)) and torch.equal(s0.edge_attr, torch.from_numpy(ew))
    # index batching: windows of the time-major z-scored series, split 70 / 10 / 20
    L = cls(raw_data_dir=d, index=True)
    tr, va, te, edges, w, means, stds = L.get_index_dataset(lags=4, batch_size=5, shuffle=False)
    assert (len(tr.dataset), len(va.dataset), len(te.dataset)) == (37, 5, 11)
    x, y = next(iter(tr))
    assert x.shape == (5, 4, 9, 2) and y.shape == (5, 4, 9, 2) and x.dtype == torch.float32
    Xn = X.transpose(1, 2, 0).astype(np.float32)
    assert np.allclose(means.numpy(), Xn.mean(axis=(0, 2)), rtol=1e-6)
    assert torch.equal(x[1, :, :, 0].t(), ds[1].x[:, 0, :])
    assert torch.equal(y[0], x.new_tensor(np.asarray(L._cache.series)[4:8]))
    # DDP sharding of the windows (metr_la.py:220-228)
    tr0 = L.get_index_dataset(lags=4, batch_size=5, world_size=2, ddp_rank=0)[0]
    tr1 = L.get_index_dataset(lags=4, batch_size=5, world_size=2, ddp_rank=1)[0]
    assert len(tr0.sampler) + len(tr1.sampler) >= 37 and set(tr0.sampler).isdisjoint(set(list(tr1.sampler)[:18]))
    # persisted cache: same arrays, and the .npy files are no longer needed
    L.write_cache()
    os.remove(os.path.join(d, prefix + "adj_mat.npy"))
    os.remove(os.path.join(d, prefix + "node_values.npy"))
    L2 = cls(raw_data_dir=d, index=True)
    for k in L._cache.arrays:
        assert np.array_equal(np.asarray(L._cache.arrays[k]), np.asarray(L2._cache.arrays[k])), k
    assert L2._cache.path is not None
    ds2 = cls(raw_data_dir=d).get_dataset(4, 3)
    assert torch.equal(ds2[3].x, ds[3].x) and torch.equal(ds2[3].y, ds[3].y)
    if ref_import.reference_available():
        A, X = _write_sensor_files(d, prefix)
        open(os.path.join(d, zipname), "wb").close()       # the reference only checks that the archive exists
        ref = getattr(ref_import.load_dataset(mod), cls.__name__)(raw_data_dir=d)
        a = ref.get_dataset(num_timesteps_in=4, num_timesteps_out=3)
        os.remove(os.path.join(d, L._CACHE))
        b = cls(raw_data_dir=d).get_dataset(4, 3)
        assert np.array_equal(a.edge_index, b.edge_index) and np.array_equal(a.edge_weight, b.edge_weight)
        for fa, fb, ta, tb in zip(a.features, b.features, a.targets, b.targets):
            assert np.array_equal(fa, fb) and np.array_equal(ta, tb)


def test_fuzz_cache_round_trip(tmp_path):
    """.pgtc files: any mix of dtypes and shapes (zero-sized arrays, scalars-as-1-element, odd byte counts between the
    64-byte aligned blocks) comes back bit for bit, memory-mapped read-only, with the metadata intact."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import HealthCheck, given, settings, strategies as hst
    from pytorch_geometric_temporal_amd.dataset.cache import load_cache, save_cache

    dtypes = [np.float32, np.float64, np.int64, np.int32, np.int16, np.uint8, np.bool_]

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(spec=hst.lists(hst.tuples(hst.sampled_from(range(len(dtypes))),
                                     hst.lists(hst.integers(0, 5), min_size=1, max_size=3)), min_size=1, max_size=6),
           seed=hst.integers(0, 1000))
    def run(spec, seed):
        rng = np.random.default_rng(seed)
        arrays = {}
        for i, (dt, shape) in enumerate(spec):
            a = (rng.random(shape) * 100).astype(dtypes[dt])
            arrays[f"a{i}"] = a if i % 2 == 0 else np.asfortranarray(a)          # non-contiguous inputs are normalised
        path = str(tmp_path / f"f{seed}.pgtc")
        meta = {"seed": seed, "names": sorted(arrays), "nested": {"x": [1, 2.5, "s"]}}
        save_cache(path, "fuzz", arrays, meta)
        c = load_cache(path)
        assert c.name == "fuzz" and c.meta == meta and sorted(c.arrays) == sorted(arrays)
        for k, a in arrays.items():
            b = c.arrays[k]
            assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(np.asarray(b), a)
            if a.size:
                assert not b.flags.writeable
                assert (b.ctypes.data - np.asarray(b).ctypes.data) == 0
                assert b.offset % 64 == 0                                        # 64-byte aligned blocks

    run()
    with pytest.raises(ValueError, match="not a PGTCACHE"):
        p = tmp_path / "junk.pgtc"
        p.write_bytes(b"0123456789abcdef0123")
        load_cache(str(p))


def test_pems_california_loader_pkl_h5_and_cache(tmp_path, monkeypatch):
    """PemsDatasetLoader (dataset/pems.py:14-179): pickled adjacency + pandas .h5 frame -> z-scored [T, N, 2] series
    (speed, time of day), index-batch loaders; value for value against the reference loader on the same files when the
    reference is mounted.  PyTables is not installed here, so `pandas.read_hdf` is served from a pickled frame of the
    same content (the loaders call nothing else on the .h5)."""
    import pickle
    import pandas as pd
    from pytorch_geometric_temporal_amd.dataset import PemsDatasetLoader
    d = str(tmp_path)
    with pytest.raises(FileNotFoundError, match="never downloads"):
        PemsDatasetLoader(raw_data_dir=d, index=True)
    rng = np.random.default_rng(0)
    n, steps = 7, 80
    A = rng.random((n, n)).astype(np.float32)
    A[A < 0.55] = 0.0
    np.fill_diagonal(A, 1.0)
    with open(os.path.join(d, "pems_cali_adj_mat.pkl"), "wb") as f:
        pickle.dump(([str(i) for i in range(n)], {str(i): i for i in range(n)}, A), f)
    frame = pd.DataFrame(55 + 9 * rng.standard_normal((steps, n)),
                         index=pd.date_range("2018-01-01 00:00", periods=steps, freq="5min"))
    frame.to_pickle(os.path.join(d, "pems_cali_speed.h5"))
    monkeypatch.setattr(pd, "read_hdf", lambda path, key=None, **kw: pd.read_pickle(path))
    L = PemsDatasetLoader(raw_data_dir=d, index=True)
    tr, va, te, edges, w, means, stds = L.get_index_dataset(lags=4, batch_size=6, shuffle=False)
    ei, ew = dense_to_sparse_numpy(A)
    assert torch.equal(edges, torch.from_numpy(ei)) and torch.equal(w, torch.from_numpy(ew))
    nwin = steps - 7
    assert (len(tr.dataset), len(va.dataset), len(te.dataset)) == (round(nwin * 0.7), nwin - round(nwin * 0.7) - round(nwin * 0.2),
                                                                  round(nwin * 0.2))
    x, y = next(iter(tr))
    assert x.shape == (6, 4, n, 2) and y.shape == (6, 4, n, 2) and x.dtype == torch.float64
    tod = (np.arange(steps) * 5 / 1440.0) % 1.0
    raw = np.stack([frame.values, np.tile(tod[:, None], (1, n))], axis=-1)
    assert np.allclose(means.numpy(), raw.mean(axis=(0, 1)), rtol=1e-6)
    assert np.allclose(x[2].numpy(), ((raw - raw.mean(axis=(0, 1))) / raw.std(axis=(0, 1)))[2:6], rtol=1e-12, atol=1e-12)
    if ref_import.reference_available():
        import sys
        import types
        if "dask" not in sys.modules:       # the reference's index_dataset.py imports dask.array at module level (unused here)
            monkeypatch.setitem(sys.modules, "dask", types.ModuleType("dask"))
            monkeypatch.setitem(sys.modules, "dask.array", types.ModuleType("dask.array"))
        ref = ref_import.load_dataset("pems").PemsDatasetLoader(raw_data_dir=d, index=True)   # files exist: no download
        rtr, rva, rte, redges, rw, rmeans, rstds = ref.get_index_dataset(lags=4, batch_size=6, shuffle=False)
        assert torch.equal(redges, edges) and torch.equal(rw, w)
        assert torch.equal(rmeans, means) and torch.equal(rstds, stds)
        for a, b in ((rtr, tr), (rva, va), (rte, te)):
            assert len(a.dataset) == len(b.dataset)
            for (xa, ya), (xb, yb) in zip(a, b):
                assert torch.equal(xa, xb) and torch.equal(ya, yb)
    # persisted cache: the raw files (and pandas / PyTables) are no longer needed
    L.write_cache()
    os.remove(os.path.join(d, "pems_cali_adj_mat.pkl"))
    os.remove(os.path.join(d, "pems_cali_speed.h5"))
    monkeypatch.setattr(pd, "read_hdf", lambda *a, **k: (_ for _ in ()).throw(AssertionError("cache must not read the .h5")))
    L2 = PemsDatasetLoader(raw_data_dir=d, index=True)
    x2, y2 = next(iter(L2.get_index_dataset(lags=4, batch_size=6)[0]))
    assert torch.equal(x2, x) and torch.equal(y2, y)


def test_out_of_scope_loaders_import_and_say_why_they_do_not_load():
    """PedalMe / Montevideo-bus (reference dataset/pedalme.py, montevideo_bus.py) are outside SURVEY §8: the names import (an
    import-swapped script does not die on its import line), the call explains."""
    from pytorch_geometric_temporal_amd.dataset import MontevideoBusDatasetLoader, PedalMeDatasetLoader
    for cls in (PedalMeDatasetLoader, MontevideoBusDatasetLoader):
        with pytest.raises(NotImplementedError, match="outside this package's scope"):
            cls()
