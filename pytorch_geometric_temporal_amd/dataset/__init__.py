"""Dataset loaders of the reference (torch_geometric_temporal/dataset) for a machine WITHOUT a network, on top of one
binary cache format (dataset/cache.py).

Same class names, constructor / method signatures and return values as the reference loaders for the datasets the hot
path's examples use:

    ChickenpoxDatasetLoader      dataset/chickenpox.py:10-128      static graph, vendored JSON
    EnglandCovidDatasetLoader    dataset/encovid.py:8-75           dynamic graph, vendored JSON
    METRLADatasetLoader          dataset/metr_la.py:15-262         adj_mat.npy + node_values.npy
    PemsBayDatasetLoader         dataset/pems_bay.py:14-250        pems_adj_mat.npy + pems_node_values.npy

Differences, all forced by "no network" and all loud: nothing is downloaded.  The two vendored datasets ship inside the
package as `.pgtc` caches (generated from the reference's JSON files by scripts/make_dataset_cache.py), so
`ChickenpoxDatasetLoader()` works out of the box; `path=` accepts the original JSON or another cache.  The sensor
networks read `raw_data_dir` for the reference's `.npy` files (or a `.pgtc` written by `write_cache()`, which skips the
dense -> sparse conversion, the transposes and the z-scoring on later runs) and raise FileNotFoundError otherwise.
"""
import json
import os
from typing import Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from ..signal import DynamicGraphTemporalSignal, IndexDataset, StaticGraphTemporalSignal
from .cache import TemporalGraphCache, csr_by_destination, load_cache, save_cache

__all__ = ["ChickenpoxDatasetLoader", "EnglandCovidDatasetLoader",
           "METRLADatasetLoader", "PemsBayDatasetLoader", "PemsDatasetLoader",
           "PedalMeDatasetLoader", "MontevideoBusDatasetLoader",
           "TemporalGraphCache", "load_cache", "save_cache", "csr_by_destination", "dense_to_sparse_numpy"]

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def dense_to_sparse_numpy(A):
    """(edge_index int64 [2, E], values) of a dense adjacency in row-major order of its non-zeros — what the
    reference obtains from PyG's `dense_to_sparse` (metr_la.py:91-96)."""
    A = np.asarray(A)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError(f"adjacency must be square, got {A.shape}")
    r, c = np.nonzero(A)
    return np.stack([r, c]).astype(np.int64), A[r, c]


def _split_indices(num_steps, lags, ratio):
    """Window start indices and their train / val / test split (chickenpox.py:112-121, metr_la.py:226-235)."""
    x_i = np.arange(num_steps - (2 * lags - 1))
    n = x_i.shape[0]
    n_train, n_test = round(n * ratio[0]), round(n * ratio[2])
    n_val = n - n_train - n_test
    return x_i[:n_train], x_i[n_train:n_train + n_val], x_i[-n_test:]


def _loaders(parts, data, lags, batch_size, shuffle, gpu, lazy, world_size=-1, ddp_rank=-1):
    sets = [IndexDataset(p, data, lags, gpu=gpu, lazy=lazy) for p in parts]
    if ddp_rank != -1:
        return [DataLoader(s, batch_size=batch_size,
                           sampler=DistributedSampler(s, num_replicas=world_size, rank=ddp_rank, shuffle=shuffle))
                for s in sets]
    return [DataLoader(s, batch_size=batch_size, shuffle=shuffle) for s in sets]


def _read(path, default_cache, json_reader):
    """A TemporalGraphCache from `path` (.pgtc or the reference's JSON) or from the packaged cache."""
    if path is None:
        path = os.path.join(_DATA_DIR, default_cache)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path} is missing from the installation; pass path= to the reference's JSON file "
                                    "(this package never downloads)")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"PGTCACHE":
        return load_cache(path)
    with open(path, "r") as f:
        return json_reader(json.load(f))


# ------------------------------------------------------------------------------------------------ Chickenpox

def _chickenpox_from_json(d):
    fx = np.array(d["FX"])
    edges = np.array(d["edges"]).T
    arrays = {"series": fx[:, :, None], "edge_index": edges.astype(np.int64),
              "edge_weight": np.ones(edges.shape[1], dtype=np.float32)}
    rp, col, val = csr_by_destination(edges, arrays["edge_weight"], fx.shape[1])
    arrays.update(csr_rowptr=rp, csr_col=col, csr_val=val)
    return TemporalGraphCache("chickenpox", {"nodes": int(fx.shape[1]), "steps": int(fx.shape[0])}, arrays, None)


class ChickenpoxDatasetLoader(object):
    """County-level weekly chickenpox cases in Hungary, 2004-2014 (reference: dataset/chickenpox.py:10-128): 20 nodes,
    102 directed edges (self-loops included), 521 weeks.  `index=True` enables `get_index_dataset`."""

    def __init__(self, index=False, path=None):
        self.index = index
        self._cache = _read(path, "chickenpox.pgtc", _chickenpox_from_json)
        if index:
            self.IndexDataset = IndexDataset

    def get_dataset(self, lags: int = 4) -> StaticGraphTemporalSignal:
        """Snapshot iterator: features [N, lags] = the `lags` previous weeks, target [N] = the next week."""
        self.lags = lags
        fx = np.array(self._cache.series)[:, :, 0]          # own, writable memory (the cache is a read-only map)
        self._edges = np.array(self._cache.edge_index)
        self._edge_weights = np.ones(self._edges.shape[1])
        n = fx.shape[0] - lags
        self.features = [fx[i:i + lags, :].T for i in range(n)]
        self.targets = [fx[i + lags, :].T for i in range(n)]
        return StaticGraphTemporalSignal(self._edges, self._edge_weights, self.features, self.targets)

    def get_index_dataset(self, lags=4, batch_size=4, shuffle=False, allGPU=-1, ratio=(0.7, 0.1, 0.2),
                          dask_batching=False):
        """(train, val, test DataLoaders over window start indices, edges [2, E] int64, edge_weights [E] float32)."""
        if not self.index:
            raise ValueError("get_index_dataset requires 'index=True' in the constructor.")
        data = np.array(self._cache.series)                          # [T, N, 1]
        edges = torch.from_numpy(np.array(self._cache.edge_index, dtype=np.int64))
        edge_weights = torch.ones(edges.shape[1], dtype=torch.float)
        steps = data.shape[0]
        if allGPU != -1:
            data = torch.tensor(data, dtype=torch.float).to(f"cuda:{allGPU}")
        parts = _split_indices(steps, lags, ratio)
        tr, va, te = _loaders(parts, data, lags, batch_size, shuffle, gpu=allGPU != -1, lazy=dask_batching)
        return tr, va, te, edges, edge_weights


# ------------------------------------------------------------------------------------------------ England Covid

def _covid_from_json(d):
    steps = int(d["time_periods"])
    ei = [np.array(d["edge_mapping"]["edge_index"][str(t)], dtype=np.int64).reshape(-1, 2).T for t in range(steps)]
    ew = [np.array(d["edge_mapping"]["edge_weight"][str(t)], dtype=np.float64) for t in range(steps)]
    off = np.zeros(steps + 1, dtype=np.int64)
    off[1:] = np.cumsum([e.shape[1] for e in ei])
    y = np.array(d["y"])
    arrays = {"series": y[:, :, None], "edge_index": np.concatenate(ei, axis=1), "edge_weight": np.concatenate(ew),
              "edge_offset": off}
    return TemporalGraphCache("england_covid", {"nodes": int(y.shape[1]), "steps": steps}, arrays, None)


class EnglandCovidDatasetLoader(object):
    """Daily COVID-19 cases in the NUTS3 regions of England with the day's mobility graph (reference:
    dataset/encovid.py:8-75): 129 nodes, 61 days, a different directed weighted graph every day."""

    def __init__(self, path=None):
        self._cache = _read(path, "england_covid.pgtc", _covid_from_json)

    def get_dataset(self, lags: int = 8) -> DynamicGraphTemporalSignal:
        self.lags = lags
        steps = int(self._cache.meta["steps"])
        y = np.array(self._cache.series)[:, :, 0]
        z = (y - np.mean(y, axis=0)) / (np.std(y, axis=0) + 10 ** -10)
        n = steps - lags
        self._edges, self._edge_weights = [], []
        for t in range(n):
            ei, ew = self._cache.step_edges(t)
            self._edges.append(np.array(ei))
            self._edge_weights.append(np.array(ew))
        self.features = [z[i:i + lags, :].T for i in range(n)]
        self.targets = [z[i + lags, :].T for i in range(n)]
        return DynamicGraphTemporalSignal(self._edges, self._edge_weights, self.features, self.targets)


# ------------------------------------------------------------------------------------------------ METR-LA / PeMS-BAY

class _SensorNetworkLoader(object):
    """Shared body of the two DCRNN-paper traffic datasets: a dense weighted adjacency [N, N] and node values
    [T, N, F] on disk; z-scored per feature; windows of `num_timesteps_in` -> `num_timesteps_out`."""

    _ADJ, _VALUES, _CACHE, _NAME = "", "", "", ""
    _TARGET_FEATURE_0_ONLY = False

    def __init__(self, raw_data_dir=os.path.join(os.getcwd(), "data"), index: bool = False):
        self.index = index
        self.raw_data_dir = raw_data_dir
        self._cache = None
        self._load()
        if index:
            self.IndexDataset = IndexDataset
        if not index:
            # [T, N, F] -> the reference's X [N, F, T] (metr_la.py:76-87) and dense A
            series = np.array(self._cache.series)
            self.X = torch.from_numpy(np.ascontiguousarray(series.transpose(1, 2, 0)))
            self.A = None

    # -- raw files or cache -> TemporalGraphCache with z-scored time-major series
    def _load(self):
        cpath = os.path.join(self.raw_data_dir, self._CACHE)
        if os.path.isfile(cpath):
            self._cache = load_cache(cpath)
            return
        apath, vpath = os.path.join(self.raw_data_dir, self._ADJ), os.path.join(self.raw_data_dir, self._VALUES)
        if not (os.path.isfile(apath) and os.path.isfile(vpath)):
            raise FileNotFoundError(
                f"{self._NAME}: neither {cpath} nor {apath} + {vpath} exist.  This package never downloads: unpack the "
                f"reference's archive into raw_data_dir ({self.raw_data_dir}) yourself.")
        A = np.load(apath)
        X = np.load(vpath).transpose((1, 2, 0)).astype(np.float32)            # [N, F, T]
        means = np.mean(X, axis=(0, 2))
        X = X - means.reshape(1, -1, 1)
        stds = np.std(X, axis=(0, 2))
        X = X / stds.reshape(1, -1, 1)
        ei, ew = dense_to_sparse_numpy(A)
        rp, col, val = csr_by_destination(ei, ew, A.shape[0])
        arrays = {"series": np.ascontiguousarray(X.transpose(2, 0, 1)), "edge_index": ei, "edge_weight": ew,
                  "csr_rowptr": rp, "csr_col": col, "csr_val": val, "means": means, "stds": stds}
        self._cache = TemporalGraphCache(self._NAME, {"nodes": int(A.shape[0]), "steps": int(X.shape[2])}, arrays, None)

    def write_cache(self, path=None):
        """Persist the converted dataset as `raw_data_dir/<name>.pgtc`; later constructions map it instead of
        re-reading and re-normalising the .npy files."""
        path = path or os.path.join(self.raw_data_dir, self._CACHE)
        return save_cache(path, self._cache.name, {k: np.asarray(v) for k, v in self._cache.arrays.items()},
                          self._cache.meta)

    def _get_edges_and_weights(self):
        self.edges = np.array(self._cache.edge_index)
        self.edge_weights = np.array(self._cache.edge_weight)

    def _generate_task(self, num_timesteps_in: int = 12, num_timesteps_out: int = 12):
        total = num_timesteps_in + num_timesteps_out
        features, target = [], []
        for i in range(self.X.shape[2] - total + 1):
            features.append(self.X[:, :, i:i + num_timesteps_in].numpy())
            if self._TARGET_FEATURE_0_ONLY:
                target.append(self.X[:, 0, i + num_timesteps_in:i + total].numpy())
            else:
                target.append(self.X[:, :, i + num_timesteps_in:i + total].numpy())
        self.features, self.targets = features, target

    def get_dataset(self, num_timesteps_in: int = 12, num_timesteps_out: int = 12) -> StaticGraphTemporalSignal:
        """features [N, F, num_timesteps_in] -> target [N, num_timesteps_out] (METR-LA: feature 0) or
        [N, F, num_timesteps_out] (PeMS-BAY)."""
        if self.index:
            raise ValueError("get_dataset requires 'index=False' in the constructor.")
        self._get_edges_and_weights()
        self._generate_task(num_timesteps_in, num_timesteps_out)
        return StaticGraphTemporalSignal(self.edges, self.edge_weights, self.features, self.targets)

    def get_index_dataset(self, lags: int = 12, batch_size: int = 64, shuffle: bool = False, allGPU: int = -1,
                          ratio: Tuple[float, float, float] = (0.7, 0.1, 0.2), world_size: int = -1,
                          ddp_rank: int = -1, dask_batching: bool = False):
        """(train, val, test DataLoaders, edges [2, E], edge_weights [E], means [F], stds [F]); the loaders yield
        (x [B, lags, N, F], y [B, lags, N, F]) windows of the time-major z-scored series (device-resident when
        `allGPU` names a GPU: "GPU-index-batching", metr_la.py:180-190)."""
        if not self.index:
            raise ValueError("get_index_dataset requires 'index=True' in the constructor.")
        edges = torch.from_numpy(np.array(self._cache.edge_index, dtype=np.int64))
        edge_weights = torch.from_numpy(np.array(self._cache.edge_weight))
        data = np.array(self._cache.series)                            # [T, N, F], already z-scored
        means = torch.tensor(np.asarray(self._cache.means), dtype=torch.float)
        stds = torch.tensor(np.asarray(self._cache.stds), dtype=torch.float)
        steps = data.shape[0]
        if allGPU != -1:
            data = torch.from_numpy(data).to(f"cuda:{allGPU}")
            means, stds = means.to(data.device), stds.to(data.device)
        parts = _split_indices(steps, lags, ratio)
        tr, va, te = _loaders(parts, data, lags, batch_size, shuffle, gpu=allGPU != -1, lazy=dask_batching,
                              world_size=world_size, ddp_rank=ddp_rank)
        return tr, va, te, edges, edge_weights, means, stds


class METRLADatasetLoader(_SensorNetworkLoader):
    """Los Angeles loop-detector speeds, 207 sensors, 5-minute steps, March-June 2012 (reference:
    dataset/metr_la.py:15-262).  Files: `adj_mat.npy`, `node_values.npy` (or `metr_la.pgtc`) in `raw_data_dir`."""
    _ADJ, _VALUES, _CACHE, _NAME = "adj_mat.npy", "node_values.npy", "metr_la.pgtc", "METR-LA"
    _TARGET_FEATURE_0_ONLY = True


class PemsBayDatasetLoader(_SensorNetworkLoader):
    """Bay Area loop-detector speeds, 325 sensors (reference: dataset/pems_bay.py:14-250).  Files:
    `pems_adj_mat.npy`, `pems_node_values.npy` (or `pems_bay.pgtc`) in `raw_data_dir`."""
    _ADJ, _VALUES, _CACHE, _NAME = "pems_adj_mat.npy", "pems_node_values.npy", "pems_bay.pgtc", "PEMS-BAY"
    _TARGET_FEATURE_0_ONLY = False


class PemsDatasetLoader(object):
    """The California-wide PeMS speed dataset of the DCRNN-partitioning paper (reference: dataset/pems.py:14-179):
    11 160 sensors; `pems_cali_adj_mat.pkl` (a pickled (ids, id->index, dense adjacency) triple) and
    `pems_cali_speed.h5` (a pandas frame [T, N] with a DatetimeIndex) in `raw_data_dir`.  Index batching only, as in
    the reference.  Channel 0 = speed, channel 1 = time of day; z-scored over (time, node) per channel.

    Never downloads.  The first construction from the raw files needs pandas + PyTables for the .h5 (as the reference
    does); `write_cache()` then persists `pems_cali.pgtc` (time-major series, edge list, CSR by destination,
    statistics), which later constructions memory-map without either dependency."""

    _ADJ, _VALUES, _CACHE, _NAME = "pems_cali_adj_mat.pkl", "pems_cali_speed.h5", "pems_cali.pgtc", "PEMS-CALI"

    def __init__(self, raw_data_dir=os.path.join(os.getcwd(), "data"), index=False):
        self.index = index
        self.raw_data_dir = raw_data_dir
        self.IndexDataset = IndexDataset
        self._cache = None
        self._load()

    def _load(self):
        cpath = os.path.join(self.raw_data_dir, self._CACHE)
        if os.path.isfile(cpath):
            self._cache = load_cache(cpath)
            return
        apath, vpath = os.path.join(self.raw_data_dir, self._ADJ), os.path.join(self.raw_data_dir, self._VALUES)
        if not (os.path.isfile(apath) and os.path.isfile(vpath)):
            raise FileNotFoundError(
                f"{self._NAME}: neither {cpath} nor {apath} + {vpath} exist.  This package never downloads: put the "
                f"reference's two files into raw_data_dir ({self.raw_data_dir}) yourself.")
        import pickle
        import pandas as pd
        with open(apath, "rb") as f:
            _, _, adj_mx = pickle.load(f)                                       # pems.py:103-104
        df = pd.read_hdf(vpath, "df")                                           # pems.py:108
        values = df.values
        num_nodes = values.shape[1]
        time_ind = (df.index.values - df.index.values.astype("datetime64[D]")) / np.timedelta64(1, "D")
        time_in_day = np.tile(time_ind, [1, num_nodes, 1]).transpose((2, 1, 0))
        data = np.concatenate([np.expand_dims(values, axis=-1), time_in_day], axis=-1)     # [T, N, 2], pems.py:119-139
        means = np.mean(data, axis=(0, 1))
        stds = np.std(data, axis=(0, 1))
        data = (data - means) / stds
        ei, ew = dense_to_sparse_numpy(np.asarray(adj_mx))
        rp, col, val = csr_by_destination(ei, ew, np.asarray(adj_mx).shape[0])
        arrays = {"series": np.ascontiguousarray(data), "edge_index": ei, "edge_weight": ew, "csr_rowptr": rp,
                  "csr_col": col, "csr_val": val, "means": means, "stds": stds}
        self._cache = TemporalGraphCache(self._NAME, {"nodes": int(num_nodes), "steps": int(data.shape[0])}, arrays, None)

    def write_cache(self, path=None):
        path = path or os.path.join(self.raw_data_dir, self._CACHE)
        return save_cache(path, self._cache.name, {k: np.asarray(v) for k, v in self._cache.arrays.items()},
                          self._cache.meta)

    def get_index_dataset(self, lags: int = 12, batch_size: int = 64, shuffle: bool = False, allGPU: int = -1,
                          ratio: Tuple[float, float, float] = (0.7, 0.1, 0.2), world_size: int = -1,
                          ddp_rank: int = -1, dask_batching: bool = False):
        """(train, val, test DataLoaders, edges [2, E], edge_weights [E], means, stds) as pems.py:71-179 returns them:
        float64 windows on the CPU path, float32 device-resident windows when `allGPU` names a GPU.  One difference in
        the last digits: the z-score statistics are the cache's (numpy, population std, float64) on both paths, where the
        reference's allGPU path uses torch.std (unbiased, float32) — a factor sqrt(T / (T - 1)) on `stds`."""
        edges = torch.from_numpy(np.array(self._cache.edge_index, dtype=np.int64))
        edge_weights = torch.from_numpy(np.array(self._cache.edge_weight))
        data = np.asarray(self._cache.series)         # (no copy: the loaders index it, the GPU path converts it once)
        means = torch.tensor(np.asarray(self._cache.means), dtype=torch.float)
        stds = torch.tensor(np.asarray(self._cache.stds), dtype=torch.float)
        if allGPU != -1:
            data = torch.from_numpy(data).to(f"cuda:{allGPU}", dtype=torch.float)
            means, stds = means.to(data.device).view(1, 1, -1), stds.to(data.device).view(1, 1, -1)
        parts = _split_indices(data.shape[0], lags, ratio)
        tr, va, te = _loaders(parts, data, lags, batch_size, shuffle, gpu=allGPU != -1, lazy=dask_batching,
                              world_size=world_size, ddp_rank=ddp_rank)
        return tr, va, te, edges, edge_weights, means, stds



class _OutOfScopeLoader:
    """A reference loader this package does not carry (not on the path `north_star` names): the name imports, so a
    script that only swaps its import line fails at the call with the reason instead of at the import with none."""
    _reference = ""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__} ({self._reference}) is outside this package's scope: its data ships with the reference "
            f"(there is no network here) and none of the models on the accelerated path use it.  Build a "
            f"StaticGraphTemporalSignal from your own arrays, or write them once with dataset.save_cache().")


class PedalMeDatasetLoader(_OutOfScopeLoader):
    _reference = "torch_geometric_temporal/dataset/pedalme.py"


class MontevideoBusDatasetLoader(_OutOfScopeLoader):
    _reference = "torch_geometric_temporal/dataset/montevideo_bus.py"
