"""Snapshot iterators of the reference (torch_geometric_temporal/signal) without the PyG dependency.

`StaticGraphTemporalSignal` / `DynamicGraphTemporalSignal` keep the reference's constructor, attributes
(`snapshot_count`, `features`, `targets`, extra keyword sequences), `[int] -> Data`, `[slice] -> signal`, re-iterable
`__iter__/__next__`, and `temporal_signal_split`.  Two MI355X-first additions, both opt-in and value-preserving:

* the static graph's `edge_index` / `edge_attr` tensors are created ONCE and handed out again for every snapshot
  (the reference rebuilds them per snapshot, static_graph_temporal_signal.py:113-115, which would defeat the
  identity-keyed graph-preparation cache of the kernels);
* `signal.to(device)` uploads the whole [T, N, F] feature / target arrays once and serves snapshots as device views
  ("GPU-index-batching", dataset/metr_la.py:180-190, applied to the snapshot iterator).
"""
from typing import Sequence, Union

import numpy as np
import torch

__all__ = ["Data", "StaticGraphTemporalSignal", "DynamicGraphTemporalSignal", "temporal_signal_split", "IndexDataset"]


class Data:
    """Attribute bag standing in for torch_geometric.data.Data: .x .edge_index .edge_attr .y + extra keys."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        self._extra = list(kwargs)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in ["x", "edge_index", "edge_attr", "y"] + self._extra if getattr(self, k) is not None]

    @property
    def num_nodes(self):
        if self.x is not None:
            return self.x.size(0)
        return int(self.edge_index.max()) + 1 if self.edge_index is not None and self.edge_index.numel() else 0

    def to(self, device):
        for k in self.keys():
            v = getattr(self, k)
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    def __repr__(self):
        parts = [f"{k}={list(getattr(self, k).shape)}" for k in self.keys() if isinstance(getattr(self, k), torch.Tensor)]
        return "Data(" + ", ".join(parts) + ")"


def _tensor(a):
    """torch.LongTensor / torch.FloatTensor by numpy kind (static_graph_temporal_signal.py:77-94)."""
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a
    a = np.asarray(a)
    if a.dtype.kind in "iu":
        return torch.as_tensor(a, dtype=torch.int64)
    if a.dtype.kind == "f":
        return torch.as_tensor(a, dtype=torch.float32)
    return None


def _stack_or_none(seq):
    """[T] sequence of equally shaped float arrays -> one [T, ...] float32 array (None if ragged / has None)."""
    try:
        if len(seq) == 0 or any(s is None for s in seq):
            return None
        first = np.asarray(seq[0])
        if first.dtype.kind != "f" or any(np.asarray(s).shape != first.shape for s in seq):
            return None
        return np.stack([np.asarray(s, dtype=np.float32) for s in seq])
    except Exception:
        return None


class _SignalBase:
    def _check_temporal_consistency(self):
        assert len(self.features) == len(self.targets), "Temporal dimension inconsistency."
        for key in self.additional_feature_keys:
            assert len(self.targets) == len(getattr(self, key)), "Temporal dimension inconsistency."

    def _set_snapshot_count(self):
        self.snapshot_count = len(self.features)

    def _get_features(self, t):
        if self._dev_features is not None:
            return self._dev_features[t]
        f = self.features[t]
        return None if f is None else torch.as_tensor(np.asarray(f), dtype=torch.float32).to(self._device)

    def _get_target(self, t):
        if self._dev_targets is not None:
            return self._dev_targets[t]
        y = _tensor(self.targets[t])
        return y if y is None else y.to(self._device)

    def _get_additional_features(self, t):
        out = {}
        for key in self.additional_feature_keys:
            v = _tensor(getattr(self, key)[t])
            out[key] = v if v is None else v.to(self._device)
        return out

    def to(self, device):
        """Make the signal device-resident: one upload of the stacked features / targets, snapshots become views."""
        self._device = torch.device(device)
        f, y = _stack_or_none(self.features), _stack_or_none(self.targets)
        self._dev_features = None if f is None else torch.from_numpy(f).to(self._device)
        self._dev_targets = None if y is None else torch.from_numpy(y).to(self._device)
        self._graph_tensors = {}
        return self

    def __next__(self):
        if self.t < len(self.features):
            snapshot = self[self.t]
            self.t = self.t + 1
            return snapshot
        self.t = 0
        raise StopIteration

    def __iter__(self):
        self.t = 0
        return self

    def __len__(self):
        return self.snapshot_count


class StaticGraphTemporalSignal(_SignalBase):
    r"""Static graph, temporal node features and targets (reference: signal/static_graph_temporal_signal.py:13-134).

    Args: edge_index [2,E] numpy, edge_weight [E] numpy, features / targets: sequences of per-snapshot numpy arrays,
    **kwargs: extra per-snapshot sequences."""

    def __init__(self, edge_index, edge_weight, features: Sequence, targets: Sequence, **kwargs):
        self.edge_index = edge_index
        self.edge_weight = edge_weight
        self.features = features
        self.targets = targets
        self.additional_feature_keys = []
        for key, value in kwargs.items():
            setattr(self, key, value)
            self.additional_feature_keys.append(key)
        self._check_temporal_consistency()
        self._set_snapshot_count()
        self._device = torch.device("cpu")
        self._dev_features = self._dev_targets = None
        self._graph_tensors = {}

    def _get_edge_index(self):
        if self.edge_index is None:
            return None
        if "ei" not in self._graph_tensors:      # memoised: the same tensor object for every snapshot
            self._graph_tensors["ei"] = torch.as_tensor(np.asarray(self.edge_index), dtype=torch.int64).to(self._device)
        return self._graph_tensors["ei"]

    def _get_edge_weight(self):
        if self.edge_weight is None:
            return None
        if "ew" not in self._graph_tensors:
            self._graph_tensors["ew"] = torch.as_tensor(np.asarray(self.edge_weight), dtype=torch.float32).to(self._device)
        return self._graph_tensors["ew"]

    def __getitem__(self, time_index: Union[int, slice]):
        if isinstance(time_index, slice):
            s = StaticGraphTemporalSignal(
                self.edge_index, self.edge_weight, self.features[time_index], self.targets[time_index],
                **{key: getattr(self, key)[time_index] for key in self.additional_feature_keys})
            if self._device.type != "cpu":
                s.to(self._device)
            return s
        return Data(x=self._get_features(time_index), edge_index=self._get_edge_index(),
                    edge_attr=self._get_edge_weight(), y=self._get_target(time_index),
                    **self._get_additional_features(time_index))


class DynamicGraphTemporalSignal(_SignalBase):
    r"""Per-snapshot graphs, features and targets (reference: signal/dynamic_graph_temporal_signal.py:13-139)."""

    def __init__(self, edge_indices: Sequence, edge_weights: Sequence, features: Sequence, targets: Sequence, **kwargs):
        self.edge_indices = edge_indices
        self.edge_weights = edge_weights
        self.features = features
        self.targets = targets
        self.additional_feature_keys = []
        for key, value in kwargs.items():
            setattr(self, key, value)
            self.additional_feature_keys.append(key)
        self._check_temporal_consistency()
        self._set_snapshot_count()
        self._device = torch.device("cpu")
        self._dev_features = self._dev_targets = None
        self._graph_tensors = {}

    def _check_temporal_consistency(self):
        assert len(self.features) == len(self.targets), "Temporal dimension inconsistency."
        assert len(self.edge_indices) == len(self.edge_weights), "Temporal dimension inconsistency."
        assert len(self.features) == len(self.edge_weights), "Temporal dimension inconsistency."
        for key in self.additional_feature_keys:
            assert len(self.targets) == len(getattr(self, key)), "Temporal dimension inconsistency."

    def _graph_at(self, t):
        # memoised per step: a second epoch hands out the same tensors, so graph preparation is reused
        if t not in self._graph_tensors:
            ei, ew = self.edge_indices[t], self.edge_weights[t]
            ei = None if ei is None else torch.as_tensor(np.asarray(ei), dtype=torch.int64).to(self._device)
            ew = None if ew is None else torch.as_tensor(np.asarray(ew), dtype=torch.float32).to(self._device)
            self._graph_tensors[t] = (ei, ew)
        return self._graph_tensors[t]

    def __getitem__(self, time_index: Union[int, slice]):
        if isinstance(time_index, slice):
            s = DynamicGraphTemporalSignal(
                self.edge_indices[time_index], self.edge_weights[time_index], self.features[time_index],
                self.targets[time_index],
                **{key: getattr(self, key)[time_index] for key in self.additional_feature_keys})
            if self._device.type != "cpu":
                s.to(self._device)
            return s
        ei, ew = self._graph_at(time_index)
        return Data(x=self._get_features(time_index), edge_index=ei, edge_attr=ew, y=self._get_target(time_index),
                    **self._get_additional_features(time_index))


def temporal_signal_split(data_iterator, train_ratio: float = 0.8):
    """Split a signal by a fixed ratio (reference: signal/train_test_split.py:36-54)."""
    train_snapshots = int(train_ratio * data_iterator.snapshot_count)
    return data_iterator[0:train_snapshots], data_iterator[train_snapshots:]


class IndexDataset(torch.utils.data.Dataset):
    r"""Index-batching dataset (reference: signal/index_dataset.py:7-57) without the hard dask import: sample i is
    (data[idx : idx+h], data[idx+h : idx+2h]) for idx = indices[i].  `gpu=True`: `data` is a device tensor and the
    samples are views; `lazy=True`: `data` is any array whose slices have `.compute()` (a dask array)."""

    def __init__(self, indices, data, horizon, lazy=False, gpu=False):
        self.indices = indices
        self.data = data
        self.horizon = horizon
        self.lazy = lazy
        self.gpu = gpu

    def __len__(self):
        return self.indices.shape[0]

    def __getitem__(self, x):
        idx = self.indices[x]
        y_start = idx + self.horizon
        if self.gpu:
            return self.data[idx:y_start, ...], self.data[y_start:y_start + self.horizon, ...]
        if self.lazy:
            return (torch.from_numpy(self.data[idx:y_start, ...].compute()),
                    torch.from_numpy(self.data[y_start:y_start + self.horizon, ...].compute()))
        return torch.from_numpy(self.data[idx:y_start, ...]), torch.from_numpy(self.data[y_start:y_start + self.horizon, ...])

    def _check_window_range(self):
        """Every start index must leave room for both windows (idx + 2 h <= T): the reference's slicing would hand back a
        short window, the fused gather kernel clamps rows instead of reading outside the series — neither is what a
        caller wants, so the range is checked once, where the indices are (one reduction; cached)."""
        if getattr(self, "_range_checked", False):
            return
        n = int(self.data.shape[0])
        idx = self.indices
        if len(idx):
            lo = int(idx.min())
            hi = int(idx.max())
            if lo < 0 or hi + 2 * int(self.horizon) > n:
                raise IndexError(f"IndexDataset: start indices span [{lo}, {hi}] but windows of 2 x {self.horizon} steps "
                                 f"need 0 <= idx <= {n - 2 * int(self.horizon)} (series length {n})")
        self._range_checked = True

    def gather(self, batch_indices, device=None):
        """Whole batch in one fused gather from a resident [T, N, F] tensor: -> (X [B,h,N,F], Y [B,h,N,F])."""
        self._check_window_range()
        data = self.data if isinstance(self.data, torch.Tensor) else torch.from_numpy(np.asarray(self.data))
        if device is not None:
            data = data.to(device)
        if isinstance(batch_indices, torch.Tensor) and isinstance(self.indices, torch.Tensor):
            idx = self.indices.to(data.device)[batch_indices.to(data.device)].long()       # no host round trip
        else:
            idx = torch.as_tensor(np.asarray(self.indices)[np.asarray(batch_indices)], device=data.device).long()
        if data.is_cuda and data.dtype == torch.float32:
            # resident series: both windows of the whole batch in ONE launch (pgt_window_gather_f32)
            from .. import ops
            return ops.window_gather(data, idx, self.horizon)
        ar = torch.arange(self.horizon, device=data.device)
        return data[idx[:, None] + ar[None, :]], data[idx[:, None] + self.horizon + ar[None, :]]
