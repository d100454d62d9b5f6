"""Snapshot iterators: values pinned by the fixture produced with the reference's own
StaticGraphTemporalSignal on the vendored Chickenpox file; behaviour pinned by the reference's iterator contract
(test/dataset_test.py:304-312, test/index_test.py:93-115)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from pytorch_geometric_temporal_amd.signal import (Data, DynamicGraphTemporalSignal, IndexDataset,
                                                   StaticGraphTemporalSignal, temporal_signal_split)


def _chickenpox_like(g):
    fx = g["in"]["FX_head"].numpy()
    lags = int(g["meta"]["lags"])
    feats = [fx[i:i + lags, :].T for i in range(fx.shape[0] - lags)]
    targs = [fx[i + lags, :].T for i in range(fx.shape[0] - lags)]
    return g["out"]["edge_index"].numpy(), g["out"]["edge_weight"].numpy(), feats, targs


def test_static_signal_matches_reference_fixture():
    g = load_golden("chickenpox_signal_head")
    ei, ew, feats, targs = _chickenpox_like(g)
    s = StaticGraphTemporalSignal(ei, ew, feats, targs)
    assert s.snapshot_count == len(feats)
    snaps = list(s)
    for t in range(3):
        assert isinstance(snaps[t], Data)
        assert torch.equal(snaps[t].x, g["out"][f"x{t}"]) and snaps[t].x.dtype == torch.float32
        assert torch.equal(snaps[t].y, g["out"][f"y{t}"])
        assert torch.equal(snaps[t].edge_index, g["out"]["edge_index"]) and snaps[t].edge_index.dtype == torch.int64
        assert torch.equal(snaps[t].edge_attr, g["out"]["edge_weight"])
    # the static graph tensors are the SAME objects for every snapshot (identity-keyed graph-prep cache hits)
    assert snaps[0].edge_index is snaps[1].edge_index and snaps[0].edge_attr is snaps[2].edge_attr
    assert len(list(s)) == len(feats)          # re-iterable (self.t reset, static_graph_temporal_signal.py:129,133)
    sub = s[1:3]
    assert isinstance(sub, StaticGraphTemporalSignal) and sub.snapshot_count == 2
    assert torch.equal(sub[0].x, snaps[1].x)


def test_split_additional_features_and_integer_targets():
    rng = np.random.default_rng(0)
    ei = rng.integers(0, 5, size=(2, 12))
    feats = [rng.random((5, 3)) for _ in range(10)]
    targs = [rng.integers(0, 3, size=5) for _ in range(10)]
    extra = [rng.random((5,)) for _ in range(10)]
    s = StaticGraphTemporalSignal(ei, None, feats, targs, mask=extra)
    tr, te = temporal_signal_split(s, train_ratio=0.8)
    assert tr.snapshot_count == 8 and te.snapshot_count == 2
    snap = te[1]
    assert snap.edge_attr is None and snap.y.dtype == torch.int64 and snap.mask.dtype == torch.float32
    assert torch.allclose(snap.mask, torch.tensor(extra[9], dtype=torch.float32))
    assert snap.x.dtype == torch.float32 and snap.num_nodes == 5
    try:
        StaticGraphTemporalSignal(ei, None, feats, targs[:-1])
        raise SystemExit("expected an assertion")
    except AssertionError as e:
        assert "Temporal dimension inconsistency" in str(e)


def test_dynamic_signal():
    rng = np.random.default_rng(1)
    eis = [rng.integers(0, 6, size=(2, 7 + t)) for t in range(4)]
    ews = [rng.random(7 + t) for t in range(4)]
    feats = [rng.random((6, 2)) for _ in range(4)]
    targs = [rng.random(6) for _ in range(4)]
    s = DynamicGraphTemporalSignal(eis, ews, feats, targs)
    snaps = list(s)
    assert [sn.edge_index.shape[1] for sn in snaps] == [7, 8, 9, 10]
    assert torch.allclose(snaps[2].edge_attr, torch.tensor(ews[2], dtype=torch.float32))
    assert s[2].edge_index is snaps[2].edge_index          # memoised per step
    tr, te = temporal_signal_split(s, 0.5)
    assert tr.snapshot_count == 2 and te[0].edge_index.shape[1] == 9


def test_resident_signal_serves_views_of_one_upload():
    rng = np.random.default_rng(2)
    ei = rng.integers(0, 4, size=(2, 9))
    feats = [rng.random((4, 3)) for _ in range(6)]
    targs = [rng.random(4) for _ in range(6)]
    s = StaticGraphTemporalSignal(ei, np.ones(9), feats, targs).to("cpu")
    a, b = s[1], s[2]
    assert a.x.untyped_storage().data_ptr() == b.x.untyped_storage().data_ptr()
    assert torch.allclose(a.x, torch.tensor(feats[1], dtype=torch.float32))
    assert torch.allclose(b.y, torch.tensor(targs[2], dtype=torch.float32))


def test_index_dataset_equals_snapshot_windows():
    """index-batching yields exactly the windows of the full array (test/index_test.py:93-115)."""
    data = np.random.default_rng(3).random((40, 5, 2)).astype(np.float32)
    idx = np.arange(0, 40 - 2 * 4)
    ds = IndexDataset(idx, data, horizon=4)
    assert len(ds) == idx.shape[0]
    x, y = ds[3]
    assert torch.equal(x, torch.from_numpy(data[3:7])) and torch.equal(y, torch.from_numpy(data[7:11]))
    dsg = IndexDataset(idx, torch.from_numpy(data), horizon=4, gpu=True)
    xg, yg = dsg[3]
    assert torch.equal(xg, x) and torch.equal(yg, y)
    X, Y = dsg.gather([0, 3, 5])
    assert X.shape == (3, 4, 5, 2) and torch.equal(X[1], x) and torch.equal(Y[1], y)
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False)
    xb, yb = next(iter(loader))
    assert xb.shape == (8, 4, 5, 2) and torch.equal(xb[3], x)


@pytest.mark.parametrize("time_major", [False, True])
@pytest.mark.parametrize("shape", [(40, 5, 2), (33, 7, 1), (64, 207, 2)])
def test_window_gather_kernel_equals_the_reference_windows(backend, shape, time_major):
    """pgt_window_gather_f32: both windows of a whole index batch in one launch == IndexDataset.__getitem__ per sample
    (signal/index_dataset.py:32-57), batch-major and time-major."""
    from pytorch_geometric_temporal_amd import ops
    T, h = shape[0], 4
    data = torch.randn(*shape, generator=torch.Generator().manual_seed(T))
    starts = torch.tensor([0, 3, T - 2 * h, 5, 3], dtype=torch.int64)
    X, Y = ops.window_gather(backend.t(data), backend.t(starts), h, time_major=time_major)
    ds = IndexDataset(starts.numpy(), data.numpy(), horizon=h)
    for b in range(starts.numel()):
        x, y = ds[b]
        xs, ys = (X[:, b], Y[:, b]) if time_major else (X[b], Y[b])
        assert torch.equal(xs.cpu(), x) and torch.equal(ys.cpu(), y)


@pytest.mark.gpu
def test_resident_signal_and_index_batches_on_the_gpu():
    """§8 f2 on the device: `.to(cuda)` uploads the series once and every snapshot is a view of that upload with
    memoised edge tensors (the graph cache then hits by identity); IndexDataset.gather on the resident series runs
    the one-launch window gather and equals the CPU dataset sample for sample."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    from pytorch_geometric_temporal_amd.dataset import synthetic as syn
    ei, ew = syn.sensor_graph(30, 120, seed=4)                  # unique edges (DConv's dense path rejects duplicates)
    feats = [rng.random((30, 3)).astype(np.float32) for _ in range(20)]
    targs = [rng.random(30).astype(np.float32) for _ in range(20)]
    s = StaticGraphTemporalSignal(ei, ew, feats, targs).to(dev)
    a, b = s[1], s[7]
    assert a.x.is_cuda and a.x.untyped_storage().data_ptr() == b.x.untyped_storage().data_ptr()
    assert a.edge_index is b.edge_index and a.edge_attr is b.edge_attr
    assert torch.equal(b.x.cpu(), torch.from_numpy(feats[7])) and torch.equal(b.y.cpu(), torch.from_numpy(targs[7]))
    from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN
    from pytorch_geometric_temporal_amd import ops
    m = DCRNN(3, 4, 2).to(dev)
    ops.GRAPH_CACHE.clear()
    with torch.no_grad():
        for snap in s:
            m(snap.x, snap.edge_index, snap.edge_attr)
    assert len(ops.GRAPH_CACHE._d) == 1                      # one prepared graph for the whole signal
    data = torch.randn(500, 207, 2)
    idx = np.arange(0, 500 - 24)
    cpu_ds = IndexDataset(idx, data.numpy(), horizon=12)
    gpu_ds = IndexDataset(idx, data.to(dev), horizon=12, gpu=True)
    batch = [0, 17, 475, 230, 17]
    X, Y = gpu_ds.gather(batch)
    assert X.is_cuda and X.shape == (5, 12, 207, 2)
    for j, i in enumerate(batch):
        x, y = cpu_ds[i]
        assert torch.equal(X[j].cpu(), x) and torch.equal(Y[j].cpu(), y)
    xg, yg = gpu_ds[230]
    assert torch.equal(xg.cpu(), cpu_ds[230][0])


def test_index_dataset_gather_rejects_windows_that_leave_the_series():
    """An index that leaves no room for both windows would be clamped by the fused gather kernel (and silently shortened by
    the reference's slicing): IndexDataset.gather checks the index array once and raises."""
    import numpy as np
    from pytorch_geometric_temporal_amd.signal import IndexDataset
    data = np.arange(40 * 3 * 2, dtype=np.float32).reshape(40, 3, 2)
    ok = IndexDataset(np.arange(0, 40 - 2 * 6 + 1), data, 6)
    X, Y = ok.gather(np.array([0, 28]))
    assert X.shape == (2, 6, 3, 2) and float(Y[1, -1, 0, 0]) == float(data[39, 0, 0])
    bad = IndexDataset(np.arange(0, 40 - 2 * 6 + 2), data, 6)
    with pytest.raises(IndexError, match="start indices"):
        bad.gather(np.array([0]))
