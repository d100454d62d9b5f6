"""Seeded synthetic graphs / signals with the shapes of the reference's datasets (there is no network here):
METR-LA (207 nodes; 1 515 edges per BASELINE.json, 1 722 in the reference's data, test/dataset_test.py:397),
PeMS-BAY (325 / 2 694, test/dataset_test.py:419), plus the large roofline graphs of SURVEY.md §8(d).
All generators are numpy-only and deterministic in `seed`.  edge_index follows the reference's convention:
int64 [2, E], row = source, col = target, sorted row-major as dense_to_sparse yields (dataset/metr_la.py:91-96).
"""
import numpy as np


def _row_major(src, dst, w):
    order = np.lexsort((dst, src))
    return np.stack([src[order], dst[order]]).astype(np.int64), w[order].astype(np.float32)


def sensor_graph(num_nodes=207, num_edges=1515, seed=0, symmetric=False):
    """Thresholded-Gaussian-kernel k-NN adjacency on random 2-D sensor coordinates (the DCRNN recipe), unit
    diagonal included (real METR-LA/PeMS adjacencies have one; it also keeps every in-degree > 0 so DConv stays
    finite).  `symmetric=True` gives a structurally symmetric pattern (reverse list positionally aligned);
    False gives a directed pattern that exercises DConv's positional norm_in quirk."""
    rng = np.random.default_rng(seed)
    n = int(num_nodes)
    m = int(num_edges) - n
    if m < 0:
        raise ValueError("num_edges must be at least num_nodes (self-loops)")
    xy = rng.random((n, 2))
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    sigma2 = np.median(np.sort(d2, axis=1)[:, 1:9])
    wfull = np.exp(-d2 / sigma2)
    np.fill_diagonal(wfull, -1.0)
    if symmetric:
        if m % 2:
            raise ValueError("symmetric pattern needs an even number of off-diagonal edges")
        iu = np.triu_indices(n, 1)
        pick = np.argsort(-wfull[iu], kind="stable")[: m // 2]
        a, b = iu[0][pick], iu[1][pick]
        src, dst = np.concatenate([a, b]), np.concatenate([b, a])
        # direction-dependent weights (one-way streets): pattern symmetric, values not
        w = wfull[src, dst] * (0.75 + 0.5 * rng.random(src.size))
    else:
        noisy = wfull * (0.6 + 0.8 * rng.random((n, n)))
        np.fill_diagonal(noisy, -1.0)
        flat = np.argsort(-noisy, axis=None, kind="stable")[:m]
        src, dst = np.unravel_index(flat, (n, n))
        w = wfull[src, dst]
    loops = np.arange(n)
    src = np.concatenate([src, loops])
    dst = np.concatenate([dst, loops])
    w = np.concatenate([w, np.ones(n)])
    return _row_major(src, dst, w)


def watts_strogatz_directed(num_nodes=100, k=10, p=0.5, seed=0):
    """The reference tests' mock graph (test/recurrent_test.py:16-23): networkx.watts_strogatz_graph(...).edges(),
    every undirected edge listed once -> a directed, asymmetric list with zero-in-degree nodes (DConv then emits
    inf/nan, which the parity tests must reproduce)."""
    import networkx as nx
    g = nx.watts_strogatz_graph(int(num_nodes), int(k), float(p), seed=int(seed))
    e = np.array(list(g.edges()), dtype=np.int64).T
    return e


def local_graph(num_nodes=200_000, degree=8, window=64, seed=0):
    """Locality-ordered graph: every node draws `degree` distinct in-neighbours within +-window/2 positions
    (what a bandwidth-reducing ordering of a road network looks like).  E = num_nodes * degree exactly."""
    rng = np.random.default_rng(seed)
    n, d = int(num_nodes), int(degree)
    offs_all = np.concatenate([np.arange(-(window // 2), 0), np.arange(1, window // 2 + 1)])
    pick = np.argsort(rng.random((n, offs_all.size)), axis=1)[:, :d]
    offs = offs_all[pick]
    dst = np.repeat(np.arange(n), d)
    src = (dst + offs.reshape(-1)) % n
    w = (0.5 + rng.random(src.size)).astype(np.float32)
    return _row_major(src, dst, w)


def hub_graph(num_nodes=200_000, degree=8, hubs=20, hub_degree=2000, seed=0, base=None):
    """local_graph (or `base` = (edge_index, edge_weight) of another graph on `num_nodes` nodes) plus `hubs` nodes that also receive
    `hub_degree` uniformly drawn in-edges (a skewed in-degree distribution: the long rows a row-per-lane-group aggregation kernel
    chokes on)."""
    rng = np.random.default_rng(seed + 1)
    n = int(num_nodes)
    ei, ew = local_graph(n, degree, seed=seed) if base is None else base
    rows = rng.choice(n, hubs, replace=False)
    src = np.concatenate([rng.choice(n, hub_degree, replace=False) for _ in rows])
    dst = np.repeat(rows, hub_degree)
    e2 = np.concatenate([ei, np.stack([src, dst])], axis=1)
    w2 = np.concatenate([ew, (0.5 + rng.random(src.size)).astype(np.float32)])
    keep = np.unique(e2[0].astype(np.int64) * n + e2[1], return_index=True)[1]
    return _row_major(e2[0][keep], e2[1][keep], w2[keep])


def uniform_graph(num_nodes=200_000, degree=8, seed=0):
    """Uniform-random in-neighbours (no locality; worst case for the L2): `degree` random permutations, duplicates
    removed, so E is within a few edges of num_nodes * degree."""
    rng = np.random.default_rng(seed)
    n, d = int(num_nodes), int(degree)
    dst = np.tile(np.arange(n), d)
    src = np.concatenate([rng.permutation(n) for _ in range(d)])
    key = np.unique(src.astype(np.int64) * n + dst)
    src, dst = key // n, key % n
    w = (0.5 + rng.random(src.size)).astype(np.float32)
    return _row_major(src, dst, w)


def _hilbert_d(order, x, y):
    """Index of cell (x, y) along the Hilbert curve of a 2**order x 2**order square (vectorised)."""
    x, y = x.astype(np.int64).copy(), y.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (order - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = ry == 0
        swap = flip & (rx == 1)
        x = np.where(swap, s - 1 - x, x)
        y = np.where(swap, s - 1 - y, y)
        x, y = np.where(flip, y, x), np.where(flip, x, y)
        s >>= 1
    return d


def grid2d_graph(side=447, order="hilbert", seed=0):
    """A 2-D mesh (side x side nodes, 8-neighbourhood: in-degree 8 in the interior — a road / sensor network embedded in
    the plane) numbered along a space-filling curve ("hilbert": consecutive nodes form compact patches, the ordering a
    locality-aware partitioner yields), row by row ("rowmajor": the bandwidth-`side` ordering reverse Cuthill-McKee gives
    a mesh) or at random ("shuffled").  N = side**2 (447**2 = 199 809), E ~ 8 N.  Weights are random: a diffusion operator, not a stencil."""
    rng = np.random.default_rng(seed)
    n_side = int(side)
    yy, xx = np.divmod(np.arange(n_side * n_side), n_side)
    if order == "hilbert":
        bits = max(1, int(np.ceil(np.log2(n_side))))
        rank = np.empty(n_side * n_side, dtype=np.int64)
        rank[np.argsort(_hilbert_d(bits, xx, yy), kind="stable")] = np.arange(n_side * n_side)
    elif order == "rowmajor":
        rank = np.arange(n_side * n_side, dtype=np.int64)
    elif order == "shuffled":                      # node ids in no order at all (a sensor list in file order)
        rank = np.random.default_rng(seed + 7).permutation(n_side * n_side).astype(np.int64)
    else:
        raise ValueError(order)
    src, dst = [], []
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            ok = (xx + dx >= 0) & (xx + dx < n_side) & (yy + dy >= 0) & (yy + dy < n_side)
            dst.append(rank[ok])
            src.append(rank[(yy[ok] + dy) * n_side + xx[ok] + dx])
    src, dst = np.concatenate(src), np.concatenate(dst)
    w = (0.5 + rng.random(src.size)).astype(np.float32)
    return _row_major(src, dst, w)


def traffic_series(num_steps, num_nodes, seed=0):
    """[T, N, 2] float32: z-scored AR(1) "speed" channel + time-of-day channel (layout of dataset/metr_la.py:143-176)."""
    rng = np.random.default_rng(seed)
    T, n = int(num_steps), int(num_nodes)
    x = np.empty((T, n), dtype=np.float32)
    x[0] = rng.standard_normal(n)
    noise = rng.standard_normal((T, n)).astype(np.float32)
    for t in range(1, T):
        x[t] = 0.9 * x[t - 1] + 0.4359 * noise[t]
    tod = ((np.arange(T) % 288) / 288.0).astype(np.float32)
    out = np.stack([x, np.broadcast_to(tod[:, None], (T, n))], axis=-1)
    return np.ascontiguousarray(out, dtype=np.float32)
