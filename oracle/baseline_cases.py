"""TEST INFRASTRUCTURE — inputs of the BASELINE.json configurations at their OWN shapes, rebuilt from seeds.

The fixtures `tests/golden/baseline_*.npz` (written by oracle/make_golden.py from the reference's actual module files)
hold the parameters, a sample of the reference output and per-slice checksums; the multi-megabyte inputs are NOT stored:
both the fixture generator (build container) and the parity tests (GPU box) rebuild them here from seeds with the
same numpy / torch CPU generators, so the two sides see identical bits.

  config 2  METR-LA-shaped: 207 nodes, 1 515 edges (BASELINE.json) and 1 722 edges (the reference's data,
            test/dataset_test.py:397), BatchedDCRNN(2, 64, K=3), B = 64, 12 steps — the benchmarked model
  config 3  PeMS-BAY-shaped: 325 nodes / 2 694 edges, A3TGCN2(2, 32, periods=12), B = 64
  config 4  50 000 nodes / 400 000 edges, TGCN2(2, 32), B = 8
  config 5  the vendored england_covid.json through EvolveGCNH(129, 8), every snapshot the loader yields
"""
import numpy as np
import torch

from pytorch_geometric_temporal_amd.dataset import synthetic as syn


def rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def randomise(module, seed, gain=1.0):
    """Re-draw every parameter (including the zero-initialised biases) so that no term is trivially zero.  `gain`
    scales the matrices: a 12-step GRU at hidden 64 with weights at gain 1 is a chaotic map — the reference's OWN fp32
    forward then sits 1e-4 away from an fp64 evaluation of itself at the last step, so no reordering of the sums can
    agree with it to 1e-5; at gain 0.25 (still 3x the reference's xavier initialisation, dcrnn.py:35-37) fp32 and
    fp64 agree to 3e-7 and the 1e-5 comparison measures the implementation, not the conditioning."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in module.named_parameters():
            bound = gain * (6.0 / (p.size(-2) + p.size(-1))) ** 0.5 if p.dim() >= 2 else 0.5
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)


def metrla(num_edges):
    """config 2: (edge_index, edge_weight, X [64, 12, 207, 2])."""
    ei, ew = syn.sensor_graph(207, num_edges, seed=0, symmetric=False)
    return torch.from_numpy(ei), torch.from_numpy(ew), rand((64, 12, 207, 2), 201)


METRLA_SAMPLE_B, METRLA_SAMPLE_T = (0, 37, 63), (0, 5, 11)


def pemsbay():
    """config 3: (edge_index, edge_weight, X [64, 325, 2, 12], H0 [64, 325, 32])."""
    ei, ew = syn.sensor_graph(325, 2694, seed=0, symmetric=False)
    return torch.from_numpy(ei), torch.from_numpy(ew), rand((64, 325, 2, 12), 301), rand((64, 325, 32), 302)


PEMSBAY_SAMPLE_B = (0, 17, 40, 63)


def graph50k(kind):
    """config 4: (edge_index, edge_weight, X [8, 50 000, 2], H0 [8, 50 000, 32]); `kind` = "local" (locality-ordered)
    or "uniform" (no locality: neighbours anywhere)."""
    gen = syn.local_graph if kind == "local" else syn.uniform_graph
    ei, ew = gen(50_000, 8, seed=3)
    return torch.from_numpy(ei), torch.from_numpy(ew), rand((8, 50_000, 2), 401), rand((8, 50_000, 32), 402)


def sample_nodes_50k():
    return torch.from_numpy(np.random.default_rng(44).choice(50_000, size=384, replace=False).astype(np.int64)).sort().values


def slice_sums(out, keep_dims):
    """fp64 sums over every axis NOT in keep_dims (a checksum per slice)."""
    red = [d for d in range(out.dim()) if d not in keep_dims]
    return out.detach().double().cpu().sum(dim=red)
