#!/usr/bin/env python
"""The reference's index-batched A3T-GCN loop (examples/indexBatching/A3TGCN/pems_bay_main.py / metr_la_main.py: TemporalGNN =
A3TGCN2(2, 32, periods = 12) -> relu -> Linear(32, 12), MSE on the speed channel, Adam 1e-3) on the drop-in module, over a synthetic
PeMS-Bay-sized sensor graph (325 nodes / 2 694 edges: BASELINE.json configs[2]; no network here for the dataset file).  The series
stays resident on the GPU and a batch is one gather of B windows (the reference's "GPU index batching").

    python examples/a3tgcn_index_batched_synthetic.py --epochs 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/a3tgcn_index_batched_synthetic.py
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd import dp  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import A3TGCN2  # noqa: E402


class TemporalGNN(torch.nn.Module):           # pems_bay_main.py: the attention cell, then a single-shot read-out of all periods
    def __init__(self, node_features, periods, batch_size):
        super().__init__()
        self.tgnn = A3TGCN2(in_channels=node_features, out_channels=32, periods=periods, batch_size=batch_size)
        self.linear = torch.nn.Linear(32, periods)

    def forward(self, x, edge_index):
        h = self.tgnn(x, edge_index)          # x [B, N, F, T] -> [B, N, 32]
        return self.linear(F.relu(h))         # [B, N, T]


def main(argv=None, device=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=325)
    ap.add_argument("--edges", type=int, default=2694)
    ap.add_argument("--steps", type=int, default=2000, help="length of the synthetic series")
    ap.add_argument("--windows", type=int, default=512, help="training windows per epoch")
    args = ap.parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local_rank, world = dp.init_from_env()
    device = device or torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    periods = 12
    ei_np, _ = syn.sensor_graph(args.nodes, args.edges, seed=0, symmetric=True)
    edge_index = torch.from_numpy(ei_np).to(device)
    series = torch.from_numpy(syn.traffic_series(args.steps, args.nodes, seed=1)).to(device)       # resident [T, N, 2]
    torch.manual_seed(0)
    model = TemporalGNN(node_features=2, periods=periods, batch_size=args.batch_size).to(device)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradients(model.parameters())                # every gradient in one buffer: ONE all-reduce per step
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    loss_fn = torch.nn.MSELoss()
    ar = torch.arange(periods, device=device)
    last = float("nan")
    for epoch in range(args.epochs):
        starts = dp.shard_indices(args.windows, rank, world, epoch=epoch, shuffle=True, seed=0).to(device) % (args.steps - 2 * periods)
        t0, total, n = time.time(), 0.0, 0
        for i in range(0, starts.numel(), args.batch_size):
            idx = starts[i:i + args.batch_size]
            x = series[idx[:, None] + ar[None, :]].permute(0, 2, 3, 1)             # [B, N, F, T]
            y = series[idx[:, None] + periods + ar[None, :]][..., 0].permute(0, 2, 1)  # speed channel, [B, N, T]
            loss = loss_fn(model(x, edge_index), y)
            flat.zero()
            loss.backward()
            flat.all_reduce_mean(world)
            optimizer.step()
            total, n = total + float(loss.detach()), n + 1
        if device.type == "cuda":
            torch.cuda.synchronize()
        last = total / max(n, 1)
        if rank == 0:
            print(f"epoch {epoch}: train MSE {last:.4f}  ({time.time() - t0:.2f} s, {n} steps of {args.batch_size} windows x {world} GPU(s))")
    if world > 1:
        torch.distributed.destroy_process_group()
    return last


if __name__ == "__main__":
    main()
