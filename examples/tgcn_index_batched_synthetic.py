#!/usr/bin/env python
"""The reference's index-batched T-GCN training loop (examples/indexBatching/tgcn/metr_la_main.py) on the drop-in modules:
`BatchedTGCN` = T x (TGCN2 -> relu -> Linear), masked MAE on the de-normalised prediction, Adam — on a synthetic static graph
(BASELINE.json configs[3]: 50 000 nodes / 400 000 edges by default; there is no network here for the METR-LA file).

    python examples/tgcn_index_batched_synthetic.py --epochs 1 --nodes 5000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/tgcn_index_batched_synthetic.py --epochs 1                     # 8 GPUs, one RCCL all-reduce per step

The only change against the reference script is the import line (`torch_geometric_temporal.nn.recurrent` ->
`pytorch_geometric_temporal_amd.nn.recurrent`) and where the windows come from (a resident series instead of a DataLoader).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd import dp  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2  # noqa: E402


class BatchedTGCN(nn.Module):                 # metr_la_main.py:29-47, unchanged
    def __init__(self, in_channels, hidden_dim, out_channels):
        super().__init__()
        self.tgnn = TGCN2(in_channels, hidden_dim, 1)
        self.linear = nn.Linear(hidden_dim, out_channels)

    def forward(self, x, edge_index, edge_weight):
        B, N, Fin, T = x.shape
        h = None
        output_sequence = []
        for t in range(T):
            h = self.tgnn(x[..., t], edge_index, edge_weight, h)
            h_t = F.relu(h)
            output_sequence.append(self.linear(h_t).unsqueeze(1))
        return torch.cat(output_sequence, dim=1)


def masked_mae_loss(y_pred, y_true):          # metr_la_main.py:49-56
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = torch.abs(y_pred - y_true) * mask
    return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=50_000)
    ap.add_argument("--steps", type=int, default=400, help="length of the synthetic series")
    ap.add_argument("--windows", type=int, default=64, help="training windows per epoch")
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local_rank, world = dp.init_from_env()
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    seq = 12
    ei_np, ew_np = syn.local_graph(args.nodes, 8, seed=0)
    edge_index, edge_weight = torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device)
    series = torch.from_numpy(syn.traffic_series(args.steps, args.nodes, seed=1)).to(device)      # resident [T, N, 2]
    mean, std = 54.0, 19.5
    torch.manual_seed(0)
    model = BatchedTGCN(in_channels=2, out_channels=2, hidden_dim=32).to(device)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradients(model.parameters())       # every gradient in one buffer: ONE all-reduce per step
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    ar = torch.arange(seq, device=device)
    for epoch in range(args.epochs):
        starts = dp.shard_indices(args.windows, rank, world, epoch=epoch, shuffle=True, seed=0).to(device) % (args.steps - 2 * seq)
        t0, total, n = time.time(), 0.0, 0
        for i in range(0, starts.numel(), args.batch_size):
            idx = starts[i:i + args.batch_size]
            x = series[idx[:, None] + ar[None, :]].permute(0, 2, 3, 1)           # [B, N, F, T]  (metr_la_main.py:82-84)
            y = series[idx[:, None] + seq + ar[None, :]]
            y_hat = model(x, edge_index, edge_weight)
            loss = masked_mae_loss(y_hat * std + mean, y * std + mean)
            flat.zero()
            loss.backward()
            flat.all_reduce_mean(world)
            optimizer.step()
            total, n = total + float(loss.detach()), n + 1
        torch.cuda.synchronize()
        if rank == 0:
            print(f"epoch {epoch}: train MAE {total / max(n, 1):.4f}  ({time.time() - t0:.2f} s, {n} steps of {args.batch_size} windows "
                  f"x {world} GPU(s))")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
