#!/usr/bin/env python
"""The reference's index-batching DCRNN training loop (examples/indexBatching/DCRNN/pems_bay_main.py, pems_ddp.py)
on the drop-in modules, with METR-LA-shaped synthetic data (there is no network here for the real file).

    python examples/dcrnn_metrla_synthetic.py --epochs 1                       # one MI355X
    python examples/dcrnn_metrla_synthetic.py --epochs 1 --graph               # the whole step as ONE hipGraph (launch-bound
                                                                               # regime: batch 64 is ~250 launches of a few us)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/dcrnn_metrla_synthetic.py --epochs 1                           # 8 GPUs, one RCCL all-reduce per step
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd import dp  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.nn.conv import Linear  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN  # noqa: E402
from pytorch_geometric_temporal_amd.signal import IndexDataset  # noqa: E402


def masked_mae_loss(y_pred, y_true):          # examples/indexBatching/DCRNN/utils.py:10-18
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = torch.abs(y_pred - y_true) * mask
    return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--steps", type=int, default=4000, help="length of the synthetic series")
    ap.add_argument("--graph", action="store_true", help="capture forward + loss + backward + Adam as one hipGraph (1 GPU)")
    args = ap.parse_args()
    rank, local_rank, world = dp.init_from_env()
    dev = torch.device("cuda", local_rank)
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    edge_index, edge_weight = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    series = torch.from_numpy(syn.traffic_series(args.steps, 207, seed=1)).to(dev)      # resident [T, N, 2]
    lags = 12
    n = args.steps - (2 * lags - 1)
    train = IndexDataset(np.arange(int(0.7 * n)), series, lags, gpu=True)
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"rnn": BatchedDCRNN(2, args.hidden, K=3), "head": Linear(args.hidden, 2)}).to(dev)
    dp.broadcast_parameters(model)
    flat = dp.FlatGradients(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=args.graph)
    graphed = None
    if args.graph:
        if world != 1:
            raise SystemExit("--graph captures the optimizer step too: single GPU only (all-reduce between two graphs otherwise)")
        from pytorch_geometric_temporal_amd.graphed import GraphedStep
        index_tensor = torch.as_tensor(np.asarray(train.indices), device=dev)

        def step(starts):
            X, y = train_windows(starts)
            loss = masked_mae_loss(model["head"](model["rnn"](X, edge_index, edge_weight)), y)
            flat.zero()
            loss.backward()
            opt.step()
            return loss

        def train_windows(starts):                      # window gather on the device, indices stay on the device
            from pytorch_geometric_temporal_amd import ops
            return ops.window_gather(series, index_tensor[starts], lags)

        graphed = GraphedStep(step, (torch.zeros(args.batch_size, dtype=torch.int64, device=dev),))
    for epoch in range(args.epochs):
        order = dp.shard_indices(len(train), rank, world, epoch=epoch, shuffle=True)
        t0, total, nb, n_read = time.perf_counter(), 0.0, 0, 0
        order_dev = order.to(dev)
        for i in range(0, order.numel() - args.batch_size + 1, args.batch_size):
            if graphed is not None:
                loss = graphed(order_dev[i:i + args.batch_size])
                if nb % 50 == 0:                                      # a host read every 50 steps only
                    total, n_read = total + float(loss), n_read + 1
                nb += 1
                continue
            X, y = train.gather(order[i:i + args.batch_size].numpy())
            out = model["head"](model["rnn"](X, edge_index, edge_weight))
            loss = masked_mae_loss(out, y)
            flat.zero()
            loss.backward()
            flat.all_reduce_mean(world)
            opt.step()
            total += float(loss.detach())
            nb += 1
        torch.cuda.synchronize()
        stats = dp.reduce_scalars([total, n_read if graphed is not None else nb])
        if rank == 0:
            dt = time.perf_counter() - t0
            print(f"epoch {epoch}: mean train MAE {float(stats[0]) / max(float(stats[1]), 1):.4f}, {dt:.2f} s, "
                  f"{world * nb * args.batch_size * lags * 1515 / dt / 1e6:.1f} M snapshot-edges/s on {world} GPU(s)")


if __name__ == "__main__":
    main()
