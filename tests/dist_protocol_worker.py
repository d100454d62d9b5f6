"""Worker of tests/test_distributed.py::test_bench_protocol_*: the multi-rank protocol of bench.py (dp.timed_steps:
barrier, exact step count, MAX-reduce of the elapsed time; flat all-reduce issued async and finished before the update;
uneven shards padded like DistributedSampler; per-epoch scalar reduce) under gloo with a plain torch model — no kernels
involved, so world sizes up to 8 run in seconds."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import dp  # noqa: E402


def main():
    out_path, total = sys.argv[1], int(sys.argv[2])
    rank, _, world = dp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    dp.broadcast_parameters(model, src=0)
    flat = dp.FlatParameters(model.parameters())
    opt = flat.optimizer(torch.optim.SGD, lr=0.05)
    g = torch.Generator().manual_seed(3)
    data, target = torch.randn(total, 6, generator=g), torch.randn(total, 1, generator=g)
    calls = []

    def step(i):
        mine = dp.shard_indices(total, rank, world, epoch=i, shuffle=True, seed=11)     # uneven totals are padded
        loss = (model(data[mine]) - target[mine]).pow(2).mean()
        flat.zero()
        loss.backward()
        work = flat.all_reduce_mean(world, async_op=True)
        if rank == world - 1:
            time.sleep(0.02)                              # one slow rank: the MAX must report it on every rank
        flat.finish(work, world)
        opt.step()
        calls.append(i)
        return loss

    dt, last = dp.timed_steps(step, 2, 3)
    dp.timed_steps(step, 5, 2)                            # the instrumented steps run on EVERY rank (collective inside)
    sums = dp.reduce_scalars([float(last), 1.0])
    res = {"rank": rank, "world": world, "dt": dt, "calls": calls, "shard": int(dp.shard_indices(total, rank, world).numel()),
           "params": flat.data.clone(), "sums": sums.tolist()}
    torch.save(res, f"{out_path}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
