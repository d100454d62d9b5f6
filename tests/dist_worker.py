"""Worker of tests/test_distributed.py: one data-parallel training step of BatchedDCRNN / TGCN2 under gloo, kernels on
the CPU test double.  Run as `python tests/dist_worker.py <model> <out.pt>` with RANK / WORLD_SIZE / MASTER_* set."""
import os
import sys
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import build_emu_library  # noqa: E402
from pytorch_geometric_temporal_amd import _lib, dp  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import TGCN2, BatchedDCRNN  # noqa: E402


def build(model_name):
    torch.manual_seed(1234)
    if model_name == "dcrnn":
        return BatchedDCRNN(2, 4, K=2)
    if model_name in ("ddp", "ddp_flat"):             # pems_ddp.py:80-85's model + a per-node torch.nn.Linear read-out (the
        class Net(torch.nn.Module):                   # returned states are a Tensor subclass: _StatesTensor must survive DDP)
            def __init__(self):
                super().__init__()
                self.rnn = BatchedDCRNN(2, 4, K=3)
                self.head = torch.nn.Linear(4, 2)

            def forward(self, X, ei, ew):
                return self.head(self.rnn(X, ei, ew))
        return Net()
    if model_name in ("tgcn_seq", "tgcn_ddp"):        # BASELINE configs[3]'s model: the T-step loop of bench_tgcn.BatchedTGCN
        import bench_tgcn
        return bench_tgcn.BatchedTGCN(2, 4, 2)
    return TGCN2(2, 4, batch_size=1)


def batch_loss(model_name, model, X, y, ei, ew):
    if model_name in ("ddp", "ddp_flat"):
        out = model(X, ei, ew)                       # [B, T, N, 2]
        return (out[..., 0] - y).abs().mean() + 0.1 * out[..., 1].square().mean()
    if model_name == "dcrnn":
        out = model(X, ei, ew)                       # [B, T, N, O]
        return (out.mean(dim=-1) - y).abs().mean()
    if model_name in ("tgcn_seq", "tgcn_ddp"):
        import bench_tgcn
        out = model(X.permute(0, 2, 3, 1), ei, ew)   # x [B, N, F, T] -> [B, T, N, 2]; the example's de-normalised masked MAE
        return bench_tgcn.masked_mae_loss(out[..., 0] * bench_tgcn.STD + bench_tgcn.MEAN, y * bench_tgcn.STD + bench_tgcn.MEAN)
    out = model(X[:, -1], ei, ew)                    # [B, N, O]
    return (out.mean(dim=-1) - y[:, -1]).abs().mean()


def main():
    model_name, out_path = sys.argv[1], sys.argv[2]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _lib._set_library_for_testing(_lib.PgtLib(build_emu_library()))
    rank, _, world = dp.init_from_env(backend="gloo")
    n, T, total = 12, 3, 8
    ei_np, ew_np = syn.sensor_graph(n, 60, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    series = torch.from_numpy(syn.traffic_series(40, n, seed=1))
    model = build(model_name)
    if rank != 0:                                      # ranks start different on purpose: broadcast must fix it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(model, src=0)
    ddp = None
    if model_name in ("ddp", "tgcn_ddp"):              # torch's own wrapper exactly as the reference uses it (pems_ddp.py:83-85)
        from torch.nn.parallel import DistributedDataParallel as DDP
        if world == 1:                                 # DDP wants a process group even for one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=0, world_size=1)
        ddp = DDP(model, gradient_as_bucket_view=True)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.1)
        flat = None
    elif model_name == "dcrnn":                          # flat parameters + one optimizer update (what bench.py does)
        flat = dp.FlatParameters(model.parameters())
        opt = flat.optimizer(torch.optim.SGD, lr=0.1)
    else:                                              # flat gradients + the ordinary per-parameter optimizer
        flat = dp.FlatGradients(model.parameters())
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
    losses = []
    for epoch in range(2):
        mine = dp.shard_indices(total, rank, world, epoch=epoch, shuffle=True, seed=7)
        ar = torch.arange(T)
        X = series[mine[:, None] + ar[None, :]]                      # [b, T, N, 2]
        y = series[mine[:, None] + T + ar[None, :]][..., 0]          # [b, T, N]
        loss = batch_loss(model_name, ddp if ddp is not None else model, X, y, ei, ew)
        if ddp is not None:
            opt.zero_grad()
            loss.backward()                            # DDP's bucket hooks average the gradients
        else:
            flat.zero()
            loss.backward()
            flat.all_reduce_mean(world)
        opt.step()
        losses.append(float(dp.reduce_scalars([float(loss)])[0]) / world)
    if rank == 0:
        torch.save({"params": {k: v.detach().clone() for k, v in model.state_dict().items()}, "losses": losses,
                    "world": world}, out_path)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
