"""Data-parallel path (SURVEY.md §8 e) under gloo with world_size 2 on CPU: two ranks on disjoint index shards +
one flat gradient all-reduce must reproduce the single-process step on the union of the shards."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from pytorch_geometric_temporal_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(model, world, out_path):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), model, out_path],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(out_path)


@pytest.mark.parametrize("model", ["dcrnn", "tgcn", "tgcn_seq"])
def test_two_rank_step_equals_single_process_step(tmp_path, model):
    one = _run(model, 1, str(tmp_path / "w1.pt"))
    two = _run(model, 2, str(tmp_path / "w2.pt"))
    assert two["world"] == 2
    for k in one["params"]:
        # mean of the two half-batch gradients == gradient of the full-batch mean loss (equal shard sizes)
        assert torch.allclose(one["params"][k], two["params"][k], atol=2e-6, rtol=1e-5), k
    assert one["losses"] == pytest.approx(two["losses"], rel=1e-5, abs=1e-6)


def test_torch_ddp_wrapper_as_the_reference_uses_it_equals_flat_gradients(tmp_path):
    """examples/indexBatching/DCRNN/pems_ddp.py:80-85 wraps the model in torch's DistributedDataParallel(gradient_as_bucket_view=
    True).  BatchedDCRNN + torch.nn.Linear (whose input is the Tensor subclass BatchedDCRNN returns) under that wrapper at world
    2 must train exactly like this package's one-flat-all-reduce path at world 2 — and like one process on the union."""
    flat1 = _run("ddp_flat", 1, str(tmp_path / "f1.pt"))
    flat2 = _run("ddp_flat", 2, str(tmp_path / "f2.pt"))
    ddp2 = _run("ddp", 2, str(tmp_path / "d2.pt"))
    ddp1 = _run("ddp", 1, str(tmp_path / "d1.pt"))
    for k in flat1["params"]:
        assert torch.allclose(ddp2["params"][k], flat2["params"][k], atol=2e-6, rtol=1e-5), k
        assert torch.allclose(ddp2["params"][k], flat1["params"][k], atol=2e-6, rtol=1e-5), k
        assert torch.allclose(ddp1["params"][k], flat1["params"][k], atol=2e-6, rtol=1e-5), k
    assert ddp2["losses"] == pytest.approx(flat2["losses"], rel=1e-5, abs=1e-6)


def test_torch_ddp_around_the_tgcn_sequence_loop(tmp_path):
    """The T-step loop over TGCN2 packs the cell's folded operands ONCE per training step (nn/_states.py: packed_once), so every
    parameter receives its gradient once, from one adjoint launch, at the end of the backward pass — under torch's
    DistributedDataParallel (bucket hooks on the parameters' gradient accumulators) that must train exactly like the one
    flat all-reduce, at world 2."""
    flat2 = _run("tgcn_seq", 2, str(tmp_path / "f2.pt"))
    ddp2 = _run("tgcn_ddp", 2, str(tmp_path / "d2.pt"))
    for k in flat2["params"]:
        assert torch.allclose(ddp2["params"][k], flat2["params"][k], atol=2e-6, rtol=1e-5), k
    assert ddp2["losses"] == pytest.approx(flat2["losses"], rel=1e-5, abs=1e-6)


@pytest.mark.parametrize("world,total", [(2, 10), (4, 10), (8, 21)])
def test_bench_protocol_under_gloo_with_uneven_shards(tmp_path, world, total):
    """bench.py's multi-rank protocol (dp.timed_steps + async flat all-reduce + DistributedSampler-style padding) with
    world sizes 2 / 4 / 8 and sample counts the ranks do not divide: every rank runs exactly the same steps, reports the
    same (MAX-reduced) time — which contains the slow rank's sleeps — and ends with identical parameters."""
    port = _free_port()
    out = str(tmp_path / "proto")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_protocol_worker.py"), out, str(total)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    per = -(-total // world)
    for r in res:
        assert r["world"] == world and r["calls"] == [2, 3, 4, 5, 6]
        assert r["shard"] == per                                   # padded by wrap-around to equal shard sizes
        assert r["dt"] == res[0]["dt"] and r["dt"] >= 3 * 0.02     # one number for the job: the slowest rank's
        assert torch.equal(r["params"], res[0]["params"])
    assert res[0]["sums"][1] == float(world)                       # reduce_scalars: SUM onto rank 0


def test_shard_indices_follow_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(23))
    for world in (1, 2, 4, 8):
        for epoch in (0, 3):
            got = []
            for rank in range(world):
                s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=5)
                s.set_epoch(epoch)
                mine = dp.shard_indices(len(ds), rank, world, epoch=epoch, shuffle=True, seed=5)
                assert mine.tolist() == list(iter(s))
                got += mine.tolist()
            assert set(got) == set(ds)          # every sample is covered (padding only repeats)
    assert dp.shard_indices(10, 1, 2, shuffle=False).tolist() == [1, 3, 5, 7, 9]


def test_flat_gradients_are_views_of_one_buffer():
    m = torch.nn.Linear(3, 2)
    flat = dp.FlatGradients(m.parameters())
    m(torch.ones(4, 3)).sum().backward()
    assert flat.flat.numel() == 8 and float(flat.flat.abs().sum()) > 0
    assert m.weight.grad.data_ptr() == flat.flat.data_ptr()
    flat.zero()
    assert float(m.bias.grad.abs().sum()) == 0.0
    assert flat.all_reduce_mean(1) is None       # world 1: no collective


def test_flat_parameters_adam_equals_per_parameter_adam():
    """One Adam update over the flat parameter buffer == torch.optim.Adam over the individual parameters, and the
    module keeps working (views), including load_state_dict."""
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 2))
    b = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 2))
    b.load_state_dict(a.state_dict())
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-2)
    flat = dp.FlatParameters(b.parameters())
    opt_b = flat.optimizer(torch.optim.Adam, lr=1e-2)
    assert flat.data.numel() == sum(p.numel() for p in a.parameters())
    assert b[0].weight.data_ptr() == flat.data.data_ptr()
    x, y = torch.randn(16, 5), torch.randn(16, 2)
    for _ in range(5):
        opt_a.zero_grad()
        (a(x) - y).pow(2).mean().backward()
        opt_a.step()
        flat.zero()
        (b(x) - y).pow(2).mean().backward()
        opt_b.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-7)
    b.load_state_dict(a.state_dict())                      # copies into the views
    assert b[2].bias.data_ptr() == flat.data[-2:].data_ptr()
    torch.testing.assert_close(flat.data[-2:], a[2].bias.detach())


def test_bench_starts_its_own_ranks_when_no_launcher_is_in_the_environment():
    """`python bench.py --gpus 2` with no WORLD_SIZE set re-runs itself under torch.distributed.run (one process per GPU, rendezvous
    on 127.0.0.1) and rank 0 prints ONE line of the printed format with n_gpus = 2.  PGT_BENCH_LAUNCH_PROBE=1 stops the ranks after
    the rendezvous + one all-reduce (gloo, no GPU here); the same command without it is the 2-GPU bench."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PGT_BENCH_LAUNCH_PROBE="1", PGT_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    # nothing but the line: what the libraries under the ranks write to stdout (gloo announces its connections there) goes to stderr
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rank_sum"] == 3.0        # ranks 0 and 1 met: 1 + 2
    assert "torch.distributed.run" in r.stderr


@pytest.mark.gpu
def test_rccl_carries_the_flat_gradient_all_reduce_on_one_rank():
    """What one GPU can show of the RCCL path (8-GPU runs are the driver's): a world-1 "nccl" process group comes up on cuda:0, the
    flat gradient buffer of the benchmarked model goes through all_reduce_mean / reduce_scalars / timed_steps' MAX-reduce, and comes
    back unchanged — in a child process, so that the group's lifetime is the test's."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pytorch_geometric_temporal_amd import dp
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
m = torch.nn.Sequential(BatchedDCRNN(2, 64, 3), torch.nn.Linear(64, 2)).cuda()
flat = dp.FlatParameters(m.parameters())
flat.flat.copy_(torch.randn_like(flat.flat))
want = flat.flat.clone()
dist.all_reduce(flat.flat)                      # SUM over one rank
flat.all_reduce_mean(world=1)                   # (a no-op by construction at world 1)
torch.cuda.synchronize()
assert torch.equal(flat.flat, want)
assert float(dp.reduce_scalars([1.5, 2.0])[1]) == 2.0
dt, out = dp.timed_steps(lambda i: i, 0, 3, device="cuda:0")
assert out == 2 and dt >= 0.0
dist.barrier()
dist.destroy_process_group()
print("rccl world-1 ok", flat.flat.numel())
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(_free_port())], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
