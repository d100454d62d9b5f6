"""Data-parallel fan-out of index batches: one process per GPU, gradients exchanged with ONE RCCL all-reduce per step.

The reference wraps the model in torch DDP over gloo, launched through Dask (examples/indexBatching/DCRNN/
pems_ddp.py:83-85, 204-207), and shards the window start indices with DistributedSampler (dataset/metr_la.py:220-228).
The whole model is 150 - 76 000 fp32 parameters (0.6 - 305 KB): the exchange is latency-bound, so instead of DDP's
bucket/hook machinery every gradient lives in one flat buffer and a step issues exactly one all-reduce over xGMI
(backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).  No graph partitioning: the graph and the [T, N, F]
series are replicated on every GPU ("GPU-index-batching").
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())   # bind BEFORE the communicator is created
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # (hosts whose driver only supports dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 in the launcher's environment
        # for RCCL; a library does not edit its user's environment — bench.py, a launcher, sets it for itself)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(num_samples, rank, world, epoch=0, shuffle=True, seed=0, drop_last=False):
    """torch.utils.data.DistributedSampler's index assignment (dataset/metr_la.py:220-228; `set_epoch`,
    pems_ddp.py:96): a seeded permutation per epoch, padded by wrap-around to a multiple of `world`, rank r takes
    positions r, r + world, ...  Returned as a LongTensor so a whole batch is gathered with one device index op."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(num_samples, generator=g)
    else:
        idx = torch.arange(num_samples)
    if drop_last:
        total = (num_samples // world) * world
        idx = idx[:total]
    else:
        total = -(-num_samples // world) * world
        pad = total - num_samples
        if pad:
            reps = -(-pad // max(num_samples, 1))
            idx = torch.cat([idx, idx.repeat(reps)[:pad]])
    return idx[rank:total:world]


class FlatGradients:
    """Every parameter's .grad is a view into one contiguous buffer -> one all-reduce per optimisation step.

    Clear gradients with `.zero()` (or `optimizer.zero_grad(set_to_none=False)`): `zero_grad(set_to_none=True)` — the
    torch default — replaces each .grad by None and silently detaches the views, after which the all-reduce would
    average a buffer no backward pass writes to."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, world=None, async_op=False):
        """SUM over ranks then 1/world (DDP's gradient averaging).  Returns the work handle when async_op (finish() waits
        and scales).  The whole gradient is 0.6 - 305 KB and final only when BPTT ends (the weight gradients of all T steps
        are one product at the end), so there is nothing to overlap it with: bench.py issues it synchronously."""
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if world <= 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
        self.flat.mul_(1.0 / world)
        return None

    def finish(self, work, world):
        if work is not None:
            work.wait()
            self.flat.mul_(1.0 / world)


class FlatParameters(FlatGradients):
    """Parameters AND gradients as views of two contiguous buffers: one all-reduce per step and ONE optimizer update
    over one tensor (Adam / SGD are elementwise, so updating the concatenation is the same arithmetic as updating
    each parameter; per-parameter options such as weight-decay groups need the ordinary per-parameter optimizer).
    Call after the module is on its final device: `.to()` / `.cuda()` would re-allocate the parameters individually.
    `load_state_dict` keeps working (it copies into the views in place)."""

    def __init__(self, params):
        super().__init__(params)
        self.data = torch.empty_like(self.flat)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.data[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.data[off:off + n].view_as(p)
                off += n
        self.master = torch.nn.Parameter(self.data, requires_grad=True)   # same storage as every p.data
        self.master.grad = self.flat

    def optimizer(self, cls=torch.optim.Adam, **kwargs):
        """An optimizer over the single flat parameter; `fused=True` is used when the installed torch accepts it for
        this device (one kernel per step instead of ~9 per parameter)."""
        if "fused" not in kwargs and self.data.is_cuda:
            try:
                return cls([self.master], fused=True, **kwargs)
            except (RuntimeError, TypeError, ValueError):
                pass
        return cls([self.master], **kwargs)

    def adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """torch.optim.Adam's update over the flat buffer on the library's own kernel (FlatAdam)."""
        return FlatAdam(self, lr, betas, eps, weight_decay)


class FlatAdam:
    """torch.optim.Adam's update (amsgrad off) over a FlatParameters buffer through ONE elementwise launch of the library
    (pgt_adam_f32; + a one-thread launch that advances the device-side step count, so `step()` is hipGraph-capturable as it is).
    torch's fused Adam gives a single 76 k-element tensor to two workgroups: 96 us per step at every batch size."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if flat.data.dtype != torch.float32:
            raise TypeError("FlatAdam: fp32 parameters only")
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.steps = torch.zeros(1, dtype=torch.float32, device=flat.data.device)

    def reset(self):
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps.zero_()

    def zero_grad(self, set_to_none=False):
        self.flat.zero()                 # (the gradients are views of one buffer: never set to None)

    def step(self):
        from . import _lib
        lib = _lib.get_lib()
        f = self.flat
        lib.call("pgt_adam_f32", _lib.ptr(f.data), _lib.ptr(f.flat), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), _lib.ptr(self.steps),
                 f.data.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, _lib.stream_of(lib, f.data))


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def reduce_scalars(values, dst=0):
    """Per-epoch metric reduction (pems_ddp.py:160-161): SUM of a small float vector onto rank `dst`."""
    t = values if isinstance(values, torch.Tensor) else torch.tensor(values, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        # RCCL ("nccl") only moves device tensors: stage the vector on this rank's GPU and hand back a host tensor
        if dist.get_backend() == "nccl" and not t.is_cuda:
            d = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.reduce(d, dst=dst, op=dist.ReduceOp.SUM)
            return d.cpu()
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t


def _sync(device):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step, first, count, device=None):
    """The timing protocol of bench.py: barrier + device synchronize, EXACTLY `count` calls step(first), ...,
    step(first + count - 1), barrier + synchronize; returns (the MAX over ranks of the elapsed seconds, the last step's
    return value).  Every rank gets the same number — the job is as slow as its slowest rank."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1:
        dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    out = None
    for i in range(first, first + count):
        out = step(i)
    if world > 1:
        dist.barrier()
    _sync(device)
    dt = time.perf_counter() - t0
    if world > 1:
        on_gpu = device is not None and torch.device(device).type == "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out
