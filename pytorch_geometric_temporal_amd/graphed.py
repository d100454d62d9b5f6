"""Whole-step hipGraph capture for the launch-bound regime.

A DCRNN training step at the reference's batch size (64 windows, examples/indexBatching/DCRNN/pems_bay_main.py:130-138)
is ~250 kernel launches of a few microseconds each; issued one by one from Python (ctypes + autograd bookkeeping,
~15-30 us of host time per launch) the GPU idles most of the step.  Every entry point of the C ABI is stream-ordered and
allocation-free, so the whole step — forward, loss, hand-written BPTT, optimizer update — can be recorded once into a
hipGraph and replayed with one host call per step.  This is the MI355X replacement for what the reference would get from
a tracing compiler: no tracing, no code generation, the same kernels in the same order.

    step = GraphedStep(fn, example_inputs)     # fn(*inputs) -> tensor or tuple of tensors; runs `warmup` times eagerly
    out = step(*inputs)                        # copies the inputs into the captured buffers, replays, returns the
                                               # captured output tensors (overwritten by the next call)

What comes back from a call are DETACHED views of the captured outputs: the step ran its own backward pass inside the graph, and
a loss that kept its grad_fn would keep the captured step's whole autograd graph — and through it every parameter's AccumulateGrad
node, bound to the stream of this capture — alive for as long as the GraphedStep lives.  A second capture would then run its
gradient accumulation across two streams inside the capture (torch warns: "may ... break CUDA graph capture"); on this stack that
ended in a GPU memory access fault, every time (round 5: bench_configs.covid_epoch's two captures, 20 of 20 runs; none of 20 once the
first capture's outputs were detached — profiles/r05b_covid_graph_inputs_runs.txt).  All captures of a device also share ONE side
stream.

Rules (hipGraph capture): `fn` must not synchronise with the host (no .item(), no shape-changing data dependence) and
must do the same work on every call; tensors it closes over (parameters, optimizer state, graph operators, the resident
series) keep their addresses.  Graph operators are prepared (and validated, which does sync) during the eager warm-up
calls and come out of ops.GRAPH_CACHE afterwards.  Optimizers must be constructed `capturable=True`.
"""
import torch

from . import ops


def _detached(out):
    """The captured outputs without their autograd graph (same storage: a replay refreshes the values)."""
    if isinstance(out, torch.Tensor):
        return out.detach()
    if isinstance(out, (tuple, list)):
        return type(out)(_detached(o) for o in out)
    if isinstance(out, dict):
        return {k: _detached(v) for k, v in out.items()}
    return out


class GraphedStep:
    _side_streams = {}                 # one capture stream per device for every GraphedStep of the process

    def __init__(self, fn, example_inputs, warmup=3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a HIP device (hipGraph capture)")
        self.fn = fn
        self.static_inputs = [t.clone() if isinstance(t, torch.Tensor) else t for t in example_inputs]
        dev = device
        if dev is None:
            dev = next((t.device for t in self.static_inputs if isinstance(t, torch.Tensor)),
                       torch.device("cuda", torch.cuda.current_device()))
        self.device = dev
        if ops.KERNEL_TIMER is not None:
            raise RuntimeError("GraphedStep: per-launch timing (ops.KERNEL_TIMER) cannot be captured")
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        side = GraphedStep._side_streams.get(key)
        if side is None:
            side = GraphedStep._side_streams[key] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):          # graph preparation, allocator growth, code-object loads
                fn(*self.static_inputs)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            out = fn(*self.static_inputs)
        self.static_outputs = _detached(out)
        del out                                      # (with it goes the captured step's autograd graph)
        torch.cuda.synchronize(dev)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"expected {len(self.static_inputs)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static_inputs, inputs):
            if isinstance(dst, torch.Tensor):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
            elif dst != src:
                raise ValueError("non-tensor inputs are baked into the captured graph and cannot change")
        self.graph.replay()
        return self.static_outputs
