"""The tensor subclass the batched recurrent layers return (`BatchedDCRNN`, `TGCN2`): a plain tensor in every respect but one —
a skinny `torch.nn.Linear` read-out applied to the states (directly, or behind the `relu` the reference's own models put there)
runs on this package's streaming kernels instead of the BLAS library's pathological tile for 1 – 4 output features."""
import torch

from .. import ops


class _StatesTensor(torch.Tensor):
    """What `BatchedDCRNN.forward` returns: a plain tensor in every respect but one — the reference's examples feed the
    `[B, T, N, out]` states to a per-node read-out `torch.nn.Linear(out, 1 … 4)` (examples/indexBatching/DCRNN/*_main.py,
    examples/recurrent/dcrnn_example.py:24-31), and for 1 – 4 output features over millions of rows the BLAS library
    behind `F.linear` picks a pathological tile (≈ 5 ms per call at 2.5 M rows against 0.14 ms for one streaming pass:
    bench.py `variants.dropin_default`).  `F.linear(states, weight, bias)` with a skinny fp32 weight is therefore routed to
    this package's streaming kernels (same parameters, same arithmetic type, ordinary autograd) — also when a `relu` stands
    between the states and the read-out, as in the reference's own models; every other operation — and `F.linear` with any
    other operand, or under autocast / tracing / torch.compile — runs as usual and returns plain tensors.  Pickling stores a
    plain tensor.
    `BatchedDCRNN.readout_interception = False` hands out plain tensors instead."""

    _RELUS = (torch.nn.functional.relu, torch.relu, torch.Tensor.relu)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear and not kwargs and 2 <= len(args) <= 3 and _interception_allowed(args[0]):
            x, w = args[0], args[1]
            b = args[2] if len(args) == 3 else None
            if (type(x) is _StatesTensor and type(w) in (torch.Tensor, torch.nn.Parameter) and w.dim() == 2 and
                    1 <= w.size(0) <= 4 and w.size(1) == x.size(-1) and x.dtype == w.dtype == torch.float32 and
                    x.device == w.device and (b is None or (type(b) in (torch.Tensor, torch.nn.Parameter) and
                                                            b.dtype == torch.float32 and b.device == w.device))):
                from .conv import _rows_in_memory_order
                pre = getattr(x, "_pgt_pre", None)
                # (the fused pass recomputes the relu from the states: only while neither the states nor the relu's own result —
                # h.mul_(2), F.dropout(h, inplace=True) — were written since)
                if pre is not None and pre[0]._version == pre[1] and x._version == pre[2] and ops.readout_fits(x, w, b):
                    # linear(relu(states)): one pass each way over the PRE-relu states (csrc/readout.hip) — relu, the product, and in
                    # the adjoint relu's mask, the input gradient and the weight gradient together.  The relu tensor itself exists
                    # (it was computed when the caller asked for it) and is an ordinary operand for anything else done with it.
                    x2, restore = _rows_in_memory_order(pre[0].as_subclass(torch.Tensor))
                    if x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0:
                        return restore(ops.ReadoutFunction.apply(x2, w, b, True))
                x2, restore = _rows_in_memory_order(x.as_subclass(torch.Tensor))
                return restore(ops.linear(x2, w.t(), b))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        # the reference's own models put a relu between the recurrent layer and the read-out (examples/recurrent/
        # dcrnn_example.py:27-28, examples/indexBatching/tgcn/metr_la_main.py:43-44): relu(states) is still "the states" for the
        # one purpose of this class, so the read-out that follows is routed as well
        if func in cls._RELUS and not kwargs.get("inplace", False) and isinstance(out, torch.Tensor) and \
                type(args[0]) is _StatesTensor:
            res = out.as_subclass(_StatesTensor)
            inner = getattr(args[0], "_pgt_pre", None)
            if inner is None:
                # the states this is the relu of, their version then, and the version of the relu's own result
                res._pgt_pre = (args[0], args[0]._version, res._version)
            elif inner[0]._version == inner[1] and args[0]._version == inner[2]:
                # relu of an untouched relu: the inner states stay the reference point (relu is idempotent)
                res._pgt_pre = (inner[0], inner[1], res._version)
            return res
        return _plain(out)

    def __reduce_ex__(self, proto):
        # torch.save / pickle of a result stores a plain tensor: the subclass is a routing hint, not data
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


def _interception_allowed(x):
    """The routed read-out returns fp32 from the package's kernels: under autocast stock torch would return the autocast dtype,
    and a tracer / compiler should see the stock op — in those contexts the call is left alone."""
    dev = x.device.type if isinstance(x, torch.Tensor) else "cuda"
    if torch.is_autocast_enabled(dev) or torch.jit.is_tracing():
        return False
    comp = getattr(torch, "compiler", None)
    return not (comp is not None and comp.is_compiling())


def _plain(out):
    if type(out) is _StatesTensor:
        return out.as_subclass(torch.Tensor)
    if type(out) in (tuple, list):               # (torch.Size and other non-tensor results go back untouched)
        return type(out)(_plain(o) for o in out)
    return out


# ---- packed operands shared by the calls of one training step -------------------------------------------------------------------
import threading
import weakref

from torch.nn.modules.module import register_module_forward_hook, register_module_forward_pre_hook

_PACKED = weakref.WeakKeyDictionary()
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_scope = threading.local()        # .depth: nn.Module calls in flight on this thread; .epoch: outermost calls finished so far


_compiling = getattr(getattr(torch, "compiler", None), "is_compiling", lambda: False)


def _enter_module(module, args):
    if _compiling():                  # (a tracer sees no mutation of the thread-local: compiled code does not come through packed_once)
        return
    _scope.depth = getattr(_scope, "depth", 0) + 1


def _leave_module(module, args, output):
    if _compiling():
        return
    d = getattr(_scope, "depth", 1) - 1
    _scope.depth = d
    if d <= 0:
        _scope.depth = 0
        _scope.epoch = getattr(_scope, "epoch", 0) + 1


# The scope of "the same operands" is one OUTERMOST module call (the user's model: its T-step loop over the cell runs inside it).
# Two process-wide hooks count the nesting; they do nothing else — and they are installed by the first call of packed_once, i.e.
# only in a process that actually runs one of this package's gated cells (importing the package leaves torch.nn.Module's
# hook-free call path alone).  The outermost call during which they appear is seen as "no enclosing module": its cells write
# their operands' values per call, which is always correct.
_hooks = []


def _install_scope_hooks():
    if not _hooks:
        _hooks.append(register_module_forward_pre_hook(_enter_module))
        _hooks.append(register_module_forward_hook(_leave_module, always_call=True))


def reset_call_scope():
    """Forget the nesting count of this thread.  A module call abandoned by a BaseException that is not an Exception
    (KeyboardInterrupt in a notebook) never runs its forward hook: the count then stays above zero and the cells keep treating
    later calls as part of that one — correct for parameters changed in the ordinary ways (`_version`), blind to `p.data` writes.
    Call this after such an interrupt."""
    _scope.depth = 0
    _scope.epoch = getattr(_scope, "epoch", 0) + 1


def _in_backward():
    return _graph_task_id is not None and _graph_task_id() != -1


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def packed_once(module, params, build, repack=None):
    """`build()` = the stacked / folded operands of a gated cell from `params` (one launch, an autograd node).  A sequence loop
    calls the cell once per time step with the SAME parameters (examples/indexBatching/tgcn/metr_la_main.py:41-45, examples/
    recurrent/dcrnn_example.py:38-46): packing per call also means one adjoint launch and a dozen gradient-accumulation adds per
    call (144 five-microsecond adds per T = 12 training step of config 4).  The operands are therefore kept until the parameters
    change (`_version`) or a backward pass has run through them (a hook on the first operand), so a T-step loop packs once and
    autograd sums the T gradients at the packed level.

    What keeps this from ever serving stale weights: (1) within one outermost module call the operands are reused as they are;
    across outermost calls (a per-snapshot loop in a script, two forwards before a backward) the autograd node is reused but the
    VALUES are written again by `repack(params, packed)` — one pack launch, so a write that leaves `_version` alone
    (`p.data.copy_()`, an EMA swap) is picked up; (2) operands made outside a hipGraph capture are not used inside one and vice
    versa (the capture must hold its own pack launch and its own buffers); (3) only while gradients are being recorded: an
    inference call packs for itself."""
    _install_scope_hooks()
    if not (torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params)) or _in_backward():
        # inference: one cheap launch per call and nothing to accumulate — and no way to go stale.  A forward that runs INSIDE a
        # backward pass (torch.utils.checkpoint re-running a segment) gets operands of its own: the cached ones belong to the graph
        # being walked right now
        return build()
    key = (tuple((p.data_ptr(), p._version) if p is not None else None for p in params), _capturing())
    epoch = (getattr(_scope, "epoch", 0), getattr(_scope, "depth", 0) > 0)
    hit = _PACKED.get(module)
    if hit is not None and hit[0] == key and (repack is not None or hit[2] == epoch):
        if hit[2] != epoch or not epoch[1]:            # another outermost call (or no enclosing module at all): values anew
            repack(params, hit[1])
            _PACKED[module] = (key, hit[1], epoch)
        return hit[1]
    packed = build()
    _PACKED[module] = (key, packed, epoch)
    first = packed[0]
    if first.requires_grad:
        ref = weakref.ref(module)

        def spent(_grad, ref=ref):
            m = ref()
            if m is not None:
                _PACKED.pop(m, None)
        first.register_hook(spent)
    return packed
