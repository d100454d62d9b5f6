"""`pytorch_geometric_temporal_amd.ops` is a package of one module per model family; `ops.NAME` stays the public spelling and the
place where switches are flipped.  A write to an attribute of the package has to reach the family module whose code reads it."""
from pytorch_geometric_temporal_amd import ops
from pytorch_geometric_temporal_amd.ops import _core, _graphs, dcrnn, generic, stconv, tgcn


def test_a_switch_written_on_the_package_reaches_every_module_that_holds_it():
    assert dcrnn.USE_SEQ64 is ops.USE_SEQ64
    old = ops.USE_SEQ64
    try:
        ops.USE_SEQ64 = not old
        assert dcrnn.USE_SEQ64 == (not old) and ops.USE_SEQ64 == (not old)
    finally:
        ops.USE_SEQ64 = old
    assert dcrnn.USE_SEQ64 == old
    # a switch defined in one family module and imported by others: all copies follow
    sentinel = object()
    old = ops.KERNEL_TIMER
    try:
        ops.KERNEL_TIMER = sentinel
        assert all(m.KERNEL_TIMER is sentinel for m in (_core, dcrnn, generic, tgcn, stconv))
    finally:
        ops.KERNEL_TIMER = old
    assert all(m.KERNEL_TIMER is old for m in (_core, dcrnn, generic, tgcn, stconv))


def test_monkeypatching_a_function_on_the_package_is_seen_by_its_callers(monkeypatch):
    real = ops.slab_fits
    monkeypatch.setattr(ops, "slab_fits", lambda *a, **k: False)
    assert dcrnn.slab_fits(None, 66, 3) is False and ops.slab_fits(None, 66, 3) is False
    monkeypatch.undo()
    assert dcrnn.slab_fits is real and ops.slab_fits is real


def test_every_family_module_is_re_exported():
    for m in (_graphs, _core, dcrnn, generic, tgcn, stconv):
        for name, value in vars(m).items():
            if not (name.startswith("__") and name.endswith("__")):
                assert hasattr(ops, name), name
    assert ops.Csr is _graphs.Csr and ops.spmm is _core.spmm and ops.DCRNNSeq64Function is dcrnn.DCRNNSeq64Function
