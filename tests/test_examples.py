"""examples/recurrent_models.py on the CPU test double: every in-scope model of the reference's examples/recurrent/ scripts
trains for two short epochs through the drop-in modules (loss finite, parameters move)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


@pytest.mark.parametrize("name", ["dcrnn", "tgcn", "a3tgcn", "evolvegcnh", "evolvegcno", "gconvgru", "gconvlstm", "gclstm"])
def test_recurrent_model_examples_train(emu_backend, name):
    import recurrent_models as rm
    a = rm.main(["--model", name, "--epochs", "1", "--snapshots", "2"], device=emu_backend.device)
    b = rm.main(["--model", name, "--epochs", "2", "--snapshots", "2"], device=emu_backend.device)
    assert all(map(lambda v: v == v and abs(v) < 1e6, a + b))
    assert b[0] != a[0]                     # the update moved the parameters


def test_a3tgcn_index_batched_example_trains(emu_backend):
    """examples/a3tgcn_index_batched_synthetic.py (the reference's index-batched A3T-GCN loop) at a toy size: the loss goes down."""
    import a3tgcn_index_batched_synthetic as ex
    common = ["--nodes", "12", "--edges", "40", "--steps", "120", "--windows", "8", "--batch-size", "4"]
    first = ex.main(common + ["--epochs", "1"], device=emu_backend.device)
    later = ex.main(common + ["--epochs", "4"], device=emu_backend.device)
    assert first == first and later < first
