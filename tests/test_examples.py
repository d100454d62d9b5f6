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
    a = rm.main(["--model", name, "--epochs", "1", "--snapshots", "4"], device=emu_backend.device)
    b = rm.main(["--model", name, "--epochs", "3", "--snapshots", "4"], device=emu_backend.device)
    assert all(map(lambda v: v == v and abs(v) < 1e6, a + b))
    assert b[0] != a[0]                     # the update moved the parameters
