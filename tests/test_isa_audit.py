"""The hand-counted waits of csrc/gemm_bx.hip, checked on the compiled gfx950 ISA (scripts/bx_isa_audit.py): on every
control-flow path a hand-issued load is followed by a hand-written wait before its registers are read, nothing spills
(a spilled in-flight register would be stored before it lands), and every hand-issued store of more than 8 bytes carries
its `s_nop` (the store-data hazard the compiler cannot see inside inline asm).  Needs hipcc only (cross-compiles), no GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_split_bf16_kernels_pass_the_isa_audit():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bx_isa_audit.py")], capture_output=True, text=True,
                         timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if "asm loads" in ln]
    assert len(lines) >= 15, res.stdout + res.stderr          # every instantiation of the three kernel families was seen
    assert res.returncode == 0, res.stdout[-3000:]
    for ln in lines:
        assert "early reads 0  scratch 0" in ln, ln
