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


LISTING = """
_ZN4test12gemm_bx_fakeEv: ; a two-deep software pipeline: two loads in flight over the back edge
	;;#ASMSTART
	buffer_load_dword v10, v1, s[0:3], 0 offen
	;;#ASMEND
	;;#ASMSTART
	buffer_load_dword v11, v1, s[0:3], 0 offen
	;;#ASMEND
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v20, v20, v10
	;;#ASMSTART
	buffer_load_dword v12, v1, s[0:3], 0 offen
	;;#ASMEND
	s_cmp_lt_i32 s4, s5
	s_cbranch_scc0 .LBB0_3
%COPY%
	s_branch .LBB0_1
.LBB0_3:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v20, v20, v11
	s_endpgm
.Lfunc_end0:
"""


@pytest.mark.parametrize("copy,bad", [("\tv_mov_b32_e32 v13, v10", False), ("\tv_mov_b32_e32 v13, v12", True)])
def test_audit_models_vmcnt_exactly(tmp_path, copy, bad):
    """`s_waitcnt vmcnt(1)` retires all but the YOUNGEST load: a copy of the register it released is fine, a copy of the load
    issued a moment ago is a read of a register still in flight — the case that reached the GPU in round 5 (compiler-made copies
    at a loop's back edge) and that the audit used to wave through because any hand-written wait cleared its whole set."""
    f = tmp_path / "listing.s"
    f.write_text(LISTING.replace("%COPY%", copy))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bx_isa_audit.py"), str(f)], capture_output=True, text=True, timeout=60)
    assert (res.returncode != 0) == bad, res.stdout + res.stderr
    assert ("reads in-flight" in res.stdout) == bad, res.stdout
