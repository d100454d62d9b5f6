#!/usr/bin/env python
"""Audit of csrc/gemm_bx.hip's hand-counted waits on the compiled ISA (no GPU needed).

The kernels issue their block loads with inline asm and wait for them with inline-asm `s_waitcnt vmcnt(N)` statements
tied to the destination registers ("+v").  The compiler does not know those registers are in flight, so two things must
hold in the emitted code, in program order, for every kernel:
  1. between an asm load and the first instruction that READS one of its destination registers there is a
     hand-written `s_waitcnt vmcnt` (or a compiler `s_waitcnt vmcnt(0)`);
  2. no scratch (spill) instruction exists — a spilled in-flight register would be stored before it has landed, and
     scratch accesses would change the vmcnt arithmetic.
Usage: python scripts/bx_isa_audit.py   (compiles the file with hipcc -S; exit code 1 on a violation)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pytorch_geometric_temporal_amd", "csrc", "gemm_bx.hip")


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4..7}."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "gemm_bx.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), "-S", "--cuda-device-only", SRC,
                        "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    bad = 0
    for m in re.finditer(r"^(_ZN\S*gemm_bx\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        pending, in_asm, n_loads, n_waits, violations, scratch = {}, False, 0, 0, 0, 0
        for ln in body.split("\n"):
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            op, _, rest = s.partition(" ")
            if "scratch_" in op:
                scratch += 1
            if op == "s_waitcnt" and "vmcnt" in rest:
                if in_asm or "vmcnt(0)" in rest:
                    pending.clear()
                    n_waits += in_asm
                continue
            operands = rest.split(",")
            if in_asm and op.startswith("buffer_load"):
                n_loads += 1
                for r in regs(operands[0]):
                    pending[r] = True
                continue
            srcs = set()
            start = 0 if op.startswith(("buffer_store", "global_store", "ds_write", "global_atomic", "v_cmp", "s_")) else 1
            for tok in operands[start:]:
                srcs |= regs(tok)
            hit = srcs & set(pending)
            if hit:
                violations += 1
                if violations <= 3:
                    print(f"  {name[:60]}: reads in-flight v{sorted(hit)} in `{s}`")
            # a register that is overwritten is no longer the load's (the load result is dead)
            if start == 1:
                for r in regs(operands[0]):
                    pending.pop(r, None)
        print(f"{name[22:72]:52s} asm loads {n_loads:3d}  hand waits {n_waits:3d}  early reads {violations}  scratch {scratch}")
        bad += violations + scratch
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
