#!/usr/bin/env python
"""Audit of csrc/gemm_bx.hip's hand-counted waits on the compiled ISA (no GPU needed).

The kernels issue their block loads with inline asm and wait for them with inline-asm `s_waitcnt vmcnt(N)` statements
tied to the destination registers ("+v").  The compiler does not know those registers are in flight, so two things must
hold in the emitted code, in program order, for every kernel:
  1. on EVERY control-flow path between an asm load and an instruction that READS one of its destination registers there
     is a hand-written `s_waitcnt vmcnt` (or a compiler `s_waitcnt vmcnt(0)`) — the in-flight set is propagated over the
     basic-block graph to a fixed point;
  2. no scratch (spill) instruction exists where a hand-issued load can be in flight — a spilled in-flight register would be
     stored before it has landed, and scratch accesses would change the vmcnt arithmetic;
  3. every hand-issued store of more than 8 bytes is followed, inside its asm statement, by `s_nop` (the store reads its
     data registers over several cycles; the compiler pads its own wide stores against a VALU write in the next two issue
     slots and cannot see into the asm — round 3 shipped corrupted quads to the GPU tests before this rule).
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
    if len(sys.argv) > 1:                         # an ISA listing made elsewhere (a kernel kept for the record, a lab variant)
        text = open(sys.argv[1]).read()
        if ".Lfunc_end" not in text:
            text += "\n.Lfunc_end:\n"
    else:
      with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "gemm_bx.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), "-S", "--cuda-device-only", SRC,
                        "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    bad = 0
    for m in re.finditer(r"^(_ZN\S*gemm_bx\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        # ---- basic blocks (the compiler places loop blocks wherever it likes: text order says nothing about what runs
        # between a load and a read, so the in-flight set is propagated over the control-flow graph to a fixed point)
        blocks, cur, in_asm = [], {"label": None, "ins": [], "succ": [], "fall": True}, False
        for ln in body.split("\n"):
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            lab = re.match(r"^(\.LBB\d+_\d+):", s)
            if lab:
                blocks.append(cur)
                cur = {"label": lab.group(1), "ins": [], "succ": [], "fall": True}
                continue
            if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                continue
            op, _, rest = s.partition(" ")
            cur["ins"].append((op, rest.strip(), in_asm, s))
            if op == "s_branch":
                cur["succ"].append(rest.strip()); cur["fall"] = False
                blocks.append(cur); cur = {"label": None, "ins": [], "succ": [], "fall": True}
            elif op.startswith("s_cbranch"):
                cur["succ"].append(rest.strip())
                blocks.append(cur); cur = {"label": None, "ins": [], "succ": [], "fall": True}
            elif op == "s_endpgm":
                cur["fall"] = False
                blocks.append(cur); cur = {"label": None, "ins": [], "succ": [], "fall": True}
        blocks.append(cur)
        index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
        succ = []
        for i, b in enumerate(blocks):
            out = [index[t] for t in b["succ"] if t in index]
            if b["fall"] and i + 1 < len(blocks):
                out.append(i + 1)
            succ.append(out)

        # The in-flight state is the ORDERED list of hand-issued loads (oldest first, one register set each): a hand-written
        # `s_waitcnt vmcnt(N)` retires all but the N youngest — not everything.  (Until round 5 any hand-written wait cleared the
        # whole set, and a kernel whose compiler-made register copies at the loop's back edge read loads issued a few
        # instructions earlier passed the audit; it failed on the GPU.)  States meeting at a block are merged position by
        # position from the youngest end.
        def transfer(b, pend, report):
            pend = [set(x) for x in pend]
            found = []
            for op, rest, in_asm, s in b["ins"]:
                if op == "s_waitcnt" and "vmcnt" in rest:
                    n = int(re.search(r"vmcnt\((\d+)\)", rest).group(1))
                    if n == 0:
                        pend = []
                    elif in_asm:
                        pend = pend[len(pend) - n:] if n < len(pend) else pend
                    continue
                operands = rest.split(",")
                if in_asm and op.startswith("buffer_load"):
                    pend.append(set(regs(operands[0])))
                    continue
                if op.startswith(("buffer_store", "global_store", "global_atomic", "buffer_atomic", "global_load", "buffer_load")):
                    pend.append(set())              # any other vector-memory instruction counts in vmcnt as well
                start = 0 if op.startswith(("buffer_store", "global_store", "ds_write", "global_atomic", "v_cmp", "s_")) else 1
                srcs = set()
                for tok in operands[start:]:
                    srcs |= regs(tok)
                flying = set().union(*pend) if pend else set()
                hit = srcs & flying
                if hit and report:
                    found.append((sorted(hit), s))
                if start == 1:                      # an overwritten register is no longer the load's
                    dst = regs(operands[0])
                    for x in pend:
                        x -= dst
            return tuple(frozenset(x) for x in pend), found

        def merge(a, b):
            n = max(len(a), len(b))
            pa, pb = (frozenset(),) * (n - len(a)) + tuple(a), (frozenset(),) * (n - len(b)) + tuple(b)
            return tuple(x | y for x, y in zip(pa, pb))

        ins = [() for _ in blocks]
        changed, rounds = True, 0
        while changed and rounds < 200:
            changed, rounds = False, rounds + 1
            for i, b in enumerate(blocks):
                out, _ = transfer(b, ins[i], False)
                out = out[-64:]                     # (vmcnt is six bits: nothing older than 64 instructions is in flight)
                for j in succ[i]:
                    m = merge(ins[j], out)
                    if m != ins[j]:
                        ins[j] = m
                        changed = True
        violations, n_loads, n_waits, scratch = 0, 0, 0, 0
        for b in blocks:                            # rule 3
            for k, (op, rest, in_asm, s) in enumerate(b["ins"]):
                if in_asm and re.match(r"buffer_store_dwordx[34]", op):
                    nxt = b["ins"][k + 1] if k + 1 < len(b["ins"]) else None
                    if not (nxt and nxt[2] and nxt[0] == "s_nop"):
                        violations += 1
                        print(f"  {name[:60]}: wide asm store without s_nop: `{s}`")
        for i, b in enumerate(blocks):
            _, found = transfer(b, ins[i], True)
            for hit, s in found:
                violations += 1
                if violations <= 3:
                    print(f"  {name[:60]}: reads in-flight v{hit} in `{s}`")
            for op, rest, in_asm, s in b["ins"]:
                n_loads += in_asm and op.startswith("buffer_load")
                n_waits += in_asm and op == "s_waitcnt" and "vmcnt" in rest
            # rule 2, exactly: a scratch access is a violation where a hand-issued load can be in flight (it would change the
            # vmcnt arithmetic, or store a register that has not landed).  Spills on paths that never issue such a load — the
            # consumer wavefronts' epilogue of gemm_bx_tn_pc_kernel — are slow, not wrong.
            pend = [set(x) for x in ins[i]]
            for op, rest, in_asm, s in b["ins"]:
                if "scratch_" in op and any(pend):
                    scratch += 1
                if op == "s_waitcnt" and "vmcnt" in rest:
                    n = int(re.search(r"vmcnt\((\d+)\)", rest).group(1))
                    pend = [] if n == 0 else (pend[len(pend) - n:] if in_asm and n < len(pend) else pend)
                elif in_asm and op.startswith("buffer_load"):
                    pend.append(set(regs(rest.split(",")[0])))
        print(f"{name[22:72]:52s} asm loads {n_loads:3d}  hand waits {n_waits:3d}  early reads {violations}  scratch {scratch}")
        bad += violations + scratch
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
