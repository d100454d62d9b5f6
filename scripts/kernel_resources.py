#!/usr/bin/env python
"""Per-kernel register / spill / LDS / occupancy table of one csrc/*.hip file from hipcc's resource-usage remarks (no GPU).
Usage: python scripts/kernel_resources.py dconv_slab.hip [name-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorch_geometric_temporal_amd", "csrc")


def main():
    src = os.path.join(CSRC, sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                          "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = None
    rows = []
    for ln in out.split("\n"):
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, lab in (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("VGPRs Spill", "spill"), ("SGPRs", "sgpr"),
                         ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds"), ("ScratchSize [bytes/lane]", "scratch")):
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", ln)
            if m:
                cur[lab] = int(m.group(1))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        if pat and pat not in name:
            continue
        print(f"{name:60s} vgpr {r.get('vgpr', -1):4d} agpr {r.get('agpr', 0):4d} spill {r.get('spill', 0):4d} scratch {r.get('scratch', 0):5d} "
              f"sgpr {r.get('sgpr', -1):4d} occ {r.get('occ', -1)} lds {r.get('lds', 0)}")


if __name__ == "__main__":
    main()
