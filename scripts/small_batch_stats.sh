#!/bin/bash
# per-step kernel split of the B = 64 / 128, hidden-64 training step: rocprofv3 --kernel-trace --stats over
# scripts/small_batch_probe.py at 10 and at 30 eager steps; the difference / 20 is one step (set-up and warm-up cancel)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
mkdir -p $O
for B in ${BATCHES:-64 128}; do
  python scripts/small_batch_probe.py $B 64 50 ${TUNE:-}
  for k in 10 30; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sb${B}_$k -- python $OLDPWD/scripts/small_batch_probe.py $B 64 $k eager ${TUNE:-}) > $O/sb${B}_$k.log 2>&1
    find $O/sb${B}_$k -name "*kernel_trace.csv" -delete
  done
  B=$B TAG=${TAG:-r06} python - <<'PY'
import csv, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
B = os.environ["B"]
def load(k):
    f = glob.glob(os.path.join(O, f"sb{B}_{k}", "**", "*kernel_stats.csv"), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(10), load(30)
rows = []
for name in b:
    c0, t0 = a.get(name, (0, 0.0))
    c1, t1 = b[name]
    if c1 > c0:
        rows.append(((t1 - t0) / 20e3, (c1 - c0) / 20, name))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(os.path.join(O, f"{os.environ['TAG']}_small_batch_B{B}_per_step_kernels.csv"), "w") as out:
    out.write("us_per_step,launches_per_step,us_per_launch,kernel\n")
    for us, n, name in rows:
        out.write(f"{us:.1f},{n:g},{us / n:.1f},\"{name[:140]}\"\n")
print(f"B = {B}: total {tot:.0f} us per step in {sum(r[1] for r in rows):g} launches")
for us, n, name in rows[:40]:
    print(f"{us:9.1f} us {n:6g} x {us / n:7.1f}  {name[:100]}")
PY
done
