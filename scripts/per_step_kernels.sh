#!/bin/bash
# kernels per training step: rocprofv3 --kernel-trace --stats of bench.py at 2 and at 6 timed steps; the difference / 4 is one
# step's launches and time per kernel name (set-up, warm-up and instrumentation cancel out)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
mkdir -p $O
for k in 2 6; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps_$k -- python $OLDPWD/bench.py --steps $k --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra) > $O/steps_$k.log 2>&1
  find $O/steps_$k -name "*kernel_trace.csv" -delete
done
python - <<'PY'
import csv, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
def load(k):
    f = glob.glob(os.path.join(O, f"steps_{k}", "**", "*kernel_stats.csv"), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(2), load(6)
rows = []
for name in b:
    c0, t0 = a.get(name, (0, 0.0))
    c1, t1 = b[name]
    if c1 > c0:
        rows.append(((t1 - t0) / 4e3, (c1 - c0) / 4, name))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(os.path.join(O, "r03_per_step_kernels.csv"), "w") as out:
    out.write("us_per_step,launches_per_step,kernel\n")
    for us, n, name in rows:
        out.write(f"{us:.1f},{n:g},\"{name[:140]}\"\n")
print(f"total {tot:.0f} us per step in {sum(r[1] for r in rows):g} launches")
for us, n, name in rows[:45]:
    print(f"{us:9.1f} us {n:6g} x  {name[:110]}")
PY
