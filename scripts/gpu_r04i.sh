#!/bin/bash
# Round 4, trip i: the renumbered ELLW layout (tests, probe under rocprofv3 stats + PMC), the default bench, config 4 after packed_once.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
O=$PWD/gpurun_out
(timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -m gpu -q -k "ellw or renumbered or north_star or packed_once or tgcn" 2>&1 | tail -15) > $O/pytest_gpu_sel.log
tail -3 $O/pytest_gpu_sel.log
(timeout 200 python scripts/ns_renumber_probe.py) > $O/ns_renumber_probe.jsonl 2> $O/ns_renumber_probe.err; echo "probe rc=$?"; cat $O/ns_renumber_probe.jsonl; tail -3 $O/ns_renumber_probe.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/r04i_ns_stats -- python $OLDPWD/scripts/ns_renumber_probe.py 12) > $O/prof/r04i_ns_stats.log 2>&1; echo "stats rc=$?"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/r04i_ns_fetch -- python $OLDPWD/scripts/ns_renumber_probe.py 12) > $O/prof/r04i_ns_fetch.log 2>&1; echo "fetch rc=$?"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/r04i_ns_write -- python $OLDPWD/scripts/ns_renumber_probe.py 12) > $O/prof/r04i_ns_write.log 2>&1; echo "write rc=$?"
find $O/prof -name "*kernel_trace.csv" -size +8M -delete
python - <<'PY'
import csv, glob, os, collections, re
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
for tag, ctr in (("r04i_ns_fetch", "FETCH_SIZE"), ("r04i_ns_write", "WRITE_SIZE")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(O, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "spmm" not in k or row["Counter_Name"] != ctr:
                continue
            m = re.search(r"(spmm_\w+?_kernel<[^>]*>)", k)
            name = m.group(1) if m else k[:60]
            acc[name][0] += 1
            acc[name][1] += float(row["Counter_Value"])
    for name, (n, v) in sorted(acc.items()):
        print(tag, ctr, name, n, "mean", v / n)
for f in glob.glob(os.path.join(O, "r04i_ns_stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "spmm" in row["Name"]:
            print("stats", row["Name"][:90], row["Calls"], row["AverageNs"])
PY
SECONDS=0
(timeout 420 python bench.py) > $O/bench.json 2> $O/bench.err
echo "bench rc=$? wall=${SECONDS}s"; head -c 200 $O/bench.json; echo; grep "^\[bench" $O/bench.err | tail -12
(timeout 200 python bench.py --config tgcn50k) > $O/bench_tgcn.json 2> $O/bench_tgcn.err
echo "tgcn bench rc=$?"; head -c 260 $O/bench_tgcn.json; echo
