#!/bin/bash
# rocprofv3 counter passes over a single-kernel probe (separate --pmc runs beside --kernel-trace only, as MI355X_MICROARCH.md
# prescribes): per-kernel means of every counter, printed and kept under gpurun_out/prof/${TAG}_*.
#   TAG=r05c scripts/pmc_probe.sh "scripts/tgcn_cell_probe.py 8 10"
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
TAG=${TAG:-r05}
CMD="$1"
mkdir -p $O
pass() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/${TAG}_$name -- python $OLDPWD/$CMD) > $O/${TAG}_$name.log 2>&1
  echo "pass $name rc=$?"
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
pass sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum
pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python - "$O" "$TAG" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{root}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(t in name for t in ("tgcn", "spmm", "gemm", "dconv", "relu_linear", "gru_", "seq64")):
            continue
        k = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = open(f"{root}/{tag}_pmc_summary.csv", "w")
out.write("kernel,counter,dispatches,mean\n")
for k, cs in sorted(acc.items()):
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"   {c:36s} {s / n:16.1f}  ({n} dispatches)")
        out.write(f"\"{k}\",{c},{n},{s / n:.1f}\n")
PY
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete
