#!/bin/bash
# Short gpurun call: selected GPU tests + selected probes.  TESTS="-k expr" PROBES="copy spmm_ns" scripts/gpu_quick.sh
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
O=gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  (timeout 900 python -m pytest tests -m gpu -q -x ${TESTS:-} 2>&1 | tail -15) > $O/pytest_gpu.log
  tail -4 $O/pytest_gpu.log
fi
if [ -n "${PROBES:-}" ]; then
  (timeout ${PROBE_TIMEOUT:-900} python scripts/gpu_probe.py ${PROBES}) > $O/probe.jsonl 2> $O/probe.err
  echo "probe rc=$?"
  tail -5 $O/probe.err
fi
