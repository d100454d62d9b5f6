#!/bin/bash
# build products are shipped; this runs the seq64 probe + phase trace on the GPU box (the trace for the shipped source and for the
# withdrawn overlapped forward, same box)
cd "${GRAFT_REPO_ROOT:-.}"
python scripts/seq64_probe.py ${PROBE_B:-64 256 1024} 2>&1 | grep -v amdgpu.ids
for v in "" ${TRACE_LIBS:-_stagger}; do
  python scripts/seq64_trace.py ${TRACE_B:-256} 1515 $v 2>&1 | grep -v amdgpu.ids
done
