#!/bin/bash
# Rebuild the library if any kernel source changed (a stale .so makes the whole GPU trip fail at import), then gpurun a script.
#   scripts/gpu_go.sh <timeout> <script>
cd "$(dirname "$0")/.."
python -m pytorch_geometric_temporal_amd._build > /dev/null || exit 1
python - <<'PY' || exit 1
from pytorch_geometric_temporal_amd import _build
assert open(_build.BUILD_ID_PATH).read().strip() == _build.source_digest(), "stale library"
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
