#!/bin/bash
# One image at batch 1 (the reference's call shape: 2 CFG rows) under rocprofv3 --kernel-trace, reduced to per-call-site statistics.
# usage (under gpurun): bash tools/prof_batch1_sites.sh [tag] [extra bench.py flags, e.g. --weights fp8]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04_batch1}
shift || true
D=/tmp/prof_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 bench.py --images 1 --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 --no-other-configs "$@" > gpurun_out/${TAG}_line.json 2> gpurun_out/${TAG}.err
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
python3 tools/site_stats.py "$TRACE" 2 gpurun_out/${TAG}_site_stats.csv > gpurun_out/${TAG}_site_stats.txt
rm -rf "$D"
