#!/bin/bash
# tools/exp/moe_down_ab.py under rocprofv3 --kernel-trace, reduced to per-call-site statistics.  usage (under gpurun): bash tools/prof_moe_down.sh [tag]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05_moe_down}
D=/tmp/prof_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/exp/moe_down_ab.py > gpurun_out/${TAG}_ab.txt 2> gpurun_out/${TAG}_ab.err
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
python3 tools/site_stats.py "$TRACE" 2 gpurun_out/${TAG}_site_stats.csv > gpurun_out/${TAG}_site_stats.txt
rm -rf "$D"
