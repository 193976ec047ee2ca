#!/bin/bash
# C3 prefill (1024^2 image + 1 058-token prompt) under rocprofv3 --kernel-trace, reduced to per-call-site statistics.
# usage (under gpurun): bash tools/prof_c3_prefill.sh [tag] [fp32|bf16]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04_c3_prefill}
D=/tmp/prof_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/exp/c3_prefill.py ${2:-fp32} > gpurun_out/${TAG}.txt 2> gpurun_out/${TAG}.err
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
python3 tools/site_stats.py "$TRACE" 1058 gpurun_out/${TAG}_site_stats.csv > gpurun_out/${TAG}_site_stats.txt
rm -rf "$D"
