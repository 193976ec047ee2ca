#!/bin/bash
# BASELINE configs[1] (MingTok enc -> dec, 64 x 256^2, bf16 regime) under rocprofv3 --kernel-trace, reduced per call site
# (kernel, grid) like the bench's table.  usage (under gpurun): bash tools/prof_c2_sites.sh [tag]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
D=/tmp/prof_c2_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/prof_mingtok.py > gpurun_out/${TAG}_c2_under_rocprof.log 2>&1
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
STATS=$(find "$D" -name "*kernel_stats.csv" | head -1)
cp "$STATS" gpurun_out/${TAG}_c2_kernel_stats.csv
python3 tools/site_stats.py "$TRACE" 1536 gpurun_out/${TAG}_c2_site_stats.csv > gpurun_out/${TAG}_c2_site_stats.txt
rm -rf "$D"
