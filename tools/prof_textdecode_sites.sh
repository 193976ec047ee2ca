#!/bin/bash
# One-row text decode under rocprofv3 --kernel-trace, reduced to per-call-site statistics.
# usage (under gpurun): bash tools/prof_textdecode_sites.sh [tag] [bf16|fp8|int8]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04_textdecode}
D=/tmp/prof_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 tools/exp/text_decode_loop.py ${2:-bf16} > gpurun_out/${TAG}.txt 2> gpurun_out/${TAG}.err
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
python3 tools/site_stats.py "$TRACE" 1 gpurun_out/${TAG}_site_stats.csv > gpurun_out/${TAG}_site_stats.txt
rm -rf "$D"
