#!/bin/bash
# One bench.py step under rocprofv3 --kernel-trace, reduced on the box to (a) the per-kernel --stats table and (b) per-call-site
# statistics (tools/site_stats.py: dispatches grouped by kernel AND grid).  usage (under gpurun): bash tools/prof_bench_sites.sh [tag]
# BENCH_EXTRA: extra bench.py flags (default --no-fp8-mfma: the headline step only; BENCH_EXTRA=" " traces the labelled fp8-MFMA leg too)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
D=/tmp/prof_$TAG
rm -rf "$D"; mkdir -p "$D" gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-other-configs ${BENCH_EXTRA:---no-fp8-mfma} > gpurun_out/${TAG}_bench_line_under_rocprof.json 2> gpurun_out/${TAG}_bench_under_rocprof.err
TRACE=$(find "$D" -name "*kernel_trace.csv" | head -1)
STATS=$(find "$D" -name "*kernel_stats.csv" | head -1)
cp "$STATS" gpurun_out/${TAG}_bench_default_kernel_stats.csv
python3 tools/site_stats.py "$TRACE" 1536 gpurun_out/${TAG}_bench_default_site_stats.csv > gpurun_out/${TAG}_bench_default_site_stats.txt
rm -rf "$D"
