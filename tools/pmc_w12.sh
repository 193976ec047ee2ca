#!/bin/bash
# HBM traffic of the roofline kernel (gemm256 SwiGLU-split on RF w12) per launch: separate --pmc passes (FETCH_SIZE, WRITE_SIZE),
# FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md §HBM.  usage: bash tools/pmc_w12.sh ROWS
ROWS=${1:-1536}
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmcw
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcw/$c -- python3 tools/prof_gemm256.py w12 $ROWS > gpurun_out/pmcw/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmcw/stats -- python3 tools/prof_gemm256.py w12 $ROWS > gpurun_out/pmcw/stats.log 2>&1
python3 - $ROWS <<'P'
import csv, glob, json, sys
rows = int(sys.argv[1])
def avg(counter):
    f = glob.glob("gpurun_out/pmcw/%s/**/*counter_collection.csv" % counter, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm256" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
fetch, n1 = avg("FETCH_SIZE"); write, n2 = avg("WRITE_SIZE")
st = glob.glob("gpurun_out/pmcw/stats/**/*kernel_stats.csv", recursive=True)[0]
k = [r for r in csv.DictReader(open(st)) if "gemm256" in r["Name"]][0]
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 bytes?  calibrate on the known weight bytes below
out = {"kernel": k["Name"], "rows": rows, "launches": n1, "FETCH_SIZE_avg": fetch, "WRITE_SIZE_avg": write,
       "avg_us_under_rocprof": float(k["AverageNs"]) / 1e3,
       "weights_bytes": 2 * 8192 * 3072 * 2, "x_bytes": 2 * rows * 3072 * 2, "y_bytes": 2 * rows * 8192 * 2}
# counters are in units of 1 KiB on this rocprofv3 (FETCH_SIZE = TCC_EA0_RDREQ * 64 B / 1024); gfx950: double FETCH_SIZE
out["traffic_bytes_per_launch"] = (2 * fetch + write) * 1024
json.dump(out, open("gpurun_out/r02_pmc_gemm256_w12_rows%d.json" % rows, "w"), indent=1)
print(json.dumps(out))
P
cp $(ls gpurun_out/pmcw/stats/*/*kernel_stats.csv | head -1) gpurun_out/r02_w12_rows${ROWS}_kernel_stats.csv
rm -rf gpurun_out/pmcw
