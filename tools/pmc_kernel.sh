#!/bin/bash
# HBM / fabric traffic and duration of ONE kernel per launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2), then a --kernel-trace --stats pass, all over the same launch script.  FETCH_SIZE is doubled per
# the gfx950 correction of MI355X_MICROARCH.md §HBM (16-byte-per-lane streaming reads are tallied at half); counters are in KiB.
# usage (under gpurun): bash tools/pmc_kernel.sh OUT.json KERNEL_SUBSTRING ALGORITHMIC_BYTES SCRIPT [ARGS...]
# Writes gpurun_out/OUT.json; copy it to profiles/ and stamp it with tools/profile_meta.py (bench.py only reports stamped profiles).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; KERN=$2; ALG=$3; shift 3
D=/tmp/pmck_$$
rm -rf "$D"; mkdir -p "$D" gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$D/$c" -- python3 "$@" > "$D/$c.log" 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -- python3 "$@" > "$D/stats.log" 2>&1
python3 - "$D" "$KERN" "$ALG" "gpurun_out/$OUT" "$*" <<'P'
import csv, glob, json, sys
d, kern, alg, out, cmd = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
def avg(counter):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (d, counter), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)
fetch, n = avg("FETCH_SIZE")
write, _ = avg("WRITE_SIZE")
st = glob.glob("%s/stats/**/*kernel_stats.csv" % d, recursive=True)[0]
k = [r for r in csv.DictReader(open(st)) if kern in r["Name"]][0]
res = {"kernel": k["Name"], "launches": n, "command": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace / --kernel-trace --stats -- python3 " + cmd,
       "FETCH_SIZE_per_launch_KiB": fetch, "WRITE_SIZE_per_launch_KiB": write, "gfx950_fetch_correction": 2.0,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (2 * fetch + write) * 1024 / alg, "avg_us_rocprof_stats": float(k["AverageNs"]) / 1e3,
       "calls_rocprof_stats": int(k["Calls"])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
P
cp "$(find "$D/stats" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${OUT%.json}_kernel_stats.csv"
rm -rf "$D"
