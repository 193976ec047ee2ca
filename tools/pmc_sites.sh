#!/bin/bash
# Fabric traffic per call site: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over one fixed launch sequence, grouped by
# (kernel, grid) like tools/site_stats.py.  FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md §HBM; the
# counters are in KiB.  usage: bash tools/pmc_sites.sh OUT.txt SCRIPT [ARGS...]   (SCRIPT is run as `python3 SCRIPT ARGS`)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
rm -rf gpurun_out/pmcs; mkdir -p gpurun_out/pmcs
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcs/$c -- python3 "$@" > gpurun_out/pmcs/$c.log 2>&1
done
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": [], "ns": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmcs/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        name = r["Kernel_Name"]
        name = name[5:] if name.startswith("void ") else name
        key = (name.split("(G256")[0].split("(mn_")[0][:52], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))     # all workgroups of the grid
        acc[key][c].append(float(r["Counter_Value"]))
        if c == "FETCH_SIZE" and "End_Timestamp" in r: acc[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, v in acc.items():
    n = len(v["FETCH_SIZE"])
    if not n: continue
    fetch = 2 * sum(v["FETCH_SIZE"]) / n * 1024
    write = (sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))) * 1024
    us = sum(v["ns"]) / max(1, len(v["ns"])) / 1e3
    rows.append((fetch * n, k, n, fetch, write, us))
rows.sort(reverse=True)
with open(sys.argv[1], "w") as o:
    o.write("kernel, workgroups: launches, fabric read MB / launch (2 x FETCH_SIZE), write MB / launch, avg us under the counter pass\n")
    for _, k, n, fetch, write, us in rows[:40]:
        o.write("%-52s wg=%-7d %6d calls  read %9.1f MB  write %8.1f MB  %9.1f us\n" % (k[0], k[1], n, fetch / 1e6, write / 1e6, us))
print(open(sys.argv[1]).read())
P
rm -rf gpurun_out/pmcs
