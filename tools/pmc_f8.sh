#!/bin/bash
# PMC passes of the fp8-MFMA regime's dominant launch (gemm256_kernel<SWIGLU_BF16, F8> on RF w12) next to the hi/lo launch it replaces:
# SQ counters (MFMA busy, waits, LDS), fabric traffic (FETCH_SIZE x 2 per MI355X_MICROARCH.md, WRITE_SIZE), one --pmc set per pass.
# usage (under gpurun): bash tools/pmc_f8.sh [rows] > gpurun_out/r06_pmc_f8_w12.txt
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ROWS=${1:-1536}
mkdir -p gpurun_out/pmcf8
for w in f8w12 w12; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
tag=$(echo $set | cut -d' ' -f1)
rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcf8/${w}_$tag -- python3 tools/prof_gemm256.py $w $ROWS > gpurun_out/pmcf8/${w}_$tag.log 2>&1
f=$(find gpurun_out/pmcf8/${w}_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$w" <<'P'
import csv,sys,collections
f=sys.argv[1]
if not f: print(sys.argv[2],"no file"); sys.exit()
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:44]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    if "gemm256" in k:
        print(sys.argv[2], k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
P
rm -rf gpurun_out/pmcf8/${w}_$tag
done; done
rm -rf gpurun_out/pmcf8
