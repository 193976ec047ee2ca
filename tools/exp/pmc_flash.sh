#!/bin/bash
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for w in "hd64 1" "hd128x7 1"; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE"; do
tag=$(echo $set | cut -d' ' -f1)
wt=$(echo $w | tr ' ' '_')
rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/${wt}_$tag -- python3 tools/exp/prof_flash.py $w > gpurun_out/pmc/${wt}_$tag.log 2>&1
f=$(find gpurun_out/pmc/${wt}_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$w $tag" <<'P'
import csv,sys,collections
f=sys.argv[1]
if not f: print(sys.argv[2],"no file"); sys.exit()
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    if "flash" in k or "attn" in k:
        print(sys.argv[2], k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
P
rm -rf gpurun_out/pmc/${wt}_$tag
done; done
