"""Print per-kernel durations of the last N dispatches of a rocprofv3 kernel trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
def short(nm):
    m = re.search(r"(medium_\w+|stream_mfma_lds_kernel|stream_mfma_kernel|skinny_kernel<[^>]*>|gemm_bf16_kernel<[^>]*>|rf_\w+|moe_\w+|attn_\w+<?\d*>?|rope_\w+)", nm)
    return m.group(1) if m else nm[:30]
for r in rows[-n:]:
    print(f'{short(r["Kernel_Name"]):34s} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid=({r["Grid_Size_X"]},{r["Grid_Size_Y"]}) wg={r["Workgroup_Size_X"]}')
