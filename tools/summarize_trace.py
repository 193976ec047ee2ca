"""Print a window of a rocprofv3 kernel trace CSV: per-dispatch duration and the gap to the previous dispatch.
usage: summarize_trace.py trace.csv [count] [start_fraction 0..1 | -1 = tail]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
frac = float(sys.argv[3]) if len(sys.argv) > 3 else -1
i0 = len(rows) - n if frac < 0 else int(len(rows) * frac)
def short(nm):
    m = re.search(r"(medium_\w+|stream_mfma_lds_kernel<[^>]*>|skinny_kernel<[^>]*>|gemm_bf16_kernel<[^>]*>|rf_\w+|moe_\w+|attn_\w+<?\d*>?|rope_\w+|semdec_\w+|\w+_kernel)", nm)
    return m.group(1) if m else nm[:30]
prev_end = None
tot = gap_tot = 0.0
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    tot += (e - s) / 1e3; gap_tot += gap
    print(f'{short(r["Kernel_Name"]):40s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid=({r["Grid_Size_X"]},{r["Grid_Size_Y"]},{r["Grid_Size_Z"]}) wg={r["Workgroup_Size_X"]}')
print(f"window: kernels {tot:.1f} us, gaps {gap_tot:.1f} us")
