"""Per-call-site kernel statistics from a rocprofv3 --kernel-trace CSV of one bench.py step.

The wide route launches the same gemm256 instantiations from many call sites (RF w12 / grouped expert gate-up / semantic-decoder w12
are all gemm256_kernel<SWIGLU_SPLIT, hi/lo>), so the per-kernel --stats table mixes them.  The trace carries
the grid of every dispatch, and every call site has its own grid, so (kernel, grid) separates them.
usage: site_stats.py <kernel_trace.csv> [rows=1536] [out.csv]"""
import csv, re, sys
from collections import defaultdict

path = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
out_path = sys.argv[3] if len(sys.argv) > 3 else None
cd = lambda a, b: -(-a // b)
W, HID, A, H, STEPS = 3072, 8192, 12 * 3 * 3072 + 2 * 3072, 2048, 16
mt = cd(rows, 128)
# (tiles, split-K slices) -> call site, for the 16B-A3B / RF w=3072 shapes of bench.py at `rows` CFG rows in one group
SITES = {
    ("E4", mt * (HID // 128), 1): "RF w12 (SwiGLU + split epilogue)",
    ("E0", cd(STEPS * rows, 128) * cd(A, 256), 1): "RF adaLN, all Euler steps",
    ("E4", cd(rows // 2, 128) * cd(2752, 128), 1): "semantic decoder w12 (per image)",
    # the labelled fp8-MFMA regime (gemm256_kernel<.., false, true>: 256-row tiles, 128-k K-tiles)
    ("F6", cd(rows, 256) * (HID // 128), 1): "fp8-MFMA: RF w12 (SwiGLU epilogue)",
    ("F0", cd(STEPS * rows, 256) * cd(A, 256), 1): "fp8-MFMA: RF adaLN, all Euler steps",
}
for ks in range(2, 9):
    SITES[("E0", mt * cd(W, 256), ks)] = "RF w3 (split-K slabs)"
    SITES[("F0", cd(rows, 256) * cd(W, 256), ks)] = "fp8-MFMA: RF w3 (split-K slabs)"


def site_of(name, gx, gy):
    m = re.match(r"gemm256_kernel<(\d+), (true|false)(?:, (true|false))?>", name)
    if not m:
        return ""
    return SITES.get((("F" if m.group(3) == "true" else "E") + m.group(1), gx, gy), "")


def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(anonymous namespace\)::", "", nm)
    return re.sub(r"\(.*", "", nm)


acc = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in csv.DictReader(open(path)):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc[k]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in acc.values())
lines = []
for (name, gx, gy, gz), (n, t, lo, hi) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    site = site_of(name, gx, gy)
    lines.append((name, gx, gy, gz, n, t / n, lo, hi, t / 1e3, 100 * t / tot, site))
hdr = "kernel,workgroups_x,grid_y,grid_z,calls,avg_us,min_us,max_us,total_ms,percent,site"
if out_path:
    with open(out_path, "w") as f:
        f.write(hdr + "\n")
        for l in lines:
            f.write('"%s",%d,%d,%d,%d,%.2f,%.2f,%.2f,%.2f,%.2f,"%s"\n' % l)
print("total kernel time %.1f ms over %d distinct (kernel, grid) call shapes" % (tot / 1e3, len(lines)))
for l in lines[:40]:
    print("%-52s wg=%-7d y=%-3d z=%-3d %7d calls  avg %8.1f us  [%7.1f .. %8.1f]  %9.1f ms %5.1f%%  %s" % ((l[0][:52],) + l[1:]))
