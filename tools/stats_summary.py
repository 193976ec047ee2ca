"""Print the top rows of a rocprofv3 kernel_stats CSV with short kernel names."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:n]:
    nm = re.sub(r"^void ", "", r["Name"])
    nm = re.sub(r"\(anonymous namespace\)::", "", nm)
    nm = re.sub(r"\(.*", "", nm)[:58]
    print(f"{nm:58s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:9.1f} ms {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):6.2f}%")
