"""Print a rocprofv3 kernel_stats.csv compactly: calls, average us, total ms per kernel (names shortened).
usage: kstats.py <csv> [filter substring ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2:]
for r in rows:
    n = r["Name"].replace("void ", "")
    short = n.split("(")[0][:70]
    if "rocprim" in n:
        short = "rocprim:" + n.split("wrapped_")[-1].split("<")[0][:40] if "wrapped_" in n else "rocprim:" + n.split("detail::")[-1][:40]
    if flt and not any(f in n for f in flt):
        continue
    print(f"{short:72s} x{int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
