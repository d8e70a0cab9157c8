"""Top kernels of a rocprofv3 kernel_stats.csv.  usage: stats_top.py <csv> [divide totals by N] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} total_ms={float(r['TotalDurationNs']) / 1e6 / div:8.2f}")
print(f"all kernels: {tot / 1e6 / div:.2f} ms")
