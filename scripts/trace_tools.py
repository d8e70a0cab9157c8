"""Helpers over a rocprofv3 --kernel-trace CSV: per-call kernel breakdown and a timeline of one ingest call.
usage: trace_tools.py breakdown|timeline <kernel_trace.csv> [calls]"""
import collections
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def breakdown(rows, calls):
    kp = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_points")]
    start = kp[-calls]
    agg = collections.OrderedDict()
    for r in rows[start:]:
        a = agg.setdefault(r["Kernel_Name"][:60], [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = 0
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(n.ljust(60), str(c).rjust(4), f"{d / calls / 1e3:8.1f} us/call")
        tot += d
    print("sum per call us", tot / calls / 1e3)


def timeline(rows, marker="k_points"):
    kp = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker)]
    a, b = kp[-3], kp[-2]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        s = int(r["Start_Timestamp"]) - t0
        e = int(r["End_Timestamp"]) - t0
        print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:64]} grid={r.get('Grid_Size', '')}")


if __name__ == "__main__":
    rows = load(sys.argv[2])
    if sys.argv[1] == "breakdown":
        breakdown(rows, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    else:
        timeline(rows, sys.argv[3] if len(sys.argv) > 3 else "k_points")
