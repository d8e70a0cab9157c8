"""Timeline of the last <n> kernel launches of a rocprofv3 kernel_trace.csv: start offset, duration, queue, name.
usage: kernel_timeline.py <kernel_trace.csv> [n] [name filter to anchor the window start, e.g. k_points] [which occurrence from the end, 1 = last; or +k = k-th from the start] [launches to show before it]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
if len(sys.argv) > 3:
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
    arg = sys.argv[4] if len(sys.argv) > 4 else "1"
    if arg.startswith("+"):         # +k: the k-th occurrence from the start (1-based), a few launches before it included
        k = int(arg[1:])
        lead = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        rows = rows[max(0, idx[k - 1] - lead):idx[k - 1] + n] if len(idx) >= k else rows[-n:]
    else:
        back = int(arg)
        rows = rows[idx[-back]:idx[-back] + n] if len(idx) >= back else rows[-n:]
else:
    rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][:90]}")
