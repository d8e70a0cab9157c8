"""Busy / idle time of the GPU over the last <window_ms> of a rocprofv3 kernel_trace.csv: union of all kernel intervals, the longest
gaps with the kernels on either side.  usage: gpu_idle.py <kernel_trace.csv> [window_ms] [n gaps]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
ngap = int(sys.argv[3]) if len(sys.argv) > 3 else 12
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win]
t0 = int(rows[0]["Start_Timestamp"])
busy, cur_s, cur_e, last_name, gaps = 0, None, None, "", []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_s is None: cur_s, cur_e, last_name = s, e, r["Kernel_Name"]; continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, (cur_e - t0) / 1e3, last_name[:50], r["Kernel_Name"][:50]))
        cur_s, cur_e = s, e
    else:
        if e > cur_e: cur_e = e
    if e >= cur_e: last_name = r["Kernel_Name"]
busy += cur_e - cur_s
span = cur_e - t0
print(f"window {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %), {len(gaps)} gaps")
for g in sorted(gaps, reverse=True)[:ngap]:
    print(f"  gap {g[0] / 1e3:7.1f} us at {g[1]:9.1f} us  after [{g[2]}]  before [{g[3]}]")
