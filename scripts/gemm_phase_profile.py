"""Per-workgroup phase stamps of k_gemm_split (-DBSC_GEMM_PROFILE build, BSC_GEMM_PROFILE_DUMP=1): reads the "GP ..." lines of a
log (the last dump in it), prints phase medians and, per CU, the gap between one workgroup's last store ack and the next one's start.
usage: gemm_phase_profile.py <log>"""
import sys, collections, statistics as st
rows = []
for line in open(sys.argv[1]):
    if not line.startswith("GP "): continue
    f = line.split()
    w = int(f[1])
    if w == 0: rows = []
    rows.append((w, int(f[2], 16)) + tuple(int(v) for v in f[3:8]))
rows = [r for r in rows if r[2] != 0]
t0 = min(r[2] for r in rows)
end = max(r[6] for r in rows)
print(f"workgroups that ran {len(rows)}, kernel span {(end - t0) / 100:.1f} us")
def med(v): return st.median(v) / 100
print(f"prologue {med([r[3]-r[2] for r in rows]):.2f} us  loop {med([r[4]-r[3] for r in rows]):.2f}  epilogue issue {med([r[5]-r[4] for r in rows]):.2f}  "
      f"drain {med([r[6]-r[5] for r in rows]):.2f}  whole {med([r[6]-r[2] for r in rows]):.2f}")
cu = collections.defaultdict(list)
for r in rows: cu[(r[1] >> 32, (r[1] >> 8) & 0xff)].append(r)
gaps, per = [], []
for k, v in cu.items():
    v.sort(key=lambda r: r[2])
    per.append(len(v))
    for a, b in zip(v, v[1:]): gaps.append(b[2] - a[6])
print(f"CUs {len(cu)}, workgroups per CU {min(per)}..{max(per)}; gap between workgroups on a CU: median {med(gaps):.2f} us, "
      f"min {min(gaps)/100:.2f}, max {max(gaps)/100:.2f}")
# first 3 CUs' timelines
for k in list(cu)[:3]:
    print(k, " | ".join(f"{(r[2]-t0)/100:.1f}+{(r[3]-r[2])/100:.1f}/{(r[4]-r[3])/100:.1f}/{(r[5]-r[4])/100:.1f}/{(r[6]-r[5])/100:.1f}" for r in cu[k]))
# by time slice: how many workgroups are in the loop / in the epilogue
for name, lo, hi in (("loop", 3, 4), ("epilogue", 4, 6)):
    ev = sorted([(r[lo], 1) for r in rows] + [(r[hi], -1) for r in rows])
    cur, area, last = 0, 0, t0
    for t, d in ev:
        area += cur * (t - last); last = t; cur += d
    print(f"mean workgroups in {name}: {area / (end - t0):.1f}")
