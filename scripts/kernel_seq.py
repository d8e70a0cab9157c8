"""Durations of the launches of kernels whose name contains <filter>, in launch order, from a rocprofv3 kernel_trace.csv.
usage: kernel_seq.py <kernel_trace.csv> <filter> [max rows]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[-n:]:
    print(f"{r['Kernel_Name'][:40]:40s} grid={r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
