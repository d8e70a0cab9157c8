"""rocPRIM launches (> 15 us) of the last call in a kernel_trace.csv, with their inner kernel names and launch shapes."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    if (e - s) > 15000 and "rocprim" in n:
        inner = re.findall(r"detail::(\w+)", n)
        grid = r.get("Grid_Size_X", r.get("Grid_Size"))
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size"))
        print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} q={r.get('Queue_Id')} grid={grid} wg={wg} {inner[:5]}")
