"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py per libbscnav kernel -> JSON.
usage: python scripts/pmc_summary.py <dir_with_pass_subdirs> <out.json>"""
import collections, csv, json, os, sys
root, out_path = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(root, c, "pmc_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith(("k_", "void k_")):
            vv = v[2:10] if len(v) >= 10 else v          # the 8 timed launches after 2 warm-up steps
            out.setdefault(k, {})[c + "_KiB_per_launch"] = sum(vv) / len(vv)
json.dump({"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python bench.py --no-cpu-baseline --no-localize --no-iid "
                      "(two separate passes; default 8 steps x 384 frames)",
           "units": "KiB per launch, averaged over the 8 timed launches; FETCH_SIZE is raw (gfx950 reports half of wide "
                    "coalesced reads, MI355X_MICROARCH.md, HBM)", "kernels": out}, open(out_path, "w"), indent=1)
print({k: v for k, v in out.items() if "k_dense_reduce" in k})
