"""Per-kernel averages of any rocprofv3 --pmc pass.  usage: pmc_generic.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][:48]
    if flt in name:
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
