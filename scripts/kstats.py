"""Short view of a rocprofv3 kernel_stats.csv: calls, average us, short kernel name.  usage: kstats.py <csv> [rows] [filter]"""
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for r in rows[:n] if not flt else [r for r in rows if re.search(flt, r[0])][:n]:
    name = r[0]
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)<", name[name.find(">, (") if ">, (" in name else 0:])
        cfg = re.search(r"wrapped_(\w+?)_config", name)
        name = "rocprim " + (cfg.group(1) if cfg else "?") + " / " + (m.group(1) if m else "?")
    else:
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if name.startswith("at::native"):
            name = "torch " + name[:70]
    print(f"{int(r[1]):6d} x {float(r[3]) / 1e3:10.1f} us = {float(r[2]) / 1e6:9.2f} ms  {name[:100]}")
