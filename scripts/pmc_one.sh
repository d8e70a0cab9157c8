#!/bin/bash
# rocprofv3 --pmc passes (one per quoted counter group) of a command, mean per kernel whose name contains <filter>.
# usage (GPU box, repo root): scripts/pmc_one.sh <filter> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>
filt=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
export TMPDIR=/tmp
n=0
for g in "${groups[@]}"; do
  n=$((n+1)); rm -rf /tmp/pmc1_$n
  ( cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc1_$n -- "$@" > /dev/null 2>&1 )
  f=$(find /tmp/pmc1_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$filt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r:
            agg[(r["Kernel_Name"].split("(")[0][-60:], "duration_us (serialized)")].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(agg.items()):
    print(f"{k:60s} {c:28s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
PY
done
