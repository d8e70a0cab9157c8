#!/bin/bash
# FETCH_SIZE of the rgb chain kernels of bsc_ingest alone, per launch, for a hot-split threshold.  usage: pmc_chain.sh <kind> [BSC_HOT_LOG2 ...]
kind=$1; shift
ulimit -c 0
export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/pmc_chain
  if [ "$L" = none ]; then export BSC_NO_HOT_SPLIT=1; unset BSC_HOT_LOG2; else unset BSC_NO_HOT_SPLIT; export BSC_HOT_LOG2=$L; fi
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_chain -- python /root/repo/scripts/ingest_only.py 6 sync 384 $kind > /dev/null 2>&1 )
  f=$(find /tmp/pmc_chain -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$L" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_chain" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print("hot_log2=" + sys.argv[2], {k: round(2 * sum(v[-6:]) / len(v[-6:]) / 1024, 1) for k, v in agg.items()}, "MB fetched per launch (2 x FETCH_SIZE KiB), last 6 calls")
PY
done
