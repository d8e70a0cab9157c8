#!/bin/bash
# rocprofv3 kernel stats of bsc_ingest running alone.  usage (GPU box, repo root): scripts/prof_iso.sh <out csv> [ingest_only.py args]
out=$1; shift
ulimit -c 0
export TMPDIR=/tmp
rm -rf /tmp/prof_iso
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_iso -- python /root/repo/scripts/ingest_only.py "$@" > /tmp/prof_iso.log 2>&1 )
f=$(find /tmp/prof_iso -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out"; cut -d, -f1-4 "$out" | cut -c1-160 | head -${LINES_MAX:-32}; else echo "no stats file"; tail -5 /tmp/prof_iso.log; fi
