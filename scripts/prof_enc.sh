#!/bin/bash
# rocprofv3 kernel stats of the encoder running alone.  usage (GPU box, repo root): scripts/prof_enc.sh <out csv> [encoder_only.py args]
out=$1; shift
ulimit -c 0
export TMPDIR=/tmp
rm -rf /tmp/prof_enc
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -- python /root/repo/scripts/encoder_only.py "$@" > /tmp/prof_enc.log 2>&1 )
f=$(find /tmp/prof_enc -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out"; tail -2 /tmp/prof_enc.log; else echo "no stats file"; tail -5 /tmp/prof_enc.log; fi
