export TMPDIR=/tmp
rm -rf /tmp/tl2
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -- python $GRAFT_REPO_ROOT/scripts/ingest_only.py 3 sync 768 room > /dev/null 2>&1 )
f=$(find /tmp/tl2 -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 60 k_points | grep -v "at::native" | cut -c1-110
