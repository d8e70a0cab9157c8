# kernel timeline of the last bsc_ingest call of the ingest-only loop (768 frames, sync per call)
export TMPDIR=/tmp; rm -rf /tmp/tl; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/scripts/ingest_only.py 4 sync 768 room > /tmp/tl.log 2>&1
tail -2 /tmp/tl.log
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/kernel_timeline.py "$f" 70 k_points
