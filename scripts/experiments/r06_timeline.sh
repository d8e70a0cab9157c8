# kernel timeline of pipelined steps of bench.py (headline only): the 5th and 6th k_points launches of the run, chain launches before them included
export TMPDIR=/tmp
rm -rf /tmp/tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed --no-pmc > /tmp/tl.json 2>/dev/null )
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
for k in 5 6; do echo "== k_points occurrence +$k"; python scripts/kernel_timeline.py $f 60 k_points +$k 12 | grep -v "rocprim\|rocclr\|at::native" | cut -c1-118; done > gpurun_out/r06_timeline_pipeline.txt
python scripts/bench_brief.py < /tmp/tl.json
