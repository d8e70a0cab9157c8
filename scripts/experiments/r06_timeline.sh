# kernel timeline of one pipelined step of bench.py (headline only; the step three k_points launches before the last): r06_timeline.sh
export TMPDIR=/tmp
rm -rf /tmp/tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed --no-pmc > /tmp/tl.json 2>/dev/null )
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
for b in 2 3 4 5 6 7 8; do echo "== occurrence -$b"; python scripts/kernel_timeline.py $f 70 k_points $b | grep -v "k_gemm_split\|k_attention" | cut -c1-120; done > gpurun_out/r06_timeline_pipeline.txt
python scripts/bench_brief.py < /tmp/tl.json
