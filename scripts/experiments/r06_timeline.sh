# kernel timeline of whole pipelined steps of bench.py (headline only), anchored at the start of an encoder pass: r06_timeline.sh
export TMPDIR=/tmp
rm -rf /tmp/tl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed --no-pmc > /tmp/tl.json 2>/dev/null )
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
for b in 12 13 14; do echo "== k_preprocess occurrence -$b"; python scripts/kernel_timeline.py $f 1400 k_preprocess_patches $b | grep -v "k_gemm_split\|k_attention\|rocprim\|rocclr\|at::native" | cut -c1-120 | head -60; done > gpurun_out/r06_timeline_pipeline.txt
python scripts/bench_brief.py < /tmp/tl.json
