#!/bin/bash
# Round-5 artifacts that depend on the ingest kernels, re-collected after the run-order chain and the listed-heads pair tiles
# (GPU box, repo root): scripts/collect_profiles_r05b.sh <commit>
#   PMC traffic of the bench command (two passes; f32 pipeline, f32 tokens) -> r05_pmc_ingest_kernels.json / r05_pmc_summary.txt
#   the bench line at the driver's flags                                   -> r05_bench_final.json
#   rocprofv3 --kernel-trace --stats of the same command                   -> r05_bench_final_kernel_stats.csv
#   bsc_ingest alone (sync per call, f32 tokens) under rocprofv3            -> r05_ingest_isolated_kernel_stats.csv
commit=${1:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/pmc_ingest.sh r05 $commit > /dev/null
cp gpurun_out/r05_pmc_ingest_kernels.json profiles/r05_pmc_ingest_kernels.json
cp gpurun_out/r05_pmc_summary.txt profiles/r05_pmc_summary.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
rm -rf /tmp/prof_stats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed > $GRAFT_REPO_ROOT/gpurun_out/r05_bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r05_bench_final_kernel_stats.csv
LINES_MAX=1 bash scripts/prof_iso.sh gpurun_out/r05_ingest_isolated_kernel_stats.csv 6 sync 768 room > /dev/null
ls -la gpurun_out/r05_*
head -12 gpurun_out/r05_pmc_summary.txt
cat gpurun_out/r05_bench_final.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'ms_per_call', r['ms_per_call'], 'traffic', r['traffic'])"
