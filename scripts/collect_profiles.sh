#!/bin/bash
# Round artifacts for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC traffic (two separate passes).
# usage (on the GPU box, from the repo root): scripts/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | python scripts/bench_brief.py bench
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python bench.py > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc/$c -- python bench.py --no-cpu-baseline --no-localize --no-iid > /dev/null 2>&1
  f=$(find /tmp/pmc/$c -name "*counter_collection.csv" | head -1)
  mkdir -p /tmp/pmc_flat/$c && cp "$f" /tmp/pmc_flat/$c/pmc_counter_collection.csv
done
python scripts/pmc_summary.py /tmp/pmc_flat gpurun_out/${tag}_pmc_ingest_kernels.json
