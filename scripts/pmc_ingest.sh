#!/bin/bash
# The two PMC passes (read requests / write requests of the L2's memory side, by size) of the bench command, summarised per
# ingest kernel.  usage (GPU box, repo root): scripts/pmc_ingest.sh <tag> <commit>   -> gpurun_out/<tag>_pmc_*.{json,txt}
tag=${1:-r05}
commit=${2:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in RD WR; do
  rm -rf /tmp/pmc/$c
  if [ $c = RD ]; then ctr="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"; else ctr="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; fi
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc/$c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-localize --no-workloads --no-side-precision --no-host-feed --no-exact --repeats 1 > /dev/null 2>&1 )
  f=$(find /tmp/pmc/$c -name "*counter_collection.csv" | head -1)
  mkdir -p /tmp/pmc_flat/$c && cp "$f" /tmp/pmc_flat/$c/pmc_counter_collection.csv
done
python scripts/pmc_summary.py /tmp/pmc_flat gpurun_out/${tag}_pmc_ingest_kernels.json $commit > gpurun_out/${tag}_pmc_summary.txt
head -16 gpurun_out/${tag}_pmc_summary.txt
