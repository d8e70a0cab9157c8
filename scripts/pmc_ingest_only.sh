#!/bin/bash
# Quick PMC traffic of bsc_ingest ALONE (scripts/ingest_only.py, no encoder): the same two passes and summary as pmc_ingest.sh, a
# tenth of its run time — for iterating on the ingest kernels; the figure bench.py quotes comes from pmc_ingest.sh.
# usage (GPU box, repo root): scripts/pmc_ingest_only.sh <tag> [frames per call] [kind]   -> gpurun_out/<tag>_pmc_only_*.{json,txt}
tag=${1:-dev}; F=${2:-768}; kind=${3:-room}
export TMPDIR=/tmp
mkdir -p gpurun_out "$(dirname gpurun_out/$tag)"
for c in RD WR; do
  rm -rf /tmp/pmco/$c
  if [ $c = RD ]; then ctr="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"; else ctr="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; fi
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmco/$c -- python $GRAFT_REPO_ROOT/scripts/ingest_only.py 3 sync $F $kind > /dev/null 2>&1 )
  f=$(find /tmp/pmco/$c -name "*counter_collection.csv" | head -1)
  mkdir -p /tmp/pmco_flat/$c && cp "$f" /tmp/pmco_flat/$c/pmc_counter_collection.csv
done
python scripts/pmc_summary.py /tmp/pmco_flat gpurun_out/${tag}_pmc_only_kernels.json dev $F > gpurun_out/${tag}_pmc_only_summary.txt
head -14 gpurun_out/${tag}_pmc_only_summary.txt
