#!/bin/bash
# GPU box: sliced-reduce parity test, then the serialized duration + L2 hit / miss counters of the dense reduce in the iid workload,
# column-sliced (BSC_SLICED_MIN_PAIRS=1) and per-voxel (the default)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_edges.py -m gpu -x -q -k "sliced" 2>&1 | grep -E "^E|passed|failed|Error" | head -20
export BSC_TOKENS=bf16
for m in 1 99999999999; do
  echo "== BSC_SLICED_MIN_PAIRS=$m"
  BSC_SLICED_MIN_PAIRS=$m bash scripts/pmc_one.sh k_dense_reduce "TCC_HIT TCC_MISS" -- python $GRAFT_REPO_ROOT/scripts/ingest_only.py 2 sync 384 iid
done
