#!/bin/bash
# Round-6 artifacts for profiles/ (GPU box, repo root): scripts/collect_profiles_r06.sh <commit> [full]
#   PMC traffic of the bench command (two passes)                         -> r06_pmc_ingest_kernels.json / r06_pmc_summary.txt
#   the bench line at the driver's flags                                   -> r06_bench_final.json
#   rocprofv3 --kernel-trace --stats of the same command                   -> r06_bench_final_kernel_stats.csv
#   bsc_ingest alone (sync per call, f32 tokens) under rocprofv3            -> r06_ingest_isolated_kernel_stats.csv
#   "full": also the f32 encoder alone and the SQ MFMA counters             -> r06_encoder_f32_kernel_stats.csv / r06_pmc_mfma_counters.txt
commit=${1:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/pmc_ingest.sh r06 $commit > /dev/null
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
rm -rf /tmp/prof_stats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/r06_bench_under_rocprof.json 2>/dev/null )
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r06_bench_final_kernel_stats.csv
LINES_MAX=1 bash scripts/prof_iso.sh gpurun_out/r06_ingest_isolated_kernel_stats.csv 6 sync 768 room > /dev/null
if [ "$2" = full ]; then
  rm -rf /tmp/pf32
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf32 -- python $GRAFT_REPO_ROOT/scripts/encoder_f32_only.py vit_b16 768 1 > $GRAFT_REPO_ROOT/gpurun_out/r06_encoder_f32.log 2>&1 )
  cp $(find /tmp/pf32 -name "*kernel_stats.csv" | head -1) gpurun_out/r06_encoder_f32_kernel_stats.csv
  rm -rf /tmp/pmc_mfma
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python $GRAFT_REPO_ROOT/scripts/encoder_f32_only.py vit_b16 768 1 > /dev/null 2>&1 )
  python scripts/pmc_generic.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) "k_" > gpurun_out/r06_pmc_mfma_counters.txt
fi
ls -la gpurun_out/r06_*
head -14 gpurun_out/r06_pmc_summary.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1]); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'ms_per_call', r.get('ms_per_call'), 'traffic', r['traffic'])
PY
