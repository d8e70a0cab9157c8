#!/bin/bash
# Round-5 artifacts for profiles/ (GPU box, repo root): scripts/collect_profiles_r05.sh <commit>
#   PMC traffic of the bench command (two passes; f32 pipeline, f32 tokens) -> r05_pmc_ingest_kernels.json / r05_pmc_summary.txt
#   the bench line at the driver's flags                                   -> r05_bench_final.json
#   rocprofv3 --kernel-trace --stats of the same command                   -> r05_bench_final_kernel_stats.csv
#   bsc_ingest alone (sync per call, f32 tokens) under rocprofv3            -> r05_ingest_isolated_kernel_stats.csv
#   the f32 encoder alone (768 frames)                                      -> r05_encoder_f32_kernel_stats.csv
#   SQ MFMA counters of the f32 encoder's kernels and of the batched localize scan -> r05_pmc_mfma_counters.txt
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
bash scripts/pmc_one.sh k_points "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" -- python $GRAFT_REPO_ROOT/scripts/ingest_only.py 3 sync 768 room > gpurun_out/r05_pmc_k_points.txt 2>&1
rm -rf /tmp/pf32
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf32 -- python $GRAFT_REPO_ROOT/scripts/encoder_f32_only.py vit_b16 768 1 > $GRAFT_REPO_ROOT/gpurun_out/r05_encoder_f32.log 2>&1 )
cp $(find /tmp/pf32 -name "*kernel_stats.csv" | head -1) gpurun_out/r05_encoder_f32_kernel_stats.csv
rm -rf /tmp/pmc_mfma
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python $GRAFT_REPO_ROOT/scripts/encoder_f32_only.py vit_b16 768 1 > /dev/null 2>&1 )
python scripts/pmc_generic.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) "k_" > gpurun_out/r05_pmc_mfma_counters.txt
rm -rf /tmp/pmc_mfma2
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma2 -- python $GRAFT_REPO_ROOT/scripts/localize_q.py 768 256 > /dev/null 2>&1 )
python scripts/pmc_generic.py $(find /tmp/pmc_mfma2 -name "*counter_collection.csv" | head -1) "k_cosine" >> gpurun_out/r05_pmc_mfma_counters.txt
ls -la gpurun_out/r05_*
