#!/bin/bash
# Round artifacts for profiles/: PMC traffic (two separate passes) first, so that the bench line that follows quotes the traffic
# of this very commit; then the bench line, rocprofv3 kernel stats of the same command, MFMA counters of the encoder.
# usage (on the GPU box, from the repo root): scripts/collect_profiles.sh <tag> <commit>
tag=${1:-r03}
commit=${2:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in RD WR; do
  rm -rf /tmp/pmc/$c
  if [ $c = RD ]; then ctr="TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"; else ctr="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; fi
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc/$c -- python bench.py --no-cpu-baseline --no-localize --no-workloads --no-f32 --repeats 1 > /dev/null 2>&1
  f=$(find /tmp/pmc/$c -name "*counter_collection.csv" | head -1)
  mkdir -p /tmp/pmc_flat/$c && cp "$f" /tmp/pmc_flat/$c/pmc_counter_collection.csv
done
python scripts/pmc_summary.py /tmp/pmc_flat gpurun_out/${tag}_pmc_ingest_kernels.json $commit > gpurun_out/${tag}_pmc_summary.txt
cp gpurun_out/${tag}_pmc_ingest_kernels.json profiles/${tag}_pmc_ingest_kernels.json      # bench.py reads roofline.traffic from here
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
# matrix-core utilisation of the encoder GEMMs / attention and of the batched cosine (separate pass, SQ counters only)
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python bench.py --no-cpu-baseline --no-workloads --repeats 1 > /dev/null 2>&1
python scripts/pmc_generic.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) "" > gpurun_out/${tag}_pmc_mfma.txt
