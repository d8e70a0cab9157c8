#!/bin/bash
# on the GPU box: rebuild with the phase stamps, run one GEMM shape a few times, summarise.  usage: gemm_phase_profile.sh [shapes...]
cd $GRAFT_REPO_ROOT
touch bsc-nav_amd/csrc/encoder_gemm.hip
BSC_EXTRA_FLAGS="-DBSC_GEMM_PROFILE $BSC_PROFILE_MORE" bash bsc-nav_amd/csrc/build.sh > /dev/null 2>&1 || { echo build failed; exit 1; }
mkdir -p gpurun_out
for shp in ${@:-proj qkv}; do
  BSC_GEMM_PROFILE_DUMP=1 timeout 120 python scripts/gemm_split_prof.py $shp 3 2> gpurun_out/gemm_phase_$shp.log
  echo "== $shp"; python scripts/gemm_phase_profile.py gpurun_out/gemm_phase_$shp.log
done
