#!/bin/bash
# usage: gemm_vs_hog.sh <solution for fc1: Default | Gemm_Hipblaslt_NNN>
sol=${1:-Default}
sed "s/tn_3072_75648_768_ld_768_768_3072,[A-Za-z_0-9-]*,/tn_3072_75648_768_ld_768_768_3072,$sol,/" bsc-nav_amd/tunableop_gfx950.csv > /tmp/tun_${sol}_0.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tun_${sol}_.csv python scripts/gemm_vs_hog.py
