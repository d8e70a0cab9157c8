# per-kernel averages of the ingest kernels, base library against the in-tree one, then the bench A/B
cp bsc-nav_amd/libbscnav.so /tmp/new.so
for v in base new; do
  if [ $v = base ]; then cp scripts/experiments/libs/libbscnav_base.so bsc-nav_amd/libbscnav.so; else cp /tmp/new.so bsc-nav_amd/libbscnav.so; fi
  echo "== $v"; bash scripts/experiments/kstats_ingest.sh 2>&1 | grep -E "k_chain|k_expand|k_run_keys|k_hwin|k_points|k_patch_pairs|k_dense_reduce" 
done
cp /tmp/new.so bsc-nav_amd/libbscnav.so
sed -i 's/^timeout 1500 python -m pytest.*$/true/' scripts/experiments/ab_lib.sh
bash scripts/experiments/ab_lib.sh
