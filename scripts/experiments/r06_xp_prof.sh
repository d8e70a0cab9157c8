export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pf$v
  ( cd /tmp && BSC_ENC_XP=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$v -- python $GRAFT_REPO_ROOT/scripts/encoder_f32_only.py vit_b16 768 1 > /dev/null 2>&1 )
  echo "== BSC_ENC_XP=$v"; python scripts/stats_top.py $(find /tmp/pf$v -name "*kernel_stats.csv" | head -1) 6 9 | cut -c1-70,100-
done
