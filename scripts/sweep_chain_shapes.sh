for v in "" "BSC_LONG_NWV=8" "BSC_LONG_WAVES=8192" "BSC_LONG_WAVES=4096" "BSC_LONG_NWV=8 BSC_LONG_WAVES=8192" "BSC_LONG_NWV=8 BSC_LONG_WAVES=4096" "BSC_HOT_LOG2=14" "BSC_HOT_LOG2=16" "BSC_REC12=1" "BSC_CHAIN_EAGER=1"; do
  echo "== $v"; env $v python scripts/stage_times.py room 768 5 2>&1 | grep "rep 1"
done
