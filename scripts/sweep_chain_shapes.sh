for v in "" "BSC_CHAIN_BESIDE_PAIRSORT=1" "BSC_LONG_NWV=16" "BSC_LONG_WAVES=8192" "BSC_LONG_WAVES=32768"; do
  echo "== $v"; env $v python scripts/stage_times.py room 768 5 2>&1 | grep "rep 1"; env $v python scripts/ingest_only.py 5 sync 768 room 2>&1 | grep "rep 1"
done
for k in hall iid; do python scripts/ingest_only.py 4 sync 384 $k 2>&1 | grep "rep 1"; BSC_LONG_NWV=16 BSC_CHAIN_BESIDE_PAIRSORT=1 python scripts/ingest_only.py 4 sync 384 $k 2>&1 | grep "rep 1"; done
