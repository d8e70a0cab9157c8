for v in "" "BSC_SORT_ROCPRIM=1"; do
  echo "== $v"
  env $v python scripts/stage_times.py room 768 5 2>&1 | grep "rep 1"
  env $v python scripts/ingest_only.py 5 sync 768 room 2>&1 | grep "rep 1"
  env $v python scripts/stage_times.py iid 384 4 2>&1 | grep "rep 1"
  env $v python scripts/ingest_only.py 4 sync 384 iid 2>&1 | grep "rep 1"
  env $v python scripts/stage_times.py hall 384 4 2>&1 | grep "rep 1"
done
