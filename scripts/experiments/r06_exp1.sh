export TMPDIR=/tmp
for a in "6 sync 768" "6 nosync 768" "12 nosync 384" "24 nosync 192" "12 sync 384"; do echo "== $a"; python scripts/ingest_only.py $a room 2>&1 | tail -1; done
LINES_MAX=40 bash scripts/prof_iso.sh gpurun_out/r06_iid_isolated_kernel_stats.csv 4 sync 384 iid
