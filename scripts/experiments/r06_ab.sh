# alternating runs of ingest_only.py under two settings (env assignments and / or BSC_LIB_PATH), same box: r06_ab.sh "<A>" "<B>" [reps] [args]
A="$1"; B="$2"; reps=${3:-3}; shift 3
for i in $(seq $reps); do
  for v in "$A" "$B"; do echo -n "[$v] "; env $v python scripts/ingest_only.py ${@:-6 sync 768 room} 2>&1 | tail -1; done
done
