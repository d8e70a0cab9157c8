# bench.py (headline only) under alternating settings on one box: r06_bench_ab.sh "<A>" "<B>" [reps]
A="$1"; B="$2"; reps=${3:-2}
for i in $(seq $reps); do
  for v in "$A" "$B"; do echo -n "[$v] "; env $v timeout 300 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-workloads --no-localize --no-exact --no-side-precision --no-host-feed --no-pmc 2>/dev/null | python scripts/bench_brief.py; done
done
