# A/B of environment knobs under the bench's isolated ingest leg: env_ab.sh "VAR=val" "VAR2=val" ...  (one box, alternating)
for v in "" "$@" "" "$@"; do
  env $v python bench.py --no-cpu-baseline --no-localize --no-workloads --no-exact --no-host-feed --no-side-precision --repeats 1 --steps 4 --warmup 2 2>/dev/null > /tmp/line.json
  python - "[$v]" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read()); r = d["roofline"]
ks = {k: round(v["ms_per_call"], 3) for k, v in r["kernels"].items() if isinstance(v, dict) and "ms_per_call" in v}
print(sys.argv[1], "value", round(d["value"]), "wall", round(r["ms_per_call"], 3), "main", round(r["ms_per_call_main_stream_isolated"], 3), ks)
PY
done
