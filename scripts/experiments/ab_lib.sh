# A/B of two builds of the library inside one gpurun call: scripts/experiments/libs/libbscnav_base.so against the in-tree one
cp bsc-nav_amd/libbscnav.so /tmp/new.so
timeout 1200 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for v in base new base new; do
  if [ $v = base ]; then cp scripts/experiments/libs/libbscnav_base.so bsc-nav_amd/libbscnav.so; else cp /tmp/new.so bsc-nav_amd/libbscnav.so; fi
  python bench.py --no-cpu-baseline --no-localize --no-workloads --no-exact --no-host-feed --no-side-precision --repeats 1 --steps 4 --warmup 2 2>/dev/null > /tmp/line.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read()); r = d["roofline"]
ks = {k: round(v["ms_per_call"], 3) for k, v in r["kernels"].items() if isinstance(v, dict) and "ms_per_call" in v}
print(sys.argv[1], "value", round(d["value"]), "wall", round(r["ms_per_call"], 3), "main", round(r["ms_per_call_main_stream_isolated"], 3), ks)
PY
done
cp /tmp/new.so bsc-nav_amd/libbscnav.so
if [ -f scripts/experiments/libs/libbscnav_prof.so ]; then
  cp scripts/experiments/libs/libbscnav_prof.so bsc-nav_amd/libbscnav.so
  python bench.py --no-cpu-baseline --no-localize --no-workloads --no-exact --no-host-feed --no-side-precision --repeats 1 --steps 1 --warmup 1 2>&1 | grep "k_patch_pairs wg" | tail -4
  cp /tmp/new.so bsc-nav_amd/libbscnav.so
fi
