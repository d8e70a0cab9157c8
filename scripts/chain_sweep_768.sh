# chain knobs at 768 frames per call (bench.py, isolated stage times)
for cfg in "none" "BSC_LONG_WAVES=4096" "BSC_LONG_WAVES=8192" "BSC_LONG_WAVES=32768" "BSC_LONG_WAVES=65536" "BSC_CHAIN_WAVES=1024" "BSC_CHAIN_WAVES=8192" "BSC_QUAD_CHAIN_ONLY=1"; do
  if [ "$cfg" = none ]; then e=""; else e="$cfg"; fi
  env $e python bench.py --no-cpu-baseline --no-localize --no-workloads --no-exact --no-host-feed --no-side-precision --repeats 1 --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"$cfg\", \"wall\", round(r[\"ms_per_call\"],3), \"main\", round(r[\"ms_per_call_main_stream_isolated\"],3), \"chain\", round(r[\"kernels\"][\"k_chain\"][\"ms_per_call\"],3), \"order\", round(r[\"kernels\"][\"ids+point_order\"][\"ms_per_call\"],3))"
done
