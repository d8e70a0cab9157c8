ulimit -c 0
run() { timeout 300 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-localize --no-workloads $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); st = d['roofline']['stage_ms_in_pipeline']; print('$1', round(d['value']), round(d['ms_per_step'], 2), 'reduce', round(st['k_dense_reduce'], 2), 'ingest', round(st['bsc_ingest'], 2), 'enc', round(st['encoder'], 2), 'iso', round(d['roofline']['ms_per_call_isolated'], 2))"; }
for i in 1 2; do run bf16 "--tokens bf16"; run f32 "--tokens f32"; done
