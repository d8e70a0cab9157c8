ulimit -c 0
run() { timeout 300 python bench.py --kind $1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-localize --no-workloads $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); st = d['roofline']['stage_ms_in_pipeline']; print('$1 $2', round(d['value']), round(d['ms_per_step'], 2), 'ingest', round(st['bsc_ingest'], 2), 'chain', round(st['k_chain'], 2), 'enc', round(st['encoder'], 2), 'alone: enc', round(d['stages']['encoder_ms_per_step'], 2), 'ingest', round(d['stages']['ingest_ms_per_step'], 2))"; }
for k in hall room; do run $k overlap ""; run $k no-overlap "--no-overlap"; run $k serial "--serial"; done
