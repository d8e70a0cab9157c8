# pipeline step time against the number of wavefronts of the speculating rgb chain (BSC_LONG_WAVES) — run on the GPU box
ulimit -c 0
run() { timeout 200 python bench.py --kind $1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-localize --no-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); st = d['roofline']['stage_ms_in_pipeline']
print('$1 $2', round(d['ms_per_step'], 2), 'ingest', round(st['bsc_ingest'], 2), 'chain', round(st['k_chain'], 2), 'enc', round(st['encoder'], 2))"; }
for k in ${KINDS:-hall room}; do
  for wv in ${WAVES:-4096 8192 16384}; do BSC_LONG_WAVES=$wv run $k waves=$wv; done
done
