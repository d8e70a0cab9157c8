ulimit -c 0
run() { timeout 300 python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-localize --no-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'], 2), round(d['stages']['encoder_ms_per_step'], 2))"; }
for i in 1 2; do run nocopy; BSC_GRAPH_COPY=1 run copy; done
