# fc1 GEMM solution in the pipeline: the stream-K kernel of tunableop_gfx950.csv (what a fresh tune picks: 0.356 ms) against
# the tile kernel 618465 (0.427 ms alone) that was pinned while the rgb chain held CUs for 6 ms per step
ulimit -c 0
sed 's/tn_3072_75648_768_ld_768_768_3072,Gemm_Hipblaslt_618464,0.355978/tn_3072_75648_768_ld_768_768_3072,Gemm_Hipblaslt_618465,0.427/' bsc-nav_amd/tunableop_gfx950.csv > /tmp/tunableop_fc1_tile.csv
run() { timeout 300 python bench.py --kind $1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-localize --no-workloads 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1 $2', round(d['value']), round(d['ms_per_step'], 2), 'enc alone', round(d['stages']['encoder_ms_per_step'], 2), 'enc in pipeline', round(d['roofline']['stage_ms_in_pipeline']['encoder'], 2))"; }
for k in room hall; do for i in 1 2; do BSC_TUNABLEOP_FILE=/tmp/tunableop_fc1_tile.csv run $k tile; run $k streamk; done; done
