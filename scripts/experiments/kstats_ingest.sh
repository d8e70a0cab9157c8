# per-kernel averages of the ingest kernels of one bench run (isolated ingest legs included)
export TMPDIR=/tmp; rm -rf /tmp/ks; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-localize --no-workloads --no-exact --no-host-feed --no-side-precision --repeats 1 --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith(("void k_", "k_")) and not n.startswith(("void k_gemm", "k_gemm", "void k_att", "k_att", "void k_embed", "void k_final", "void k_split", "void k_pp", "void k_gs")):
        print(n[:56].ljust(56), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3))
PY
