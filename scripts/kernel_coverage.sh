#!/bin/bash
# Which kernels of libbscnav.so do the GPU tests launch?  usage (GPU box, repo root): scripts/kernel_coverage.sh
ulimit -c 0
export TMPDIR=/tmp
rm -rf /tmp/cov
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cov -- python -m pytest tests -m gpu -x -q > /tmp/cov.log 2>&1 )
tail -1 /tmp/cov.log
python - <<'PY'
import csv, glob, re, subprocess
seen = set()
for f in glob.glob("/tmp/cov/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Name"])
        if m: seen.add(m.group(1))
syms = subprocess.run("strings -a bsc-nav_amd/libbscnav.so | grep -o '_Z[0-9]*k_[a-z0-9_]*' | sed 's/_Z[0-9]*//' | sort -u", shell=True, capture_output=True, text=True).stdout.split()
have = set()
for sname in syms:
    m = re.match(r"(k_[a-z0-9_]+?)(I[A-Z]|P|i|l|v|$)", sname)
    have.add(m.group(1) if m else sname)
have = {h for h in have if h.startswith("k_")}
print(len(seen), "kernel names launched by the tests")
missing = sorted(h for h in have if not any(h == s or s.startswith(h) or h.startswith(s) for s in seen))
print("in the library but never launched:", missing)
PY
