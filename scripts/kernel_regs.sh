#!/bin/bash
# registers / scratch of the kernels of one translation unit whose name contains <filter>.  usage: kernel_regs.sh <file.hip> <filter>
d=$(mktemp -d); cd $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/root/repo/include -I/root/repo/bsc-nav_amd/csrc -c "$1" -save-temps=obj -o x.o 2>&1 | grep -E " error" | head -5
python3 - "$2" <<'PY'
import re, sys
name = None
for line in open([f for f in __import__("glob").glob("*gfx950.s")][0]):
    m = re.match(r"\s+\.name:\s+(\S+)", line)
    if m: name = m.group(1)
    m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
    if m: scratch = m.group(1)
    m = re.match(r"\s+\.group_segment_fixed_size:\s+(\d+)", line)
    if m: lds = m.group(1)
    m = re.match(r"\s+\.sgpr_count:\s+(\d+)", line)
    if m: sg = m.group(1)
    m = re.match(r"\s+\.vgpr_count:\s+(\d+)", line)
    if m and name and sys.argv[1] in name: print(f"{name[:90]:90s} vgpr {m.group(1):>4s} sgpr {sg:>4s} scratch {scratch} lds {lds}")
PY
rm -rf $d
