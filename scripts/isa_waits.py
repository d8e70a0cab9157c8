"""Memory instructions and s_waitcnt of one kernel of a translation unit, in program order — to spot a wait for every load in flight
(vmcnt(0)) that the compiler put into a hot loop.  usage: isa_waits.py <file.hip> <mangled-name prefix> [extra regex] [-D flags...]"""
import re, subprocess, sys, tempfile, glob, os
src, prefix = sys.argv[1], sys.argv[2]
extra = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else None
flags = [a for a in sys.argv[3:] if a.startswith("-")]
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I/root/repo/include",
                "-I/root/repo/bsc-nav_amd/csrc", *flags, "-c", src, "-save-temps=obj", "-o", d + "/x.o"], cwd=d, stderr=subprocess.DEVNULL)
lines = open(glob.glob(d + "/*gfx950.s")[0]).read().split("\n")
starts = [i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(";")[0].rstrip().endswith(":")]
for st in starts:
    end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = [re.sub(r"\s+;.*$", "", l) for l in lines[st:end]]
    print("==", lines[st].split(":")[0], len(body), "lines")
    pat = r"s_waitcnt|global_load|global_store|global_atomic|buffer_|s_barrier|s_cbranch|^\.LBB|flat_|scratch_" + ("|" + extra if extra else "")
    for i, l in enumerate(body):
        if re.search(pat, l.strip()):
            print(f"{i:5d} {l.strip()}")
