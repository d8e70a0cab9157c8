"""Library-GEMM choice sweep INSIDE the encoder loop: TunableOp ranks hipBLASLt / rocBLAS solutions by their time alone (short
bursts); under the sustained load of the encoder the chip is power-limited, so the ranking can differ.  For each of the encoder's
GEMM shapes at B frames, swaps the committed choice for each candidate id and times scripts/encoder_only.py.
usage (GPU box, repo root): gemm_choice_sweep.py [B] [iters]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
iters = sys.argv[2] if len(sys.argv) > 2 else "40"
M = B * 197
src = open(os.path.join(ROOT, "bsc-nav_amd", "tunableop_gfx950.csv")).read().splitlines()
shapes = {"qkv": f"tn_2304_{M}_768_", "fc1": f"tn_3072_{M}_768_", "fc2": f"tn_768_{M}_3072_", "proj": f"tn_768_{M}_768_"}
cands = ["Gemm_Hipblaslt_618464", "Gemm_Hipblaslt_618465", "Gemm_Hipblaslt_618466", "Gemm_Hipblaslt_618467", "Gemm_Hipblaslt_618613",
         "Gemm_Hipblaslt_618481", "Default"]

def run(lines):
    f = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False)
    f.write("\n".join(lines) + "\n"); f.close()
    env = dict(os.environ, BSC_TUNABLEOP_FILE=f.name, BSC_TUNING="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "encoder_only.py"), "vit_b16", str(B), iters], env=env,
                         capture_output=True, text=True).stdout
    os.unlink(f.name)
    for l in out.splitlines():
        if "ms per forward" in l: return float(l.split(":")[1].split("ms")[0])
    return None

print("committed choices:", run(src), run(src))
for name, key in shapes.items():
    idx = [i for i, l in enumerate(src) if key in l]
    if not idx: print(name, "not in the file"); continue
    cur = src[idx[0]].split(",")[2]
    for c in cands:
        if c == cur: continue
        lines = list(src)
        for i in idx:
            p = lines[i].split(","); p[2] = c; lines[i] = ",".join(p)
        print(f"{name:5s} {cur} -> {c}: {run(lines)} ms per forward", flush=True)
