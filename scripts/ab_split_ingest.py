"""Encoder batch of 768 frames with bsc_ingest called once (768 frames) or twice (2 x 384) per step.  usage: ab_split_ingest.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench, torch
a = bench.parse()
steps, warm = 10, 3
p = bench.Pipeline(a, "room", a.arch, a.grid, 768, steps + warm, 0, 0)
def run(split):
    p.eng.reset()
    torch.cuda.synchronize()
    for s in range(steps + warm):
        if s == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        tok = p.enc(p.rgbs[s])
        T = p.Ts[s * 768:(s + 1) * 768]
        if split:
            for h in (0, 1):
                sl = slice(384 * h, 384 * (h + 1))
                p.eng.ingest(p.depths[s][sl], p.rgbs[s][sl], tok[sl], T[sl])
        else:
            p.eng.ingest(p.depths[s], p.rgbs[s], tok, T)
    p.eng.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for r in range(3):
    print(f"one call of 768: {run(False):.2f} ms per step   two calls of 384: {run(True):.2f} ms per step", flush=True)
