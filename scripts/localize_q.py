"""localize latency / scan time for a few query counts over 2^20 x D rows.  usage: localize_q.py [D] [Q ...]"""
import sys, time, statistics, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
QS = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8, 12, 16, 32]
V, gL = 1 << 20, 512
eng = B.VoxelEngine(480, 640, gL, 0.1, -25.6, 25.6, 16, D, mode="mean", voxel_capacity=V + 8, max_points=1024)
gen = torch.Generator(device="cuda").manual_seed(5)
codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
rows = torch.randn((V, D), device="cuda", generator=gen)
eng.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
for Q in QS:
    q = torch.randn(Q, D, device="cuda", generator=gen)
    eng.localize(q, K=100)
    eng.kernel_stats(1, reset=True)
    lat = []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter(); eng.localize(q, K=100); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
    ls = eng.kernel_stats(1)
    ms = ls["ms"] / max(1, ls["launches"])
    print(f"D={D} Q={Q:3d} latency {statistics.median(lat) * 1e3:.3f} ms scan {ms:.3f} ms  ({V * D * 4 / ms / 1e6:.0f} GB/s of row bytes)")
