"""Batched localize loop for rocprofv3: Q queries over a 2^20 x D dense map.  usage: localize_only.py [Q] [D] [reps] [K]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
K = int(sys.argv[4]) if len(sys.argv) > 4 else 100
V, gL = 1 << 20, 512
eng = B.VoxelEngine(480, 640, gL, 0.1, -gL * 0.05, gL * 0.05, 16, D, mode="mean", voxel_capacity=V + 8, max_points=1024)
gen = torch.Generator(device="cuda").manual_seed(5)
codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
rows = torch.randn((V, D), device="cuda", generator=gen)
eng.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
q = torch.randn(Q, D, device="cuda", generator=gen)
eng.localize(q, K=K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    eng.localize(q, K=K)
torch.cuda.synchronize()
print(f"Q={Q} D={D} K={K}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call")
