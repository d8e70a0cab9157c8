"""rocprofv3 target: a few localize calls over 2^20 x D rows.  usage: profile_localize.py [Q] [D]"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bsc_nav_amd as B
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
V, gL = 1 << 20, 512
eng = B.VoxelEngine(480, 640, gL, 0.1, -25.6, 25.6, 14, D, mode="mean", voxel_capacity=V + 8, max_points=1024)
gen = torch.Generator(device="cuda").manual_seed(5)
codes = torch.randperm(gL ** 3, device="cuda", generator=gen)[:V]
keys = torch.stack([codes // (gL * gL), (codes // gL) % gL, codes % gL], dim=1).to(torch.int32).contiguous()
rows = torch.randn((V, D), device="cuda", generator=gen)
eng.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
q = torch.randn(Q, D, device="cuda", generator=gen)
for _ in range(5):
    eng.localize(q, K=100)
torch.cuda.synchronize()
