"""One-time costs: engine creation and the first bsc_ingest of a process (kernel loading) against later engines / calls."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
F, H, W, g, D, gs = 64, 480, 640, 14, 768, 256
poses = synthetic.make_poses("hall", 1000, 2 * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
tok = torch.randn((F, g, g, D), device="cuda").to(torch.bfloat16)
fr = [synthetic.make_frames(17 + s, F, H, W, "hall", poses=poses[s * F:(s + 1) * F]) for s in range(2)]
for e in range(3):
    t0 = time.perf_counter()
    eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=4_000_000, max_points=F * H * W)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ts = []
    for s in range(2):
        eng.ingest(fr[s][1], fr[s][0], tok, Ts[s * F:(s + 1) * F]); eng.sync(); torch.cuda.synchronize()
        ts.append(time.perf_counter())
    print(f"engine {e}: create {1e3*(t1-t0):.1f} ms, call0 {1e3*(ts[0]-t1):.2f} ms, call1 {1e3*(ts[1]-ts[0]):.2f} ms")
    eng.close()
