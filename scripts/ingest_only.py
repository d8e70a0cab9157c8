"""Ingest-only loop of the bench workload (no encoder), one synchronize per call: isolated kernel durations for rocprofv3."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
H, W, g, D, gs = 480, 640, 14, 768, 256
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sync = (sys.argv[2] != "nosync") if len(sys.argv) > 2 else True
F = int(sys.argv[3]) if len(sys.argv) > 3 else 384
poses = synthetic.random_walk_poses(1000, calls * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=400000, max_points=F * H * W)
tok = torch.randn((F, g, g, D), device="cuda")
frames = [synthetic.make_frames(17 + s, F, H, W, "room", poses=poses[s * F:(s + 1) * F]) for s in range(calls)]
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for s in range(calls):
        rgb, depth, _ = frames[s]
        eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F])
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"rep {rep}: {(time.perf_counter() - t0) / calls * 1e3:.2f} ms per call (sync={sync})")
