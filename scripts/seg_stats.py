"""Distribution of points per voxel in one 128-frame call of the bench workload (chain segment lengths)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
H, W, g, D, gs, F = 480, 640, 14, 64, 256, 128
poses = synthetic.random_walk_poses(1000, 4 * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=400000, max_points=F * H * W)
tok = torch.zeros((F, g, g, D), device="cuda")
prev = None
for s in range(3):
    rgb, depth, _ = synthetic.make_frames(17 + s, F, H, W, "room", poses=poses[s * F:(s + 1) * F])
    eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F])
    acc, cnt = eng.export_dense()
    c = cnt.astype(np.int64)
    d = c.copy()
    if prev is not None:
        d[:len(prev)] -= prev
    prev = c
    d = d[d > 0]
    print(f"call {s}: voxels touched {len(d)}, points {d.sum()}, mean {d.mean():.0f}, median {np.median(d):.0f}, "
          f"p99 {np.percentile(d, 99):.0f}, max {d.max()}, top5 {np.sort(d)[-5:][::-1].tolist()}, "
          f"sum(top 64) share {np.sort(d)[-64:].sum() / d.sum():.3f}")
