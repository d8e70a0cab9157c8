"""Points per voxel of one 768-frame every-pixel call of the bench workload (the rgb chain's segment lengths)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
F = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H, W, g, D, gs = 480, 640, 14, 16, 256
poses = synthetic.make_poses("room", 1000, 2 * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=400000, max_points=F * H * W)
tok = torch.zeros((F, g, g, D), device="cuda")
prev = None
for s in range(2):
    rgb, depth, _ = synthetic.make_frames(17 + s, F, H, W, "room", poses=poses[s * F:(s + 1) * F])
    eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F])
    acc, cnt = eng.export_dense()
    c = cnt.astype(np.int64)
    d = c.copy()
    if prev is not None:
        d[:len(prev)] -= prev
    prev = c
    d = np.sort(d[d > 0])[::-1]
    hot = d[d >= 32768]
    print(f"call {s}: voxels {len(d)}, points {d.sum()}, hot (>= 32768) {len(hot)} with {hot.sum() / d.sum():.3f} of the points, "
          f"top10 {d[:10].tolist()}, hot median {int(np.median(hot)) if len(hot) else 0}, "
          f"long (512..32767) {((d >= 512) & (d < 32768)).sum()} with {d[(d >= 512) & (d < 32768)].sum() / d.sum():.3f}, "
          f"short (< 512) {(d < 512).sum()} with {d[d < 512].sum() / d.sum():.4f}")
    qs = [0, 1, 2, 5, 10, 25, 50, 75, 100]
    print("   hot length percentiles (desc):", [int(np.percentile(hot, 100 - q)) for q in qs] if len(hot) else [])
