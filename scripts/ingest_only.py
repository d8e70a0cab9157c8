"""Ingest-only loop of the bench workload (no encoder): isolated kernel durations for rocprofv3.
usage: ingest_only.py [calls] [sync|nosync] [frames per call] [kind] [arch dims: g D gs]   (BSC_TOKENS=bf16: bf16 token rows)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sync = (sys.argv[2] != "nosync") if len(sys.argv) > 2 else True
F = int(sys.argv[3]) if len(sys.argv) > 3 else 384
kind = sys.argv[4] if len(sys.argv) > 4 else "room"
g, D, gs = (int(v) for v in sys.argv[5:8]) if len(sys.argv) > 7 else (14, 768, 256)
H, W = 480, 640
poses = synthetic.make_poses(kind, 1000, calls * F) if hasattr(synthetic, "make_poses") else synthetic.random_walk_poses(1000, calls * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
half = gs * 0.05
vcap = 400000 if kind == "room" else 4_000_000
eng = B.VoxelEngine(H, W, gs, 0.1, -half, half, g, D, mode="mean", voxel_capacity=vcap, max_points=F * H * W)
tok = torch.randn((F, g, g, D), device="cuda")
if os.environ.get("BSC_TOKENS") == "bf16":
    tok = tok.bfloat16()
frames = [synthetic.make_frames(17 + s, F, H, W, kind, poses=poses[s * F:(s + 1) * F]) for s in range(calls)]
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for s in range(calls):
        rgb, depth, _ = frames[s]
        eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F])
        if sync:
            eng.sync()
            torch.cuda.synchronize()
    eng.sync()
    torch.cuda.synchronize()
    c = eng.counters()
    print(f"rep {rep}: {(time.perf_counter() - t0) / calls * 1e3:.2f} ms per call (sync={sync}) kind={kind} voxels={c['max_id']} "
          f"pairs_last={c['pairs_last_call']}")
