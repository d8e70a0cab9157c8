"""Stage timers of bsc_ingest call by call (alone, a synchronize per call): the start-up transient of a scene.
usage: per_call_times.py [kind] [frames per call] [calls]"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "hall"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 384
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H, W, g, D, gs = 480, 640, 14, 768, 256
poses = synthetic.make_poses(kind, 1000, calls * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=4_000_000, max_points=F * H * W)
tok = torch.randn((F, g, g, D), device="cuda").to(torch.bfloat16)
names = {2: "points", 3: "pairs", 4: "order", 5: "pairsort", 0: "reduce", 6: "ingest", 7: "chain"}
prev = 0
for s in range(calls):
    rgb, depth, _ = synthetic.make_frames(17 + s, F, H, W, kind, poses=poses[s * F:(s + 1) * F])
    for w in names:
        eng.kernel_stats(w, reset=True)
    eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F])
    eng.sync(); torch.cuda.synchronize()
    st = {n: eng.kernel_stats(w) for w, n in names.items()}
    c = eng.counters()
    print(f"call {s}: " + " ".join(f"{n}={v['ms'] / max(1, v['launches']):.2f}" for n, v in st.items()) + f" | new voxels {c['max_id'] - prev} total {c['max_id']}")
    prev = c["max_id"]
