"""Per-stage HIP-event times of bsc_ingest running alone with a synchronize per call (library stage timers).
usage: stage_times.py [kind] [frames per call] [calls]"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 384
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
H, W, g, D, gs = 480, 640, 14, 768, 256
poses = synthetic.make_poses(kind, 1000, calls * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=4_000_000, max_points=F * H * W)
tok = torch.randn((F, g, g, D), device="cuda")
if os.environ.get("BSC_TOKENS") == "bf16":
    tok = tok.bfloat16()
frames = [synthetic.make_frames(17 + s, F, H, W, kind, poses=poses[s * F:(s + 1) * F]) for s in range(calls)]
names = {2: "points", 3: "pairs", 4: "order", 5: "pairsort", 0: "reduce", 6: "ingest", 7: "chain"}
for rep in range(2):
    for w in names:
        eng.kernel_stats(w, reset=True)
    for s in range(calls):
        eng.ingest(frames[s][1], frames[s][0], tok, Ts[s * F:(s + 1) * F])
        eng.sync()
        torch.cuda.synchronize()
    st = {n: eng.kernel_stats(w) for w, n in names.items()}
    c = eng.counters()
    print(f"rep {rep} {kind}: " + " ".join(f"{n}={v['ms'] / max(1, v['launches']):.3f}" for n, v in st.items()) + f" | voxels={c['max_id']} pairs={c['pairs_last_call']}")
