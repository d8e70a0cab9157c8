"""A/B of the rgb chain: wavefront-per-voxel speculate-and-verify kernel for long segments against the quad chain alone
(BSC_QUAD_CHAIN_ONLY=1).  Runs the same frames through two engines in two processes and compares rgb / weight / top-down map.
usage: chain_ab.py [frames per call] [calls] [kind]"""
import os, subprocess, sys, numpy as np
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "room"
if os.environ.get("CHAIN_AB_CHILD"):
    import time, torch
    sys.path.insert(0, "/root/repo")
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    H, W, g, D, gs = 480, 640, 14, 64, 256
    poses = synthetic.make_poses(kind, 1000, calls * F)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=1 << 20, max_points=F * H * W)
    tok = torch.randn((F, g, g, D), device="cuda")
    for s in range(calls):
        rgb, depth, _ = synthetic.make_frames(17 + s, F, H, W, kind, poses=poses[s * F:(s + 1) * F])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F]); eng.sync(); torch.cuda.synchronize()
        print(f"  call {s}: {(time.perf_counter() - t0) * 1e3:.2f} ms, chain {eng.kernel_stats(7)['ms']:.3f} ms", flush=True)
    pos, rgbv, wt = eng.export_rgb()
    mh, cv = eng.export_heightmap()
    np.savez(os.environ["CHAIN_AB_CHILD"], pos=pos, rgb=rgbv, w=wt, cv=cv, mh=mh)
    sys.exit(0)
outs = []
for tag, env in (("long", {}), ("quad", {"BSC_QUAD_CHAIN_ONLY": "1"})):
    out = f"/tmp/chain_ab_{tag}.npz"
    print(tag, flush=True)
    r = subprocess.run(["timeout", "240", sys.executable, __file__, str(F), str(calls), kind], env={**os.environ, **env, "CHAIN_AB_CHILD": out})
    if r.returncode:
        print(f"{tag}: exit {r.returncode}"); sys.exit(1)
    outs.append(np.load(out))
a, b = outs
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print(k, a[k].shape, "equal" if same else f"DIFFER at {np.argwhere(a[k] != b[k])[:5].tolist()} ({int((a[k] != b[k]).sum())})")
