"""A/B of two library variants on the same frames: every exported array of the two maps must be bit-identical, and the per-stage
times of both are printed.  A variant is a set of environment switches (and / or BSC_LIB_PATH=<an A/B build of csrc/>).
usage: variant_ab.py "<ENV=1 ...>" "<ENV=1 ...>" [frames per call] [calls] [kind] [D]
e.g.   variant_ab.py "" "BSC_REC12=1" 64 3 room 64        (8-byte records against the 12-byte {alpha, rgb} records)"""
import os, subprocess, sys, numpy as np
va, vb = sys.argv[1], sys.argv[2]
F = int(sys.argv[3]) if len(sys.argv) > 3 else 16
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kind = sys.argv[5] if len(sys.argv) > 5 else "room"
D = int(sys.argv[6]) if len(sys.argv) > 6 else 64
if os.environ.get("VARIANT_AB_CHILD"):
    import time, torch
    sys.path.insert(0, "/root/repo")
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    H, W, g, gs = 480, 640, 14, 256
    poses = synthetic.make_poses(kind, 1000, calls * F)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=1 << 22, max_points=F * H * W)
    torch.manual_seed(3)
    tok = torch.randn((F, g, g, D), device="cuda")
    names = {2: "points", 3: "pairs", 4: "order", 5: "pairsort", 0: "reduce", 6: "ingest", 7: "chain"}
    for rep in range(2):
        if rep:
            eng.reset()
        for w in names:
            eng.kernel_stats(w, reset=True)
        t_all = 0.0
        for s in range(calls):
            rgb, depth, _ = synthetic.make_frames(17 + s, F, H, W, kind, poses=poses[s * F:(s + 1) * F])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.ingest(depth, rgb, tok, Ts[s * F:(s + 1) * F]); eng.sync(); torch.cuda.synchronize()
            t_all += time.perf_counter() - t0
        st = {n: eng.kernel_stats(w) for w, n in names.items()}
        print(f"  rep {rep}: {t_all / calls * 1e3:.2f} ms per call + sync | " +
              " ".join(f"{n}={v['ms'] / max(1, v['launches']):.3f}" for n, v in st.items()), flush=True)
    pos, rgbv, wt = eng.export_rgb()
    mh, cv = eng.export_heightmap()
    acc, cnt = eng.export_dense()
    np.savez(os.environ["VARIANT_AB_CHILD"], pos=pos, rgb=rgbv, w=wt, cv=cv, mh=mh, acc=acc, cnt=cnt)
    sys.exit(0)
outs = []
for tag, var in (("A", va), ("B", vb)):
    env = dict(kv.split("=", 1) for kv in var.split())
    out = f"/tmp/variant_ab_{tag}.npz"
    print(f"{tag}: {var or '(default)'}", flush=True)
    r = subprocess.run(["timeout", "400", sys.executable, __file__, va, vb, str(F), str(calls), kind, str(D)],
                       env={**os.environ, **env, "VARIANT_AB_CHILD": out})
    if r.returncode:
        print(f"{tag}: exit {r.returncode}"); sys.exit(1)
    outs.append(np.load(out))
a, b = outs
bad = 0
for k in a.files:
    same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
    bad += not same
    print(k, a[k].shape, "equal" if same else f"DIFFER ({int((a[k] != b[k]).sum()) if a[k].shape == b[k].shape else 'shape'})")
sys.exit(1 if bad else 0)
