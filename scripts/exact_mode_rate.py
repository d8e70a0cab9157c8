"""Exact (reference-semantics) mode throughput at the reference's default sampling (depth_sample_rate=1000)
and at a dense setting, 640x480, D=1024, 16x16 tokens — the configuration of the survey's probe of the
reference loop (8.5 frames/s)."""
import sys, time, random
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic

H, W, g, D, gs = 480, 640, 16, 1024, 256
F = 256
poses = synthetic.random_walk_poses(3, F)
rgb, depth, _ = synthetic.make_frames(3, F, H, W, "room", poses=poses)
tokens = torch.randn(F, g, g, D, device="cuda")
for sampler, s, batch in ((B.sample_indices, 1000, 32), (B.sample_indices_fast, 1000, 32), (B.sample_indices_fast, 1000, 1),
                          (B.sample_indices_fast, 50, 32), (B.sample_indices_fast, 1, 8)):
    eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="exact", voxel_capacity=2_000_000,
                        token_capacity=6_000_000, max_points=batch * H * W)
    chain = B.PoseChain()
    np.random.seed(0); random.seed(0)
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    torch.cuda.synchronize()
    t = time.perf_counter()
    t_host = 0.0
    for a in range(0, F, batch):
        th = time.perf_counter()
        idxs = [sampler(H * W, s) for _ in range(batch)]
        off = np.concatenate([[0], np.cumsum([len(i) for i in idxs])]).astype(np.int64)
        idx = torch.from_numpy(np.concatenate(idxs)).cuda()
        t_host += time.perf_counter() - th
        eng.ingest(depth[a:a + batch], rgb[a:a + batch], tokens[a:a + batch], Ts[a:a + batch], idx, off)
    eng.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    k = eng.counters()
    print(f"{sampler.__name__} s={s} batch={batch}: {F/dt:.1f} frames/s ({dt*1e3/F:.2f} ms/frame, host shuffle {t_host*1e3/F:.2f} ms/frame) "
          f"voxels={k['max_id']} store_tokens={k['store_tokens']} flushes={k['flushes']}")
    eng.close()
