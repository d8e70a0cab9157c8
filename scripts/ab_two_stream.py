"""A/B: encoder and ingest back to back on one stream (what bench.py times) against the encoder running one batch ahead on its own
stream (double-buffered token tensors).  usage: ab_two_stream.py [steps] [kind]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
from bsc_nav_amd import synthetic, encoder
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "room"
F, H, W, gs = 384, 480, 640, 256
torch.cuda.set_stream(torch.cuda.Stream())
vit = encoder.RandomViT("vit_b16", image_size=224, seed=0).cuda()
eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, vit.grid, vit.out_dim, mode="mean", voxel_capacity=3_000_000, max_points=F * H * W)
poses = synthetic.make_poses(kind, 1000, (steps + 2) * F)
chain = B.PoseChain()
Ts = np.stack([chain.pc_transform(p) for p in poses])
frames = [synthetic.make_frames(17 + s, F, H, W, kind, device="cuda", poses=poses[s * F:(s + 1) * F])[:2] for s in range(steps + 2)]
encs = [encoder.GraphedEncoder(vit, F, H, W, 4, True) for _ in range(2)]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def one_stream(lo, hi):
    for s in range(lo, hi):
        tok = encs[0](frames[s][0])
        eng.ingest(frames[s][1], frames[s][0], tok, Ts[s * F:(s + 1) * F])


def two_streams(lo, hi):
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    for e in free:
        e.record(main)
    toks = {}

    def enc(s):
        b = s & 1
        with torch.cuda.stream(side):
            side.wait_event(free[b])
            toks[s] = encs[b](frames[s][0])
            ready[b].record(side)
    enc(lo)
    for s in range(lo, hi):
        if s + 1 < hi:
            enc(s + 1)
        main.wait_event(ready[s & 1])
        eng.ingest(frames[s][1], frames[s][0], toks.pop(s), Ts[s * F:(s + 1) * F])
        free[s & 1].record(main)


for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("one stream", one_stream), ("two streams", two_streams)):
    eng.reset()
    fn(0, 2)
    eng.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    fn(2, steps + 2)
    eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name:12s} {kind}: {1e3 * dt / steps:.2f} ms per step, {steps * F / dt:.0f} frames/s")
