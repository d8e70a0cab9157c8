"""Seeded synthetic frame source on the device: stands in for the simulator
(env.py:166-235) in benchmarks and in the create_memory_for_dataset-shaped loop.

rgb (F,H,W,4) u8 RGBA, depth (F,H,W) f32 z-depth, poses (F,7) [px,py,pz,qx,qy,qz,qw].
`room`: analytic ray cast of an axis-aligned 8 x 3 x 6 m box from a camera on a seeded random walk
(0.25 m steps, 30 degree turns, args.py:33-35); `iid`: U(0.5, 5) m per pixel (worst case, one voxel
per point).  About 2 % of the pixels are pushed outside (min_depth, max_depth).
"""
import numpy as np
import torch

ROOM_LO = (-4.0, -1.5, -3.0)
ROOM_HI = (4.0, 1.5, 3.0)


def random_walk_poses(seed, n_frames, start_yaw_steps=0):
    rs = np.random.RandomState(seed)
    lo, hi = np.array(ROOM_LO), np.array(ROOM_HI)
    poses = np.zeros((n_frames, 7), dtype=np.float64)
    pos, k = np.zeros(3), int(start_yaw_steps)
    for f in range(n_frames):
        a = rs.randint(0, 3)
        if f > 0:
            if a == 0:
                th = k * np.pi / 6.0
                nxt = pos + np.array([-np.sin(th), 0.0, -np.cos(th)]) * 0.25
                if np.all(nxt > lo + 0.6) and np.all(nxt < hi - 0.6):
                    pos = nxt
            elif a == 1:
                k += 1
            else:
                k -= 1
        th = k * np.pi / 6.0
        poses[f, :3] = pos
        poses[f, 3:] = [0.0, np.sin(th / 2.0), 0.0, np.cos(th / 2.0)]
    return poses


@torch.no_grad()
def make_frames(seed, n_frames, H, W, kind="room", device="cuda", invalid_frac=0.02, poses=None):
    gen = torch.Generator(device=device).manual_seed(seed)
    if poses is None:
        poses = random_walk_poses(seed, n_frames)
    rgb = torch.randint(0, 255, (n_frames, H, W, 4), dtype=torch.uint8, device=device, generator=gen)
    if kind == "iid":
        depth = torch.rand((n_frames, H, W), device=device, generator=gen) * 4.5 + 0.5
    elif kind == "room":
        fx = W / 2.0
        u = (torch.arange(W, device=device, dtype=torch.float32) + 0.5 - W / 2.0) / fx
        v = (torch.arange(H, device=device, dtype=torch.float32) + 0.5 - H / 2.0) / fx
        vv, uu = torch.meshgrid(v, u, indexing="ij")
        d_local = torch.stack([uu, -vv, -torch.ones_like(uu)], dim=-1)             # (H,W,3)
        p = torch.from_numpy(poses).to(device=device, dtype=torch.float32)
        th = 2.0 * torch.atan2(p[:, 4], p[:, 6])
        c, s = torch.cos(th), torch.sin(th)
        z, o = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)
        d = torch.einsum("hwk,fjk->fhwj", d_local, rot)                            # (F,H,W,3)
        org = p[:, None, None, :3]
        lo = torch.tensor(ROOM_LO, device=device)
        hi = torch.tensor(ROOM_HI, device=device)
        t = torch.where(d > 0, (hi - org) / d, (lo - org) / d)
        t = torch.where(torch.isfinite(t), t, torch.full_like(t, float("inf")))
        depth = t.min(dim=-1).values + (torch.rand((n_frames, H, W), device=device, generator=gen) - 0.5) * 0.02
    else:
        raise ValueError(kind)
    r = torch.rand((n_frames, H, W), device=device, generator=gen)
    depth = torch.where(r < invalid_frac / 2, torch.zeros_like(depth), depth)
    depth = torch.where(r > 1 - invalid_frac / 2, torch.full_like(depth, 20.0), depth)
    return rgb.contiguous(), depth.float().contiguous(), poses
