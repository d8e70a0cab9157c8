"""Seeded synthetic frame source on the device: stands in for the simulator
(env.py:166-235) in benchmarks and in the create_memory_for_dataset-shaped loop.

rgb (F,H,W,4) u8 RGBA, depth (F,H,W) f32 z-depth, poses (F,7) [px,py,pz,qx,qy,qz,qw].
`room`: analytic ray cast of an axis-aligned 8 x 3 x 6 m box from a camera on a seeded random walk
(0.25 m steps, 30 degree turns, args.py:33-35); `hall`: a 24 x 3 x 24 m hall with 36 square pillars (1 m, on a 4 m
lattice), the camera on a forward-biased walk that covers the whole floor — a scene of 10^5..10^6 surface voxels of
0.1 m, the size SURVEY.md §7 gives for real scans; `iid`: U(0.5, 5) m per pixel (worst case, one voxel
per point).  About 2 % of the pixels are pushed outside (min_depth, max_depth).  The walls, floor and ceiling of `room`
and `hall` lie exactly on boundaries of the 0.1 m voxel grid, so the +-1 cm depth noise flips every surface point between
two cells — the worst case for everything that works on runs of same-cell points; `room_off` is the same room moved by
3.7 cm along every axis (surfaces inside their cells), for comparison.
"""
import numpy as np
import torch

ROOM_LO = (-4.0, -1.5, -3.0)
ROOM_HI = (4.0, 1.5, 3.0)


HALL_LO = (-12.0, -1.5, -12.0)
HALL_HI = (12.0, 1.5, 12.0)
HALL_PILLARS = tuple((float(cx), float(cz)) for cx in range(-10, 11, 4) for cz in range(-10, 11, 4))   # centres, 1 x 1 m


def _hall_free(pos, margin=0.6):
    if not (HALL_LO[0] + margin < pos[0] < HALL_HI[0] - margin and HALL_LO[2] + margin < pos[2] < HALL_HI[2] - margin):
        return False
    return all(max(abs(pos[0] - cx), abs(pos[2] - cz)) > 0.5 + margin for cx, cz in HALL_PILLARS)


def hall_walk_poses(seed, n_frames):
    """Forward-biased walk (70 % forward steps of 0.25 m, 30 degree turns otherwise or when blocked)."""
    rs = np.random.RandomState(seed)
    poses = np.zeros((n_frames, 7), dtype=np.float64)
    pos, k = np.zeros(3), 0
    for f in range(n_frames):
        u = rs.uniform()
        if f > 0:
            th = k * np.pi / 6.0
            nxt = pos + np.array([-np.sin(th), 0.0, -np.cos(th)]) * 0.25
            if u < 0.7 and _hall_free(nxt):
                pos = nxt
            else:
                k += 1 if (u < 0.85 or u >= 0.925) else -1
        th = k * np.pi / 6.0
        poses[f, :3] = pos
        poses[f, 3:] = [0.0, np.sin(th / 2.0), 0.0, np.cos(th / 2.0)]
    return poses


ROOM_OFF = 0.037


def make_poses(kind, seed, n_frames):
    return hall_walk_poses(seed, n_frames) if kind == "hall" else random_walk_poses(seed, n_frames)


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
        poses = make_poses(kind, seed, n_frames)
    rgb = torch.randint(0, 255, (n_frames, H, W, 4), dtype=torch.uint8, device=device, generator=gen)
    if kind == "iid":
        depth = torch.rand((n_frames, H, W), device=device, generator=gen) * 4.5 + 0.5
    elif kind in ("room", "hall", "room_off"):
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
        lo = torch.tensor(HALL_LO if kind == "hall" else ROOM_LO, device=device) + (ROOM_OFF if kind == "room_off" else 0.0)
        hi = torch.tensor(HALL_HI if kind == "hall" else ROOM_HI, device=device) + (ROOM_OFF if kind == "room_off" else 0.0)
        t = torch.where(d > 0, (hi - org) / d, (lo - org) / d)
        t = torch.where(torch.isfinite(t), t, torch.full_like(t, float("inf")))
        t = t.min(dim=-1).values
        if kind == "hall":                     # nearest pillar face in front of the camera (slab test in x and z)
            inf = torch.full_like(t, float("inf"))
            dx, dz, ox, oz = d[..., 0], d[..., 2], org[..., 0], org[..., 2]
            for cx, cz in HALL_PILLARS:
                ax, bx = (cx - 0.5 - ox) / dx, (cx + 0.5 - ox) / dx
                az, bz = (cz - 0.5 - oz) / dz, (cz + 0.5 - oz) / dz
                t0 = torch.maximum(torch.minimum(ax, bx), torch.minimum(az, bz))
                t1 = torch.minimum(torch.maximum(ax, bx), torch.maximum(az, bz))
                hit = (t1 >= t0) & (t0 > 0)
                t = torch.minimum(t, torch.where(hit, t0, inf))
        depth = t + (torch.rand((n_frames, H, W), device=device, generator=gen) - 0.5) * 0.02
    else:
        raise ValueError(kind)
    r = torch.rand((n_frames, H, W), device=device, generator=gen)
    depth = torch.where(r < invalid_frac / 2, torch.zeros_like(depth), depth)
    depth = torch.where(r > 1 - invalid_frac / 2, torch.full_like(depth, 20.0), depth)
    return rgb.contiguous(), depth.float().contiguous(), poses
