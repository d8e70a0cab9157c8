"""Seeded synthetic RGB-D frames, poses and patch tokens.

Shared by the golden generator (tests/golden/gen_golden.py, runs only where
/root/reference exists), by the parity tests and by the CPU baseline leg of
bench.py.  Everything is derived from ``np.random.RandomState(seed)`` whose
stream is frozen by NumPy policy, so fixtures only need to store *outputs*
plus a checksum of the inputs generated here.

Conventions follow the simulator the reference wraps (env.py:166-235):
rgb is (H, W, 4) uint8 RGBA, depth is (H, W) float32 metres along the optical
axis, pose is (px, py, pz, qx, qy, qz, qw) with y up and yaw about +y.
"""
import hashlib

import numpy as np

ROOM_LO = np.array([-4.0, -1.5, -3.0])   # habitat world: x, y(up), z
ROOM_HI = np.array([4.0, 1.5, 3.0])


def _yaw_quat(theta):
    return np.array([0.0, np.sin(theta / 2.0), 0.0, np.cos(theta / 2.0)])


def make_poses(rs, F, start_yaw_steps=0):
    """Random walk: 0.25 m forward steps (args.py:33) and 30 degree turns (args.py:35)."""
    poses = np.zeros((F, 7), dtype=np.float64)
    pos = np.array([0.0, 0.0, 0.0])
    k = int(start_yaw_steps)
    for f in range(F):
        a = rs.randint(0, 3)
        if f > 0:
            if a == 0:
                th = k * np.pi / 6.0
                # habitat forward is -z rotated by yaw about +y
                step = np.array([-np.sin(th), 0.0, -np.cos(th)]) * 0.25
                nxt = pos + step
                if np.all(nxt > ROOM_LO + 0.6) and np.all(nxt < ROOM_HI - 0.6):
                    pos = nxt
            elif a == 1:
                k += 1
            else:
                k -= 1
        poses[f, :3] = pos
        poses[f, 3:] = _yaw_quat(k * np.pi / 6.0)
    return poses


def _room_depth(H, W, pose):
    """z-depth of an axis-aligned box room seen from `pose` (fov 90, fx from W)."""
    fx = W / 2.0
    u = (np.arange(W) + 0.5 - W / 2.0) / fx
    v = (np.arange(H) + 0.5 - H / 2.0) / fx
    uu, vv = np.meshgrid(u, v)
    # camera frame (x right, y down, z forward) -> habitat local (x, -y, -z)
    d_local = np.stack([uu, -vv, -np.ones_like(uu)], axis=-1)
    qy, qw = pose[4], pose[6]
    th = 2.0 * np.arctan2(qy, qw)
    c, s = np.cos(th), np.sin(th)
    rot = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    d = d_local @ rot.T
    o = pose[:3] + np.array([0.0, 0.0, 0.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (ROOM_LO - o) / d
        t_hi = (ROOM_HI - o) / d
    t = np.where(d > 0, t_hi, t_lo)
    t = np.where(np.isfinite(t), t, np.inf)
    return t.min(axis=-1)


def make_frames(seed, F, H, W, kind="room", start_yaw_steps=0, invalid_frac=0.02):
    """Return rgb (F,H,W,4) u8, depth (F,H,W) f32, poses (F,7) f64."""
    rs = np.random.RandomState(seed)
    poses = make_poses(rs, F, start_yaw_steps)
    rgb = rs.randint(0, 255, size=(F, H, W, 4)).astype(np.uint8)
    depth = np.empty((F, H, W), dtype=np.float32)
    for f in range(F):
        if kind == "room":
            d = _room_depth(H, W, poses[f])
            d = d + rs.uniform(-0.01, 0.01, size=d.shape)
        elif kind == "iid":
            d = rs.uniform(0.5, 5.0, size=(H, W))
        else:
            raise ValueError(kind)
        bad = rs.uniform(size=(H, W)) < invalid_frac
        far = rs.uniform(size=(H, W)) < 0.5
        d = np.where(bad, np.where(far, 20.0, 0.0), d)
        depth[f] = d.astype(np.float32)
    return rgb, depth, poses


def make_tokens(seed, F, g, D):
    """Per-frame patch tokens (F, g, g, D) f32, standing in for the ViT output."""
    rs = np.random.RandomState(seed + 7919)
    return rs.standard_normal(size=(F, g, g, D)).astype(np.float32)


def make_query_tokens(seed, B, T, D):
    rs = np.random.RandomState(seed + 104729)
    return rs.standard_normal(size=(B, T, D)).astype(np.float32)


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()
