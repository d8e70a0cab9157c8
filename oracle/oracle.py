"""ctypes front-end of the CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this module.  It wraps oracle/_build/libbsc_oracle.so (plain-C
restatement of the reference path, see bsc_oracle.h) and restates, with
NumPy/SciPy exactly as the reference does, the tiny host-side pieces the C
code takes as inputs: intrinsics (utils.py:144-150,181-186), the pose chain
(utils.py:133-141, memory_2.py:844-851,860) and the shuffled sub-sampling
(memory_2.py:747-749).
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np
from scipy.spatial.transform import Rotation as R

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbsc_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("bsc_oracle.c", "bsc_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class OrcConfig(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("grid_size", C.c_int32), ("min_h", C.c_int32),
                ("max_h", C.c_int32), ("patch_grid", C.c_int32), ("token_dim", C.c_int32), ("iter_size", C.c_int32),
                ("cache_size", C.c_int32), ("mode", C.c_int32), ("cell_size", C.c_double), ("min_depth", C.c_double),
                ("max_depth", C.c_double), ("K", C.c_double * 9), ("Kinv", C.c_double * 9), ("Kpatch", C.c_double * 9)]


DRAW_FN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig), C.c_int64]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_ingest_frame.restype = C.c_int64
        L.orc_ingest_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, DRAW_FN, C.c_void_p]
        L.orc_flush.argtypes = [C.c_void_p, DRAW_FN, C.c_void_p]
        L.orc_counters.argtypes = [C.c_void_p, C.c_void_p]
        for n in ("orc_export_rgb", "orc_export_cache"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_export_occupied.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_export_heightmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_export_store.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_export_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pool_query.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_localize.restype = C.c_int32
        L.orc_localize.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]
        L.orc_geometry.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_void_p] * 9
        L.orc_cluster_centers.restype = C.c_int32
        L.orc_cluster_centers.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frontier_mask.restype = None
        L.orc_frontier_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_frontier_clusters.restype = C.c_int32
        L.orc_frontier_clusters.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6
        L.orc_name_key.restype = C.c_uint64
        L.orc_name_key.argtypes = [C.c_int32] * 3
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- host-side restatements (NumPy/SciPy exactly as the reference) -------------------
def cam_mat_fov(h, w, fov=90):
    """utils.py:181-186 — fx = fy from the WIDTH; cx = w/2, cy = h/2."""
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / (2.0 * np.tan(np.deg2rad(fov / 2)))
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


def cam_mat_patch(h, w):
    """utils.py:144-150."""
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / 2.0
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


def pose_vec2tf(p):
    """utils.py:133-141."""
    tf = np.eye(4)
    tf[:3, 3] = np.asarray(p[:3]).flatten()
    tf[:3, :3] = R.from_quat(np.asarray(p[3:]).flatten()).as_matrix()
    return tf


BASE_TF = np.eye(4)
BASE_TF[0, :3] = [0, 0, -1]
BASE_TF[1, :3] = [-1, 0, 0]
BASE_TF[2, :3] = [0, 1, 0]


def base2cam(sensor_height=1.5):
    m = np.eye(4)
    m[:3, :3] = np.array([[1, 0, 0, 0, -1, 0, 0, 0, -1]]).reshape(3, 3)
    m[1, 3] = sensor_height
    return m


class PoseChain:
    """memory_2.py:844-851,860 — map frame anchored at the first ingested pose."""

    def __init__(self, sensor_height=1.5):
        self.inv_init = None
        self.b2c = base2cam(sensor_height)

    def pc_transform(self, pose):
        if self.inv_init is None:
            init = BASE_TF @ pose_vec2tf(pose) @ np.linalg.inv(BASE_TF)
            self.inv_init = np.linalg.inv(init)
        base = BASE_TF @ pose_vec2tf(pose) @ np.linalg.inv(BASE_TF)
        tf = self.inv_init @ base
        return np.ascontiguousarray(tf @ BASE_TF @ self.b2c)


def sample_indices(n_pixels, rate):
    """memory_2.py:747-749 — consumes the GLOBAL NumPy RNG like the reference."""
    idx = np.arange(n_pixels)
    np.random.shuffle(idx)
    return np.ascontiguousarray(idx[::rate].astype(np.int32))


def make_config(H, W, gs, cs, floor_height, map_height, g, D, iter_size=50000, cache_size=10, mode=0, min_depth=0.1,
                max_depth=10, fov=90):
    c = OrcConfig()
    c.height, c.width, c.grid_size = H, W, int(gs)
    c.max_h = int(map_height / cs)      # memory_2.py:122
    c.min_h = int(floor_height / cs)    # memory_2.py:123
    c.patch_grid, c.token_dim, c.iter_size, c.cache_size, c.mode = g, D, iter_size, cache_size, mode
    c.cell_size, c.min_depth, c.max_depth = cs, min_depth, max_depth
    K = cam_mat_fov(H, W, fov)
    c.K[:] = K.flatten()
    c.Kinv[:] = np.linalg.inv(K).flatten()     # utils.py:164
    c.Kpatch[:] = cam_mat_patch(g, g).flatten()
    return c


def py_draw(user, n):
    """memory_2.py:352."""
    return random.choice(range(n))


class OracleMemory:
    def __init__(self, cfg, voxel_capacity=None):
        self.cfg = cfg
        self.nh = cfg.max_h - cfg.min_h
        self.vcap = int(voxel_capacity or cfg.grid_size * cfg.grid_size)
        self.h = lib().orc_create(C.byref(cfg), self.vcap)
        self._draw = DRAW_FN(py_draw)
        self.chain = PoseChain()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def ingest_frame(self, depth, rgb, idx, T, tokens, alpha=None):
        depth = np.ascontiguousarray(depth, np.float32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        tokens = np.ascontiguousarray(tokens, np.float32)
        T = np.ascontiguousarray(T, np.float64)
        P = depth.size if idx is None else len(idx)
        return lib().orc_ingest_frame(self.h, _p(depth), _p(rgb), rgb.shape[-1], _p(idx), P, _p(T), _p(tokens),
                                      _p(alpha), self._draw, None)

    def flush(self):
        lib().orc_flush(self.h, self._draw, None)

    def counters(self):
        out = np.zeros(5, np.int64)
        lib().orc_counters(self.h, _p(out))
        return dict(max_id=int(out[0]), iter_id=int(out[1]), store_voxels=int(out[2]), store_tokens=int(out[3]),
                    flushes=int(out[4]))

    def export_rgb(self):
        n = self.counters()["max_id"]
        pos, rgb, w = np.zeros((n, 3), np.int32), np.zeros((n, 3), np.uint8), np.zeros(n, np.float32)
        lib().orc_export_rgb(self.h, _p(pos), _p(rgb), _p(w))
        return pos, rgb, w

    def export_occupied(self):
        occ = np.zeros((self.cfg.grid_size, self.cfg.grid_size, self.nh), np.int32)
        lib().orc_export_occupied(self.h, _p(occ))
        return occ

    def export_heightmap(self):
        gs = self.cfg.grid_size
        mh, cv = np.zeros((gs, gs), np.float64), np.zeros((gs, gs, 3), np.uint8)
        lib().orc_export_heightmap(self.h, _p(mh), _p(cv))
        return mh, cv

    def export_cache(self):
        n = self.counters()["iter_id"]
        f, p, d = np.zeros((n, self.cfg.token_dim), np.float32), np.zeros((n, 3), np.int32), np.zeros(n, np.float32)
        lib().orc_export_cache(self.h, _p(f), _p(p), _p(d))
        return f, p, d

    def export_store(self):
        c = self.counters()
        V, T = c["store_voxels"], c["store_tokens"]
        pos, cnt = np.zeros((V, 3), np.int32), np.zeros(V, np.int32)
        feats, dists = np.zeros((T, self.cfg.token_dim), np.float32), np.zeros(T, np.float32)
        lib().orc_export_store(self.h, _p(pos), _p(cnt), _p(feats), _p(dists))
        return pos, cnt, feats, dists

    def export_dense(self):
        n = self.counters()["max_id"]
        acc, cnt = np.zeros((n, self.cfg.token_dim), np.float32), np.zeros(n, np.int32)
        lib().orc_export_dense(self.h, _p(acc), _p(cnt))
        return acc, cnt

    def localize(self, q, K=100, radius=None, curr=None, floor=None):
        q = np.ascontiguousarray(q, np.float32).reshape(-1)
        pos, sim = np.zeros((K, 3), np.int32), np.zeros(K, np.float32)
        curr_a = np.ascontiguousarray(curr if curr is not None else [0, 0, 0], np.int32)
        lo, hi = floor if floor is not None else (0, -1)
        n = lib().orc_localize(self.h, _p(q), K, -1.0 if radius is None else float(radius), _p(curr_a), int(lo),
                               int(hi), _p(pos), _p(sim))
        return pos[:n], sim[:n]


def pool_query(tokens):
    tokens = np.ascontiguousarray(tokens, np.float32)
    B, T, D = tokens.shape
    out = np.zeros(D, np.float32)
    lib().orc_pool_query(_p(tokens), B, T, D, _p(out))
    return out


def geometry(cfg, depth, idx, T):
    depth = np.ascontiguousarray(depth, np.float32)
    P = depth.size if idx is None else len(idx)
    o = dict(valid=np.zeros(P, np.uint8), pc=np.zeros((P, 3)), pg=np.zeros((P, 3)), vox=np.zeros((P, 3), np.int32),
             in_range=np.zeros(P, np.uint8), pix=np.zeros((P, 2), np.int32), pat=np.zeros((P, 2), np.int32),
             r2=np.zeros(P), alpha=np.zeros(P))
    T = np.ascontiguousarray(T, np.float64)
    lib().orc_geometry(C.byref(cfg), _p(depth), _p(idx), P, _p(T), _p(o["valid"]), _p(o["pc"]), _p(o["pg"]),
                       _p(o["vox"]), _p(o["in_range"]), _p(o["pix"]), _p(o["pat"]), _p(o["r2"]), _p(o["alpha"]))
    return o


def name_key(r, c, h):
    return int(lib().orc_name_key(int(r), int(c), int(h)))


def cluster_centers(pos, sim, eps=10.0, min_samples=5):
    """BSCAgent.py:479-497 -> (centers (n,3) f64, labels (K,) i32, sizes (n,) i32)."""
    pos = np.ascontiguousarray(pos, np.int32)
    sim = np.ascontiguousarray(sim, np.float64)
    K = len(pos)
    centers, labels, sizes = np.zeros((K, 3)), np.zeros(K, np.int32), np.zeros(K, np.int32)
    n = lib().orc_cluster_centers(_p(pos), _p(sim), K, float(eps), int(min_samples), _p(centers), _p(labels), _p(sizes))
    return centers[:n], labels, sizes[:n]


def frontier_mask(cv_map, navigable=None):
    """memory_2.py:1165-1207 -> (gs,gs) u8: bit0 known, bit1 frontier."""
    cv = np.ascontiguousarray(cv_map, np.uint8)
    gs = cv.shape[0]
    nav = None if navigable is None else np.ascontiguousarray(navigable).astype(np.uint8)
    mask = np.zeros((gs, gs), np.uint8)
    lib().orc_frontier_mask(_p(cv), _p(nav), gs, _p(mask))
    return mask


def frontier_clusters(cv_map, frontier, min_cluster_size=10, ig_radius=5, max_clusters=None):
    """memory_2.py:1209-1311 -> dict(labels, first, sizes, centers, gains, best)."""
    cv = np.ascontiguousarray(cv_map, np.uint8)
    gs = cv.shape[0]
    fr = np.ascontiguousarray(frontier).astype(np.uint8)
    cap = int(max_clusters or gs * gs)
    labels = np.zeros((gs, gs), np.int32)
    first, sizes = np.zeros((cap, 2), np.int32), np.zeros(cap, np.int32)
    centers, gains, best = np.zeros((cap, 2)), np.zeros(cap), np.zeros(1, np.int32)
    n = lib().orc_frontier_clusters(_p(cv), _p(fr), gs, int(min_cluster_size), int(ig_radius), cap, _p(labels), _p(first),
                                    _p(sizes), _p(centers), _p(gains), _p(best))
    m = min(n, cap)
    return dict(n=n, labels=labels, first=first[:m], sizes=sizes[:m], centers=centers[:m], gains=gains[:m], best=int(best[0]))
