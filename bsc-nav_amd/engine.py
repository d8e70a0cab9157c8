"""VoxelEngine — thin Python handle on one libbscnav context (one GPU).

All arithmetic of the path runs in the HIP library; this class only marshals torch device tensors
(raw data_ptr) and NumPy host buffers across the C-ABI of include/bscnav.h.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib
from .geometry import cam_mat_fov, cam_mat_patch


def _hp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class VoxelEngine:
    def __init__(self, height, width, grid_size, cell_size, floor_height, map_height, patch_grid, token_dim,
                 mode="exact", iter_size=50000, cache_size=10, voxel_capacity=None, token_capacity=None,
                 max_points=None, device=0, fov=90, min_depth=0.1, max_depth=10, min_h=None, max_h=None):
        if not torch.cuda.is_available():
            raise RuntimeError("bsc_nav_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU path")
        self.lib = _lib.load()
        c = _lib.BscConfig()
        c.height, c.width, c.grid_size = int(height), int(width), int(grid_size)
        c.max_h = int(map_height / cell_size) if max_h is None else int(max_h)      # memory_2.py:122
        c.min_h = int(floor_height / cell_size) if min_h is None else int(min_h)    # memory_2.py:123
        c.patch_grid, c.token_dim = int(patch_grid), int(token_dim)
        c.iter_size, c.cache_size, c.mode = int(iter_size), int(cache_size), _lib.MODES[mode]
        c.voxel_capacity = int(voxel_capacity or int(grid_size) * int(grid_size))   # memory_2.py:715
        c.max_points = int(max_points or height * width)
        c.token_capacity = int(token_capacity or (c.voxel_capacity * 2 + iter_size)) if mode == "exact" else 0
        c.cell_size, c.min_depth, c.max_depth = float(cell_size), float(min_depth), float(max_depth)
        K = cam_mat_fov(height, width, fov)
        c.K[:] = K.flatten()
        c.Kinv[:] = np.linalg.inv(K).flatten()              # utils.py:164
        c.Kpatch[:] = cam_mat_patch(patch_grid, patch_grid).flatten()
        self.cfg = c
        self.mode = mode
        self.device = torch.device("cuda", device)
        self.nh = c.max_h - c.min_h
        torch.cuda.set_device(self.device)
        # The library launches on the stream that is current NOW; calls made later under another torch stream are
        # ordered against it in _enter() (see there).
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        _lib.check(self.lib.bsc_create(C.byref(c), device, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h
        self.log_capacity = 0
        self._draw = _lib.DRAW_FN(self._draw_cb)

    # memory_2.py:352 — Python's global RNG, one draw per row that meets a full voxel
    def _draw_cb(self, user, n, out):
        k = self.cfg.cache_size
        if n < 64:
            for i in range(n):
                out[i] = random.choice(range(k))
            return
        # many draws (saturated voxels): the same stream, advanced by the library's restatement of random.choice
        version, internal, gauss = random.getstate()
        key = np.array(internal[:624], dtype=np.uint32)
        pos = C.c_int32(internal[624])
        _lib.check(self.lib.bsc_host_choice_draws(key.ctypes.data_as(C.c_void_p), C.byref(pos), k, n,
                                                  C.cast(out, C.c_void_p)))
        random.setstate((version, tuple(key.tolist()) + (pos.value,), gauss))

    def _enter(self, *tensors):
        """Order the library stream after the caller's current stream and pin the inputs to it: when the caller works
        under a different torch stream than the one the engine was created on, the kernels must not start before
        the producers of `tensors` have finished, and the caching allocator must not hand their blocks out again
        while the library is still reading them."""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.stream)

    def _leave(self):
        """Results written by the library (device outputs) become visible to the caller's current stream."""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.bsc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _lib.check(self.lib.bsc_reset(self.h))

    def stream_wait_chain(self, stream):
        """`stream` (torch.cuda.Stream) waits on the GPU for the rgb chain kernels launched so far."""
        _lib.check(self.lib.bsc_stream_wait_chain(self.h, C.c_void_p(stream.cuda_stream)))

    def sync(self):
        """Everything ingest() has started or deferred (the rgb chain of the last call is launched lazily) is complete."""
        _lib.check(self.lib.bsc_sync(self.h))

    def ingest(self, depth, rgb, tokens, transforms, sample_idx=None, offsets=None, alpha=None):
        """depth (F,H,W) f32, rgb (F,H,W,C) u8, tokens (F,g,g,D) f32 or bf16 (widened exactly): contiguous CUDA
        tensors.  transforms (F,4,4) float64 NumPy.  sample_idx int32 CUDA + offsets (F+1) int64 NumPy, or None."""
        F = depth.shape[0] if depth.dim() == 3 else 1
        assert depth.is_cuda and rgb.is_cuda and tokens.is_cuda
        assert depth.dtype == torch.float32 and rgb.dtype == torch.uint8 and tokens.dtype in (torch.float32, torch.bfloat16)
        assert depth.is_contiguous() and rgb.is_contiguous() and tokens.is_contiguous()
        T = np.ascontiguousarray(np.asarray(transforms, dtype=np.float64).reshape(F, 16))
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        if sample_idx is not None:
            assert sample_idx.is_cuda and sample_idx.dtype == torch.int32 and off is not None and len(off) == F + 1
        if alpha is not None:
            assert alpha.is_cuda and alpha.dtype == torch.float64
        self._enter(depth, rgb, tokens, sample_idx, alpha)
        _lib.check(self.lib.bsc_ingest_typed(self.h, F, _dp(depth), _dp(rgb), rgb.shape[-1], _dp(tokens),
                                             1 if tokens.dtype == torch.bfloat16 else 0, _hp(T), _dp(sample_idx),
                                             _hp(off), _dp(alpha), self._draw, None))

    def flush(self):
        _lib.check(self.lib.bsc_flush(self.h, self._draw, None))

    def counters(self):
        out = np.zeros(10, np.int64)
        _lib.check(self.lib.bsc_counters(self.h, _hp(out)))
        keys = ["max_id", "iter_id", "store_voxels", "store_tokens", "flushes", "points_passed", "points_seen",
                "voxel_rmw", "pairs", "pairs_last_call"]
        return {k: int(v) for k, v in zip(keys, out)}

    def geometry(self, depth, transform, sample_idx=None):
        P = depth.numel() if sample_idx is None else sample_idx.numel()
        o = dict(flags=np.zeros(P, np.uint8), pc=np.zeros((P, 3)), pg=np.zeros((P, 3)), vox=np.zeros((P, 3), np.int32),
                 pix=np.zeros((P, 2), np.int32), pat=np.zeros((P, 2), np.int32), r2=np.zeros(P), alpha=np.zeros(P))
        T = np.ascontiguousarray(np.asarray(transform, np.float64).reshape(16))
        _lib.check(self.lib.bsc_geometry(self.h, _dp(depth), _hp(T), _dp(sample_idx), P, _hp(o["flags"]), _hp(o["pc"]),
                                         _hp(o["pg"]), _hp(o["vox"]), _hp(o["pix"]), _hp(o["pat"]), _hp(o["r2"]),
                                         _hp(o["alpha"])))
        return o

    def sort_pairs_u32(self, keys, vals, begin_bit=0, end_bit=32):
        """The library's stable radix sort of (u32 key, u32 value) pairs on the key bits [begin_bit, end_bit): int32 CUDA tensors
        (bit patterns) in, sorted copies out."""
        ko, vo = torch.empty_like(keys), torch.empty_like(vals)
        _lib.check(self.lib.bsc_sort_pairs_u32(self.h, _dp(keys), _dp(vals), keys.numel(), begin_bit, end_bit, _dp(ko), _dp(vo)))
        return ko, vo

    # ---- exports / imports -------------------------------------------------------------------
    def export_rgb(self):
        n = self.counters()["max_id"]
        pos, rgb, w = np.zeros((n, 3), np.int32), np.zeros((n, 3), np.uint8), np.zeros(n, np.float32)
        _lib.check(self.lib.bsc_export_rgb(self.h, _hp(pos), _hp(rgb), _hp(w)))
        return pos, rgb, w

    def export_occupied(self):
        occ = np.zeros((self.cfg.grid_size, self.cfg.grid_size, self.nh), np.int32)
        _lib.check(self.lib.bsc_export_occupied(self.h, _hp(occ)))
        return occ

    def export_heightmap(self):
        gs = self.cfg.grid_size
        mh, cv = np.zeros((gs, gs), np.float64), np.zeros((gs, gs, 3), np.uint8)
        _lib.check(self.lib.bsc_export_heightmap(self.h, _hp(mh), _hp(cv)))
        return mh, cv

    def export_cache(self):
        n = self.counters()["iter_id"]
        f, p, d = np.zeros((n, self.cfg.token_dim), np.float32), np.zeros((n, 3), np.int32), np.zeros(n, np.float32)
        _lib.check(self.lib.bsc_export_cache(self.h, _hp(f), _hp(p), _hp(d)))
        return f, p, d

    def export_store(self):
        c = self.counters()
        V, T = c["store_voxels"], c["store_tokens"]
        pos, cnt = np.zeros((V, 3), np.int32), np.zeros(V, np.int32)
        feats, dists = np.zeros((T, self.cfg.token_dim), np.float32), np.zeros(T, np.float32)
        _lib.check(self.lib.bsc_export_store(self.h, _hp(pos), _hp(cnt), _hp(feats), _hp(dists)))
        return pos, cnt, feats, dists

    def export_dense(self):
        n = self.counters()["max_id"]
        acc, cnt = np.zeros((n, self.cfg.token_dim), np.float32), np.zeros(n, np.int32)
        _lib.check(self.lib.bsc_export_dense(self.h, _hp(acc), _hp(cnt)))
        return acc, cnt

    def import_rgb(self, pos, rgb, weight):
        pos = np.ascontiguousarray(pos, np.int32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        weight = np.ascontiguousarray(weight, np.float32)
        _lib.check(self.lib.bsc_import_rgb(self.h, len(pos), _hp(pos), _hp(rgb), _hp(weight)))

    def import_store(self, pos, cnt, feats, dists):
        pos = np.ascontiguousarray(pos, np.int32)
        cnt = np.ascontiguousarray(cnt, np.int32)
        feats = np.ascontiguousarray(feats, np.float32)
        dists = np.ascontiguousarray(dists, np.float32)
        _lib.check(self.lib.bsc_import_store(self.h, len(pos), len(feats), _hp(pos), _hp(cnt), _hp(feats), _hp(dists)))

    def import_dense(self, acc, cnt):
        acc = np.ascontiguousarray(acc, np.float32)
        cnt = np.ascontiguousarray(cnt, np.int32)
        _lib.check(self.lib.bsc_import_dense(self.h, len(cnt), _hp(acc), _hp(cnt)))

    # ---- query ------------------------------------------------------------------------------------
    def pool_query(self, tokens):
        """tokens (B,T,D) f32 CUDA -> (D) f32 CUDA   (memory_2.py:591-608)"""
        assert tokens.is_cuda and tokens.dtype == torch.float32 and tokens.is_contiguous()
        B, T, D = tokens.shape
        out = torch.empty(D, dtype=torch.float32, device=tokens.device)
        self._enter(tokens, out)
        _lib.check(self.lib.bsc_pool_query(self.h, _dp(tokens), B, T, D, _dp(out)))
        self._leave()
        return out

    def localize(self, q, K=100, radius=None, curr=None, floor=None):
        """q (Q,D) or (D) f32 CUDA -> (pos (Q,n,3) int32, sim (Q,n) f32, counts)."""
        q = q.reshape(-1, self.cfg.token_dim).contiguous()
        assert q.is_cuda and q.dtype == torch.float32
        Q = q.shape[0]
        pos, sim, cnt = np.zeros((Q, K, 3), np.int32), np.zeros((Q, K), np.float32), np.zeros(Q, np.int32)
        curr_a = None if curr is None else np.ascontiguousarray(curr, np.int32)
        lo, hi = (0, -1) if floor is None else (int(floor[0]), int(floor[1]))
        self._enter(q)
        _lib.check(self.lib.bsc_localize(self.h, _dp(q), Q, K, -1.0 if radius is None else float(radius), _hp(curr_a),
                                         lo, hi, _hp(pos), _hp(sim), _hp(cnt)))
        return pos, sim, cnt

    def cluster_centers(self, pos=None, sim=None, K=None, query_index=0, eps=10.0, min_samples=5):
        """BSCAgent.weighted_cluster_centers on the GPU -> (centers (n,3) f64, labels (K,) int, sizes list).
        pos/sim None: cluster the first K results of query `query_index` of the last localize call (no host copy)."""
        if pos is not None:
            pos = np.ascontiguousarray(pos, np.int32)
            sim = np.ascontiguousarray(sim, np.float32)
            K = len(pos)
        centers, labels, sizes = np.zeros((K, 3), np.float64), np.zeros(K, np.int32), np.zeros(K, np.int32)
        n = np.zeros(1, np.int32)
        _lib.check(self.lib.bsc_cluster_centers(self.h, query_index, K, _hp(pos), _hp(sim), float(eps), int(min_samples),
                                                _hp(centers), _hp(labels), _hp(sizes), _hp(n)))
        return centers[:n[0]], labels.astype(np.int64), [int(v) for v in sizes[:n[0]]]

    # ---- FrontierExplorer helpers (memory_2.py:1147-1311) ----------------------------------------------
    def frontier_mask(self, navigable=None):
        """(gs,gs) u8: bit0 known (cv_map.sum(-1) != 0), bit1 frontier (known, navigable, an unknown 4-neighbour)."""
        gs = self.cfg.grid_size
        nav = None if navigable is None else np.ascontiguousarray(np.asarray(navigable) != 0, np.uint8)
        mask = np.zeros((gs, gs), np.uint8)
        _lib.check(self.lib.bsc_frontier_mask(self.h, _hp(nav), _hp(mask)))
        return mask

    def frontier_clusters(self, frontier=None, min_cluster_size=10, ig_radius=5, max_clusters=4096, labels=True):
        """4-connected frontier clusters in the reference's order -> dict(n, first, sizes, centers, gains, best, labels).
        frontier (gs,gs) nonzero = frontier cell; None = the cells of the last frontier_mask call."""
        gs = self.cfg.grid_size
        fr = None if frontier is None else np.ascontiguousarray(np.asarray(frontier) != 0, np.uint8)
        cap = int(max_clusters)
        n, best = np.zeros(1, np.int32), np.zeros(1, np.int32)
        lab = np.zeros((gs, gs), np.int32) if labels else None
        first, sizes = np.zeros((cap, 2), np.int32), np.zeros(cap, np.int32)
        centers, gains = np.zeros((cap, 2), np.float64), np.zeros(cap, np.float64)
        _lib.check(self.lib.bsc_frontier_clusters(self.h, _hp(fr), int(min_cluster_size), int(ig_radius), cap, _hp(n),
                                                  _hp(lab), _hp(first), _hp(sizes), _hp(centers), _hp(gains), _hp(best)))
        m = min(int(n[0]), cap)
        return dict(n=int(n[0]), first=first[:m], sizes=sizes[:m], centers=centers[:m], gains=gains[:m],
                    best=int(best[0]), labels=lab)

    def import_cv_map(self, cv_map):
        cv = np.ascontiguousarray(cv_map, np.uint8)
        assert cv.shape == (self.cfg.grid_size, self.cfg.grid_size, 3)
        _lib.check(self.lib.bsc_import_cv_map(self.h, _hp(cv)))

    def kernel_stats(self, which=0, reset=False):
        """HIP-event time of the dominant kernel: dict(ms, launches, bytes, launches_since_reset)."""
        out = np.zeros(4, np.float64)
        _lib.check(self.lib.bsc_kernel_stats(self.h, which, 1 if reset else 0, _hp(out)))
        return dict(ms=float(out[0]), launches=int(out[1]), bytes=float(out[2]), launches_since_reset=int(out[3]))

    # ---- multi-GPU helpers --------------------------------------------------------------------------
    def keys_tensor(self):
        """(max_id,3) int32 CUDA view of the voxel keys in id order (copy)."""
        ptr, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.bsc_keys_dev(self.h, C.byref(ptr), C.byref(n)))
        out = torch.empty((n.value, 3), dtype=torch.int32, device=self.device)
        if n.value:
            pos, _, _ = self.export_rgb()
            out.copy_(torch.from_numpy(pos))
        return out

    def max_height_cv_map(self):
        """(max_height (gs,gs) f64 with -inf for empty cells, cv_map (gs,gs,3) u8) — the top-down map state."""
        return self.export_heightmap()

    def dense_gather(self, keys):
        keys = keys.contiguous()
        n = keys.shape[0]
        acc = torch.empty((n, self.cfg.token_dim), dtype=torch.float32, device=self.device)
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        self._enter(keys, acc, cnt)
        _lib.check(self.lib.bsc_dense_gather(self.h, n, _dp(keys), _dp(acc), _dp(cnt)))
        self._leave()
        return acc, cnt

    def dense_gather_rgb(self, keys):
        """rgb (n,3) u8 and weight (n,) f32 of the voxels `keys` (weight 0 where this map has no such voxel)."""
        keys = keys.contiguous()
        n = keys.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.uint8, device=self.device)
        w = torch.empty(n, dtype=torch.float32, device=self.device)
        self._enter(keys, rgb, w)
        _lib.check(self.lib.bsc_dense_gather_rgb(self.h, n, _dp(keys), _dp(rgb), _dp(w)))
        self._leave()
        return rgb, w

    def dense_replace(self, keys, acc, cnt, rgb=None, weight=None):
        """The map becomes exactly these voxels (ids in the given order); rgb / weight None -> zeroed colours."""
        keys, acc, cnt = keys.contiguous(), acc.contiguous(), cnt.contiguous()
        assert keys.dtype == torch.int32 and acc.dtype == torch.float32 and cnt.dtype == torch.int32
        if rgb is not None:
            rgb, weight = rgb.contiguous(), weight.contiguous()
            assert rgb.dtype == torch.uint8 and weight.dtype == torch.float32 and rgb.shape[0] == keys.shape[0]
        self._enter(keys, acc, cnt, rgb, weight)
        _lib.check(self.lib.bsc_dense_replace_full(self.h, keys.shape[0], _dp(keys), _dp(acc), _dp(cnt), _dp(rgb), _dp(weight)))

    # ---- exact colour across ranks: point log + replay (include/bscnav.h bsc_point_log_*, bsc_replay_colour) ----------
    def point_log_enable(self, capacity):
        """Keep (cell, alpha, rgb) of every ingested point — 16 B / point, for the sub-sampled modes; 0 disables."""
        _lib.check(self.lib.bsc_point_log_enable(self.h, int(capacity)))
        self.log_capacity = int(capacity)

    def point_log(self):
        """-> (cells (n,) int32, records (n,3) int32 [alpha lo, alpha hi, rgb]) CUDA copies of the log, order of ingestion."""
        n = C.c_int64()
        _lib.check(self.lib.bsc_point_log_read(self.h, None, None, 0, C.byref(n)))
        cells = torch.empty(n.value, dtype=torch.int32, device=self.device)
        recs = torch.empty((n.value, 3), dtype=torch.int32, device=self.device)
        self._enter(cells, recs)
        _lib.check(self.lib.bsc_point_log_read(self.h, _dp(cells), _dp(recs), n.value, C.byref(n)))
        self._leave()
        return cells, recs

    def replay_colour(self, vox_sorted, records, n_vox):
        """Records (n,3) int32 grouped by voxel (vox_sorted (n,) int32 ascending in [0, n_vox)), every voxel's in global point
        order -> (rgb (n_vox,3) u8, weight (n_vox,) f32): the sequential chain of memory_2.py:888-899 from the empty state."""
        vox_sorted, records = vox_sorted.contiguous(), records.contiguous()
        assert vox_sorted.dtype == torch.int32 and records.dtype == torch.int32 and records.shape == (vox_sorted.numel(), 3)
        rgb = torch.zeros((n_vox, 3), dtype=torch.uint8, device=self.device)
        w = torch.zeros(n_vox, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.bsc_replay_colour(vox_sorted.numel(), _dp(vox_sorted), _dp(records), int(n_vox), _dp(rgb), _dp(w),
                                              C.c_void_p(st.cuda_stream)))
        return rgb, w

    def import_heightmap(self, max_height, cv_map):
        gs = self.cfg.grid_size
        mh = np.ascontiguousarray(max_height, np.float64)
        cv = np.ascontiguousarray(cv_map, np.uint8)
        assert mh.shape == (gs, gs) and cv.shape == (gs, gs, 3)
        _lib.check(self.lib.bsc_import_heightmap(self.h, _hp(mh), _hp(cv)))
