#!/usr/bin/env python3
"""Generate golden vectors by running the *reference's own code* on seeded inputs.

Runs ONLY in the build container, where the read-only reference checkout is
mounted at /root/reference.  Nothing from the reference is copied: the module
is imported in-process (heavy third-party imports replaced by stubs, h5py by a
tiny in-memory stand-in whose key iteration is name-sorted like HDF5's), its
functions are executed on the inputs of tests/golden/synth.py and only the
resulting *data* is written to tests/golden/*.npz.

Usage:  python tests/golden/gen_golden.py [--ref /root/reference]

What each fixture pins:
  g1_geometry_*.npz   utils.depth2pc / transform_pc / base_pos2grid_id_3d / project_point
  g2_ingest_*.npz     VoxelTokenMemory.obs2voxeltoken state machine (memory_2.py:842-903)
  g3_flush_*.npz      VoxelTokenMemory.update_memory_dist_base (memory_2.py:326-358)
  g4_query_*.npz      VoxelTokenMemory.voxel_localized (memory_2.py:563-671)
"""
import argparse
import os
import random
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402


import fake_h5py  # noqa: E402  (in-memory h5py stand-in, name-ordered keys like HDF5)

_File = fake_h5py.File


def import_reference(ref):
    def stub(name):
        m = MagicMock(name=name)
        m.__spec__ = types.SimpleNamespace(name=name, loader=None, origin=None, submodule_search_locations=[])
        m.__path__ = []
        return m

    for name in ["cv2", "habitat_sim", "habitat_sim.utils", "habitat_sim.utils.common", "kneed", "open3d",
                 "diffusers", "ultralytics", "torchvision", "torchvision.transforms", "env", "magnum",
                 "habitat", "transformers", "matplotlib", "matplotlib.pyplot", "matplotlib.colors"]:
        sys.modules[name] = stub(name)
    fake_h5py.install()
    sys.path.insert(0, ref)
    import utils as ref_utils  # noqa
    import memory_2 as ref_mem  # noqa
    return ref_utils, ref_mem


# ----------------------------------------------------------------------------
# build a reference VoxelTokenMemory without running __init__ (needs hub/YOLO/habitat)
# ----------------------------------------------------------------------------
def make_ref_memory(ref_utils, ref_mem, cfg, tokens, feat_path):
    M = object.__new__(ref_mem.VoxelTokenMemory)
    M.args = types.SimpleNamespace(query_width=224, query_height=224, load_single_floor=cfg.get("single_floor", False),
                                   imagenary_num=3)
    M.device = "cpu"
    M.gs, M.cs = cfg["gs"], cfg["cs"]
    M.depth_sample_rate = cfg["s"]
    M.min_depth, M.max_depth = cfg.get("min_depth", 0.1), cfg.get("max_depth", 10)
    M.maxh = int(cfg["map_height"] / M.cs)
    M.minh = int(cfg["floor_height"] / M.cs)
    M.token_dim = cfg["D"]
    M.iter_size = cfg.get("iter_size", 50000)
    M.cache_size = 10
    M.n_patch_w = M.n_patch_h = cfg["g"]
    M.base_transform = np.eye(4)
    M.base_transform[0, :3] = [0, 0, -1]
    M.base_transform[1, :3] = [-1, 0, 0]
    M.base_transform[2, :3] = [0, 1, 0]
    M.base2cam_tf = np.eye(4)
    M.base2cam_tf[:3, :3] = np.array([[1, 0, 0, 0, -1, 0, 0, 0, -1]]).reshape((3, 3))
    M.base2cam_tf[1, 3] = cfg.get("sensor_height", 1.5)
    M.calib_mat = ref_utils.get_sim_cam_mat_with_fov(cfg["H"], cfg["W"], fov=90)
    M.cv_map = np.zeros((M.gs, M.gs, 3), dtype=np.uint8)
    M.max_height = np.full((M.gs, M.gs), -np.inf)
    M.inv_init_base_tf = []
    (M.grid_feat, M.grid_feat_pos, M.grid_rgb_pos, M.grid_feat_dis, M.weight, M.occupied_ids, M.grid_rgb,
     M.max_id, M.iter_id, M.base_height) = M._init_cache()
    M.feat_path = feat_path
    M._frame = 0
    import torch

    def fake_patch_token(img):
        t = torch.from_numpy(tokens[M._frame])
        return t

    M._get_patch_token = fake_patch_token
    M.get_total_token_count = lambda: None
    return M


def tag_tokens(tokens):
    """channel 0 carries the exact source id frame*g*g + py*g + px (exact in f32 below 2^24)."""
    F, g, _, D = tokens.shape
    ids = np.arange(F * g * g, dtype=np.float32).reshape(F, g, g)
    tokens = tokens.copy()
    tokens[..., 0] = ids
    return tokens


def dump_store(path):
    """feature store in HDF5 iteration (name-sorted) order."""
    st = _File._stores.get(path, {})
    keys = sorted(st.keys())
    pos = np.array([[int(x) for x in k.split("_")[1:4]] for k in keys], dtype=np.int32).reshape(-1, 3)
    cnt = np.array([st[k]["features"].shape[0] for k in keys], dtype=np.int32)
    feats = [st[k]["features"].a for k in keys]
    dists = [st[k]["distances"].a for k in keys]
    feats = np.concatenate(feats, 0) if feats else np.zeros((0, 1), np.float32)
    dists = np.concatenate(dists, 0) if dists else np.zeros((0,), np.float32)
    return pos, cnt, feats, dists


def run_ingest(ref_utils, ref_mem, name, cfg, out_dir, flush_end=True, queries=()):
    seed = cfg["seed"]
    rgb, depth, poses = synth.make_frames(seed, cfg["F"], cfg["H"], cfg["W"], cfg["kind"],
                                          start_yaw_steps=cfg.get("yaw0", 0))
    tokens = tag_tokens(synth.make_tokens(seed, cfg["F"], cfg["g"], cfg["D"]))
    feat_path = f"mem://{name}"
    _File._stores.pop(feat_path, None)
    M = make_ref_memory(ref_utils, ref_mem, cfg, tokens, feat_path)
    np.random.seed(seed)
    random.seed(seed)
    per_frame = []
    import io
    import contextlib
    sink = io.StringIO()
    for f in range(cfg["F"]):
        M._frame = f
        with contextlib.redirect_stdout(sink):
            M.obs2voxeltoken({"rgb": rgb[f], "depth": depth[f]}, poses[f])
        per_frame.append((M.iter_id, M.max_id))
    out = {
        "cfg_keys": np.array(sorted(cfg.keys())),
        "cfg_vals": np.array([str(cfg[k]) for k in sorted(cfg.keys())]),
        "input_sha": np.array(synth.checksum(rgb, depth, poses, tokens)),
        "per_frame_iter_max": np.array(per_frame, dtype=np.int64),
        # token cache before the final flush
        "iter_id": np.array(M.iter_id),
        "cache_pos": M.grid_feat_pos[:M.iter_id].copy(),
        "cache_src": M.grid_feat[:M.iter_id, 0].astype(np.int32),
        "cache_dis": M.grid_feat_dis[:M.iter_id].copy(),
        "cache_sha": np.array(synth.checksum(M.grid_feat[:M.iter_id])),
        # rgb voxel state
        "max_id": np.array(M.max_id),
        "grid_rgb_pos": M.grid_rgb_pos[:M.max_id].copy(),
        "grid_rgb": M.grid_rgb[:M.max_id].copy(),
        "weight": M.weight[:M.max_id].copy(),
        "occ_nnz": np.array(int((M.occupied_ids >= 0).sum())),
        # top-down map (sparse)
        "map_rc": np.argwhere(np.isfinite(M.max_height)).astype(np.int32),
    }
    rc = out["map_rc"]
    out["map_h"] = M.max_height[rc[:, 0], rc[:, 1]].astype(np.int32)
    out["map_rgb"] = M.cv_map[rc[:, 0], rc[:, 1]].copy()
    # occupied ids must be consistent with grid_rgb_pos
    p = out["grid_rgb_pos"]
    assert np.array_equal(M.occupied_ids[p[:, 0], p[:, 1], p[:, 2]], np.arange(M.max_id))
    if flush_end:
        with contextlib.redirect_stdout(sink):
            M.update_memory_dist_base()
        pos, cnt, feats, dists = dump_store(feat_path)
        out.update({
            "store_pos": pos, "store_cnt": cnt, "store_src": feats[:, 0].astype(np.int32),
            "store_dis": dists, "store_sha": np.array(synth.checksum(feats)),
        })
        for qi, q in enumerate(queries):
            import torch
            B = q.get("B", 1)
            T = q.get("T", 256)
            qtok = synth.make_query_tokens(seed + qi, B, T, cfg["D"])
            if q.get("from_store") is not None:
                # query built from a stored token => tie-heavy case (many voxels share one token)
                row = feats[q["from_store"] % max(1, len(feats))]
                qtok = np.broadcast_to(row, (B, T, cfg["D"])).copy()
                qtok += 0.01 * synth.make_query_tokens(seed + qi, B, T, cfg["D"])
            M.transform = lambda prompt, _q=qtok: torch.zeros(3, 4, 4)
            M.dinov2 = types.SimpleNamespace(
                forward_features=lambda x, _q=qtok: {"x_norm_patchtokens": torch.from_numpy(_q)})
            if q.get("floor") is not None:
                M.args.load_single_floor = True
                M.floor_min_height, M.floor_max_height = q["floor"]
            else:
                M.args.load_single_floor = False
            kw = {}
            if q.get("radius") is not None:
                kw = {"region_radius": q["radius"], "curr_grid": q["curr"]}
            with contextlib.redirect_stdout(sink):
                top1, tpos, tsim = M.voxel_localized(object(), K=q["K"], **kw)
            # pooled query (memory_2.py:591-608) recomputed by the same torch ops for the fixture
            tk = torch.from_numpy(qtok)
            g = int(np.sqrt(T))
            xs = torch.arange(g).repeat(g).view(1, T)
            ys = torch.arange(g).repeat_interleave(g).view(1, T)
            c = (g - 1) / 2
            d2 = (xs - c) ** 2 + (ys - c) ** 2
            w = torch.exp(-d2 / (2 * (g / 2) ** 2))
            w = (w / w.sum(dim=1, keepdim=True)).unsqueeze(-1)
            pooled = (tk * w).sum(dim=1).mean(dim=0).unsqueeze(0).numpy()
            out.update({
                f"q{qi}_B": np.array(B), f"q{qi}_T": np.array(T), f"q{qi}_K": np.array(q["K"]),
                f"q{qi}_radius": np.array(-1.0 if q.get("radius") is None else q["radius"]),
                f"q{qi}_curr": np.array(q.get("curr", [0, 0, 0])),
                f"q{qi}_floor": np.array(q.get("floor", [0, -1])),
                f"q{qi}_from_store": np.array(-1 if q.get("from_store") is None else q["from_store"]),
                f"q{qi}_pooled": pooled,
                f"q{qi}_top1": top1, f"q{qi}_pos": tpos, f"q{qi}_sim": tsim,
            })
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: frames={cfg['F']} iter_id={int(out['iter_id'])} max_id={M.max_id} "
          f"store_voxels={len(out.get('store_cnt', []))} -> {os.path.getsize(path)/1e3:.0f} kB")


def run_geometry(ref_utils, name, H, W, gs, cs, floor_h, map_h, g, seed, n_pick, out_dir, kind="iid"):
    rgb, depth, poses = synth.make_frames(seed, 2, H, W, kind, start_yaw_steps=1)
    d = depth[1]
    B = np.eye(4)
    B[0, :3] = [0, 0, -1]
    B[1, :3] = [-1, 0, 0]
    B[2, :3] = [0, 1, 0]
    b2c = np.eye(4)
    b2c[:3, :3] = np.array([[1, 0, 0, 0, -1, 0, 0, 0, -1]]).reshape(3, 3)
    b2c[1, 3] = 1.5
    init = B @ ref_utils.cvt_pose_vec2tf(poses[0]) @ np.linalg.inv(B)
    base = B @ ref_utils.cvt_pose_vec2tf(poses[1]) @ np.linalg.inv(B)
    tf = np.linalg.inv(init) @ base
    pc_tf = tf @ B @ b2c
    K = ref_utils.get_sim_cam_mat_with_fov(H, W, fov=90)
    Kp = ref_utils.get_sim_cam_mat(g, g)
    pc, mask = ref_utils.depth2pc(d, intr_mat=K, min_depth=0.1, max_depth=10)
    rs = np.random.RandomState(seed)
    N = H * W
    pick = np.sort(rs.choice(N, size=min(n_pick, N), replace=False)).astype(np.int32)
    pcs = pc[:, pick]
    pg = ref_utils.transform_pc(pcs, pc_tf)
    minh, maxh = int(floor_h / cs), int(map_h / cs)
    vox = np.zeros((len(pick), 3), np.int32)
    pix = np.zeros((len(pick), 2), np.int32)
    pat = np.zeros((len(pick), 2), np.int32)
    r2 = np.zeros(len(pick))
    al = np.zeros(len(pick))
    for i, (p, pl) in enumerate(zip(pg.T, pcs.T)):
        if not mask[pick[i]]:      # the reference drops these before the per-point loop (memory_2.py:750-752)
            continue
        vox[i] = ref_utils.base_pos2grid_id_3d(gs, cs, p[0], p[1], p[2])
        x, y, _ = ref_utils.project_point(K, pl)
        pix[i] = (x, y)
        x, y, _ = ref_utils.project_point(Kp, pl)
        pat[i] = (x, y)
        r2[i] = np.sum(np.square(pl))
        al[i] = np.exp(-r2[i] / (2 * 0.6))
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(
        path, H=H, W=W, gs=gs, cs=cs, minh=minh, maxh=maxh, g=g, seed=seed, kind=np.array(kind),
        input_sha=np.array(synth.checksum(d, poses)), pick=pick, mask=mask[pick], pc_tf=pc_tf, K=K,
        Kinv=np.linalg.inv(K), Kp=Kp, pc=pcs, pg=pg, vox=vox, pix=pix, pat=pat, r2=r2, alpha=al)
    shifted = int((pix[:, 0] != pick % W).sum())
    print(f"{name}: {len(pick)} points, pixel-column knife-edge shifts={shifted} -> {os.path.getsize(path)/1e3:.0f} kB")


class _FakeBoxes:
    def __init__(self, xyxy, conf, cls):
        import torch
        self.xyxy, self.conf, self.cls = torch.tensor(xyxy, dtype=torch.float32), torch.tensor(conf), torch.tensor(cls)

    def __len__(self):
        return len(self.conf)


def fake_detections(seed, f, H, W, n_cls):
    """Seeded stand-in for YOLO-World output (ultralytics Results[0].boxes): xyxy, conf, cls."""
    rs = np.random.RandomState(seed * 1000 + f)
    n = rs.randint(0, 6)
    x0 = rs.uniform(0, W - 40, size=n); y0 = rs.uniform(0, H - 40, size=n)
    x1 = x0 + rs.uniform(8, 39, size=n); y1 = y0 + rs.uniform(8, 39, size=n)
    return np.stack([x0, y0, x1, y1], 1).reshape(n, 4), rs.uniform(0.55, 0.99, size=n), rs.randint(0, n_cls, size=n)


def run_long_memory(ref_utils, ref_mem, name, cfg, out_dir):
    """memory_2.py:905-945,993-1025 — long_memory / long_memory_integration with seeded detections."""
    seed = cfg["seed"]
    rgb, depth, poses = synth.make_frames(seed, cfg["F"], cfg["H"], cfg["W"], cfg["kind"])
    tokens = tag_tokens(synth.make_tokens(seed, cfg["F"], cfg["g"], cfg["D"]))
    M = make_ref_memory(ref_utils, ref_mem, cfg, tokens, f"mem://{name}")
    classes = ["chair", "table", "sofa", "plant"]
    M.args.detect_conf, M.args.detect_classes, M.args.width = 0.55, classes, cfg["W"]
    M.long_memory_dict = []
    np.random.seed(seed)
    random.seed(seed)
    import io
    import contextlib
    sink = io.StringIO()
    per_frame = []
    for f in range(cfg["F"]):
        M._frame = f
        xyxy, conf, cls = fake_detections(seed, f, cfg["H"], cfg["W"], len(classes))
        M.yolow = types.SimpleNamespace(predict=lambda img, conf=None, _b=_FakeBoxes(xyxy, conf, cls): [types.SimpleNamespace(boxes=_b)])
        with contextlib.redirect_stdout(sink):
            M.obs2voxeltoken({"rgb": rgb[f], "depth": depth[f]}, poses[f])
            M.long_memory({"rgb": rgb[f], "depth": depth[f]})
        per_frame.append(len(M.long_memory_dict))
    lm = M.long_memory_dict
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"),
                        cfg_keys=np.array(sorted(cfg.keys())), cfg_vals=np.array([str(cfg[k]) for k in sorted(cfg.keys())]),
                        input_sha=np.array(synth.checksum(rgb, depth, poses, tokens)), classes=np.array(classes),
                        per_frame=np.array(per_frame), label=np.array([classes.index(o["label"]) for o in lm]),
                        loc=np.array([o["loc"] for o in lm], dtype=np.int64).reshape(-1, 3),
                        confidence=np.array([o["confidence"] for o in lm], dtype=np.float64))
    print(f"{name}: {len(lm)} objects after {cfg['F']} frames, per-frame counts {per_frame}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    ref_utils, ref_mem = import_reference(a.ref)

    run_geometry(ref_utils, "g1_geometry_320x240", 240, 320, 128, 0.1, -2.0, 4.4, 14, 11, 12000, a.out)
    run_geometry(ref_utils, "g1_geometry_640x480", 480, 640, 256, 0.1, -12.8, 12.8, 16, 12, 12000, a.out, kind="room")
    run_geometry(ref_utils, "g1_geometry_680x680", 680, 680, 1000, 0.1, -10.0, 10.0, 16, 13, 12000, a.out)

    base = dict(gs=128, cs=0.1, floor_height=-2.0, map_height=4.4)
    qs = [dict(K=5), dict(K=100, B=3), dict(K=20, radius=25.0, curr=[64, 60, 30]), dict(K=50, floor=[18, 34]),
          dict(K=40, from_store=123), dict(K=100, B=3, from_store=4567, radius=40.0, curr=[60, 64, 28], floor=[10, 40])]
    run_ingest(ref_utils, ref_mem, "g2_mini_s1", dict(base, seed=1, F=6, H=48, W=64, kind="room", g=16, D=16, s=1),
               a.out, queries=qs)
    run_ingest(ref_utils, ref_mem, "g2_mini_s7_yaw", dict(base, seed=2, F=8, H=48, W=64, kind="room", g=16, D=16, s=7, yaw0=2),
               a.out, queries=qs[:2])
    run_ingest(ref_utils, ref_mem, "g2_c1_s50_iid", dict(base, seed=3, F=8, H=240, W=320, kind="iid", g=14, D=32, s=50),
               a.out, queries=qs[:3])
    run_ingest(ref_utils, ref_mem, "g2_c1_s1000", dict(base, seed=4, F=12, H=240, W=320, kind="room", g=14, D=32, s=1000),
               a.out, queries=qs[:2])
    run_long_memory(ref_utils, ref_mem, "g6_long_memory", dict(base, seed=8, F=24, H=240, W=320, kind="room", g=14, D=8, s=1000),
                    a.out)
    # small token cache: in-loop flushes, dropped trigger tokens, >10 tokens per voxel with random replacement
    run_ingest(ref_utils, ref_mem, "g3_flush_small_cache",
               dict(base, seed=5, F=6, H=48, W=64, kind="room", g=16, D=16, s=1, iter_size=1500), a.out, queries=qs[:2] + qs[4:5])
    run_ingest(ref_utils, ref_mem, "g3_flush_640x480_s97",
               dict(gs=256, cs=0.1, floor_height=-12.8, map_height=12.8, seed=6, F=5, H=480, W=640, kind="room", g=14,
                    D=24, s=97, iter_size=4000), a.out, queries=qs[:1])


if __name__ == "__main__":
    main()
