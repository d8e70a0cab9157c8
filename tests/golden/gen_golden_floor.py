#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8f rank 3: `load_memory(..., load_single_floor)` and the feat.h5df layout.

Runs ONLY in the build container (needs /root/reference).  The reference's own `VoxelTokenMemory.load_memory`
(memory_2.py:166-256) is executed on memory directories written here in the reference's on-disk layout
(memory_2.py:1136-1145) with a stub simulator; the floor split it derives (DBSCAN over base_height, per-floor
z-ranges, the saved grid_rgb_pos_floor_k.npy) is stored as data in g8_floor_split.npz.

The same file records what the reference's `update_memory_dist_base` (memory_2.py:326-358) leaves in an HDF5
file, observed through the in-memory h5py stand-in (fake_h5py.py): group names in iteration order and the
dataset names / shapes / dtypes / resizability of each group, plus the token data — the contract
bsc_nav_amd.store's HDF5 adapter is tested against (tests/test_store_floor.py).
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fake_h5py  # noqa: E402
import gen_golden  # noqa: E402
import synth  # noqa: E402

GS, CS, FLOOR_H, MAP_H = 40, 0.1, -3.0, 3.0


def floor_cases():
    """(name, base_height list, current agent height, voxel z-extent [lo, hi])."""
    rs = np.random.RandomState(42)
    jit = lambda n, c, s=0.04: (c + rs.uniform(-s, s, size=n)).tolist()                      # noqa: E731
    return [
        ("one_floor", jit(12, 0.1), 0.1, (3, 55)),
        ("two_floors_low", jit(10, 0.05) + jit(9, 3.1), 0.2, (2, 57)),
        ("two_floors_high", jit(10, 0.05) + jit(9, 3.1), 2.9, (2, 57)),
        ("three_floors_mid_with_noise", jit(8, -2.9) + jit(8, 0.0) + jit(8, 2.8) + [1.4, -1.5], 0.3, (0, 59)),
        ("three_floors_top", jit(8, -2.9) + jit(8, 0.0) + jit(8, 2.8), 9.0, (0, 59)),
        ("few_samples", [0.0, 0.01, 3.0], 3.2, (5, 50)),                                       # len//5 == 0 -> min_samples 1
        ("unsorted_interleaved", [3.0, 0.0, 3.02, 0.03, -0.02, 2.97, 0.01, 3.01, 0.0, 2.99], -0.5, (1, 58)),
    ]


def write_memory_dir(path, rs, zlo, zhi, base_height):
    """A memory directory in the reference's layout (memory_2.py:1136-1145); > GS voxels (loader quirk :194)."""
    nh = int(MAP_H / CS) - int(FLOOR_H / CS)
    V = 600
    codes = rs.permutation(GS * GS * (zhi - zlo + 1))[:V]
    pos = np.stack([codes // (GS * (zhi - zlo + 1)), (codes // (zhi - zlo + 1)) % GS, zlo + codes % (zhi - zlo + 1)], 1).astype(np.int32)
    pos[0, 2], pos[1, 2] = zlo, zhi
    rgb = rs.randint(0, 255, size=(V, 3)).astype(np.uint8)
    w = rs.uniform(0.01, 5, size=V).astype(np.float32)
    occ = np.full((GS, GS, nh), -1, np.int32)
    occ[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(V)
    os.makedirs(path)
    np.save(path + "/grid_rgb_pos.npy", pos)
    np.save(path + "/grid_rgb.npy", rgb)
    np.save(path + "/weight.npy", w)
    np.save(path + "/occupied_ids.npy", occ)
    np.save(path + "/max_id.npy", np.array(V))
    np.save(path + "/original_pos.npy", np.array([1.0, 0.25, -2.0], np.float32))
    np.save(path + "/map_height.npy", np.array([int(FLOOR_H / CS), int(MAP_H / CS)]))
    np.save(path + "/base_height.npy", np.array(base_height))
    with open(path + "/long_memory.json", "w") as f:
        json.dump([{"label": "chair", "loc": [3, 4, int(zlo) + 1], "confidence": 0.9},
                   {"label": "sofa", "loc": [7, 8, int(zhi) - 1], "confidence": 0.8}], f)
    return pos, rgb, w


def main():
    out_dir = HERE
    ref_utils, ref_mem = gen_golden.import_reference("/root/reference")
    out = {}
    sink = io.StringIO()
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        for ci, (name, base_height, cur_h, (zlo, zhi)) in enumerate(floor_cases()):
            rs = np.random.RandomState(100 + ci)
            path = os.path.join(tmp, name)
            pos, rgb, w = write_memory_dir(path, rs, zlo, zhi, base_height)
            cfg = dict(gs=GS, cs=CS, floor_height=FLOOR_H, map_height=MAP_H, D=8, g=4, H=8, W=8, s=1)
            M = gen_golden.make_ref_memory(ref_utils, ref_mem, cfg, np.zeros((1, 4, 4, 8), np.float32), "mem://unused")
            M.args.load_single_floor = True
            M.args.load_memory_path = path
            M.Env = MagicMock()
            M.Env.agent.get_state.return_value = types.SimpleNamespace(position=np.array([0.0, cur_h, 0.0]))
            with contextlib.redirect_stdout(sink):
                M.load_memory(init_state=None, build_map=False)           # the reference's own method
            floor_files = sorted(f for f in os.listdir(path) if f.startswith("grid_rgb_pos_floor_"))
            assert len(floor_files) == 1
            k = int(floor_files[0].split("_")[-1].split(".")[0])
            names.append(name)
            out.update({
                f"{name}_base_height": np.array(base_height, np.float64), f"{name}_current_height": np.array(cur_h),
                f"{name}_pos": pos, f"{name}_rgb": rgb, f"{name}_weight": w,
                f"{name}_floor_heights": np.array(M.floor_heights, np.float64), f"{name}_num_floors": np.array(M.num_floors),
                f"{name}_current_floor": np.array(k),
                f"{name}_range": np.array([M.floor_min_height, M.floor_max_height], np.int64),
                f"{name}_floor_pos": np.load(path + f"/grid_rgb_pos_floor_{k}.npy"),
                f"{name}_floor_rgb": np.load(path + f"/grid_rgb_floor_{k}.npy"),
                f"{name}_long_memory_filtered": np.array([o["label"] for o in M.long_memory_filter()]),
                f"{name}_minh_maxh": np.array([M.minh, M.maxh], np.int64),
            })
            print(f"{name}: floors={M.floor_heights} current={k} range=[{M.floor_min_height},{M.floor_max_height}] "
                  f"kept {len(out[f'{name}_floor_pos'])}/{len(pos)}")
    out["cases"] = np.array(names)
    out["grid"] = np.array([GS, CS, FLOOR_H, MAP_H])

    # ---- feat.h5df as the reference writes it (through the stand-in) ----------------------------------------
    cfg = dict(gs=64, cs=0.1, floor_height=-2.0, map_height=4.4, seed=21, F=3, H=48, W=64, kind="room", g=8, D=12, s=3,
               iter_size=700)
    rgbf, depth, poses = synth.make_frames(cfg["seed"], cfg["F"], cfg["H"], cfg["W"], cfg["kind"])
    tokens = gen_golden.tag_tokens(synth.make_tokens(cfg["seed"], cfg["F"], cfg["g"], cfg["D"]))
    fp = "mem://g8_h5_layout"
    fake_h5py.File._stores.pop(fp, None)
    M = gen_golden.make_ref_memory(ref_utils, ref_mem, cfg, tokens, fp)
    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    for f in range(cfg["F"]):
        M._frame = f
        with contextlib.redirect_stdout(sink):
            M.obs2voxeltoken({"rgb": rgbf[f], "depth": depth[f]}, poses[f])
    with contextlib.redirect_stdout(sink):
        M.update_memory_dist_base()
    man = fake_h5py.manifest(fp)
    pos, cnt, feats, dists = gen_golden.dump_store(fp)
    ds_names = sorted({d for _, ds in man for d, _, _, _ in ds})
    assert all([d for d, _, _, _ in ds] == ds_names for _, ds in man)
    out.update({
        "h5_group_names": np.array([g for g, _ in man]),
        "h5_dataset_names": np.array(ds_names),
        "h5_dataset_dtypes": np.array(sorted({dt for _, ds in man for _, _, dt, _ in ds})),
        "h5_all_resizable": np.array(all(r for _, ds in man for _, _, _, r in ds)),
        "h5_feature_shapes": np.array([dict((d, s) for d, s, _, _ in ds)["features"] for _, ds in man], np.int64),
        "h5_distance_shapes": np.array([dict((d, s) for d, s, _, _ in ds)["distances"][0] for _, ds in man], np.int64),
        "h5_pos": pos, "h5_cnt": cnt, "h5_feats": feats.astype(np.float32), "h5_dists": dists.astype(np.float32),
    })
    print(f"h5 layout: {len(man)} groups, datasets {ds_names}, {len(feats)} tokens, first groups {[g for g, _ in man[:4]]}")
    p = os.path.join(out_dir, "g8_floor_split.npz")
    np.savez_compressed(p, **out)
    print(f"-> {p} {os.path.getsize(p) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
