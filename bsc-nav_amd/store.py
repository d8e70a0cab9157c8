"""On-disk layout of a memory directory (memory_2.py:1136-1145 save, :189-200 load).

    grid_rgb_pos.npy (max_id,3) i32 | grid_rgb.npy (max_id,3) u8 | weight.npy (max_id,) f32
    occupied_ids.npy (gs,gs,maxh-minh) i32, -1 empty | max_id.npy 0-d int | original_pos.npy (3,) f32
    map_height.npy [minh,maxh] | base_height.npy (n,) | long_memory.json
    feat.h5df   HDF5 groups grid_{r}_{c}_{h} -> features (M,D) f32, distances (M,) f32   (memory_2.py:330-354)

h5py is not part of this image, so the token store is ALSO written as four flat arrays holding exactly the
same content in HDF5 iteration (name) order; `feat.h5df` itself is written / read when h5py is importable.
    feat_voxel_keys.npy (V,3) i32 | feat_token_offsets.npy (V+1,) i64 | feat_features.npy (T,D) f32 |
    feat_distances.npy (T,) f32
Dense-mode maps add dense_acc.npy (max_id,D) f32 (sum or max) and dense_cnt.npy (max_id,) i32.
"""
import json
import os

import numpy as np

def _h5py():
    """The h5py module, or None.  Looked up at call time: the adapter is optional (this image ships without h5py;
    the tests exercise it through an in-memory stand-in registered under the same name)."""
    try:
        import h5py
        return h5py
    except Exception:
        return None


def have_h5py():
    return _h5py() is not None


def save_rgb_state(path, pos, rgb, weight, occupied, max_id, original_pos, minh, maxh, base_height, long_memory):
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "grid_rgb_pos.npy"), pos)
    np.save(os.path.join(path, "grid_rgb.npy"), rgb)
    np.save(os.path.join(path, "weight.npy"), weight)
    np.save(os.path.join(path, "occupied_ids.npy"), occupied)
    np.save(os.path.join(path, "max_id.npy"), np.array(max_id))
    np.save(os.path.join(path, "original_pos.npy"), np.asarray(original_pos, dtype=np.float32))
    np.save(os.path.join(path, "map_height.npy"), np.array([minh, maxh]))
    np.save(os.path.join(path, "base_height.npy"), np.array(base_height))
    with open(os.path.join(path, "long_memory.json"), "w") as f:
        json.dump(long_memory, f, indent=4)


def load_rgb_state(path):
    out = dict(
        max_id=int(np.load(os.path.join(path, "max_id.npy"))),
        pos=np.load(os.path.join(path, "grid_rgb_pos.npy")),
        rgb=np.load(os.path.join(path, "grid_rgb.npy")),
        weight=np.load(os.path.join(path, "weight.npy")),
        original_pos=np.load(os.path.join(path, "original_pos.npy")),
    )
    out["minh"], out["maxh"] = (int(v) for v in np.load(os.path.join(path, "map_height.npy")))
    lm = os.path.join(path, "long_memory.json")
    out["long_memory"] = json.load(open(lm)) if os.path.exists(lm) else []
    bh = os.path.join(path, "base_height.npy")
    out["base_height"] = np.load(bh) if os.path.exists(bh) else np.zeros(0)
    return out


def save_token_store(path, pos, cnt, feats, dists, write_h5=True):
    off = np.zeros(len(cnt) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    np.save(os.path.join(path, "feat_voxel_keys.npy"), pos.astype(np.int32))
    np.save(os.path.join(path, "feat_token_offsets.npy"), off)
    np.save(os.path.join(path, "feat_features.npy"), feats.astype(np.float32))
    np.save(os.path.join(path, "feat_distances.npy"), dists.astype(np.float32))
    if write_h5 and have_h5py():
        write_h5_store(os.path.join(path, "feat.h5df"), pos, cnt, feats, dists)


def write_h5_store(h5_path, pos, cnt, feats, dists):
    """feat.h5df exactly as update_memory_dist_base leaves it (memory_2.py:330-354): one group per voxel named
    grid_{row}_{col}_{h}; `features` (M, D) f32 and `distances` (M,) f32, both resizable along axis 0."""
    h5py = _h5py()
    if h5py is None:
        raise RuntimeError("writing feat.h5df needs h5py")
    off = np.zeros(len(cnt) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    feats = np.asarray(feats, np.float32)
    dists = np.asarray(dists, np.float32)
    with h5py.File(h5_path, "w") as h5f:
        for i, p in enumerate(np.asarray(pos)):
            g = h5f.create_group(f"grid_{int(p[0])}_{int(p[1])}_{int(p[2])}")
            g.create_dataset("features", data=feats[off[i]:off[i + 1]], maxshape=(None, feats.shape[1]), chunks=True)
            g.create_dataset("distances", data=dists[off[i]:off[i + 1]], maxshape=(None,), chunks=True)


def load_token_store(path):
    """-> pos (V,3), cnt (V), feats (T,D), dists (T) in HDF5 name order; flat arrays first, feat.h5df otherwise."""
    k = os.path.join(path, "feat_voxel_keys.npy")
    if os.path.exists(k):
        pos = np.load(k)
        off = np.load(os.path.join(path, "feat_token_offsets.npy"))
        return (pos, np.diff(off).astype(np.int32), np.load(os.path.join(path, "feat_features.npy")),
                np.load(os.path.join(path, "feat_distances.npy")))
    h5 = os.path.join(path, "feat.h5df")
    if os.path.exists(h5):
        if not have_h5py():
            raise RuntimeError(f"{h5} is an HDF5 token store but h5py is not installed; convert it with "
                               "bsc_nav_amd.store.convert_h5_store on a machine that has h5py")
        return read_h5_store(h5)
    raise FileNotFoundError(f"no token store under {path}")


def read_h5_store(h5_path):
    """feat.h5df -> (pos (V,3) i32, cnt (V,) i32, feats (T,D) f32, dists (T,) f32) in h5py iteration order, which is
    HDF5's native link order = bytewise name order (the tie order of voxel_localized, memory_2.py:623-665)."""
    h5py = _h5py()
    if h5py is None:
        raise RuntimeError("reading feat.h5df needs h5py")
    pos, cnt, feats, dists = [], [], [], []
    with h5py.File(h5_path, "r") as h5f:
        for name in h5f.keys():          # h5py iterates links in name order
            g = h5f[name]
            f = np.asarray(g["features"][:], np.float32)
            pos.append([int(x) for x in name.split("_")[1:4]])
            cnt.append(f.shape[0])
            feats.append(f)
            dists.append(np.asarray(g["distances"][:], np.float32))
    D = feats[0].shape[1] if feats else 0
    return (np.array(pos, np.int32).reshape(-1, 3), np.array(cnt, np.int32),
            np.concatenate(feats) if feats else np.zeros((0, D), np.float32),
            np.concatenate(dists) if dists else np.zeros(0, np.float32))


def convert_h5_store(path):
    pos, cnt, feats, dists = read_h5_store(os.path.join(path, "feat.h5df"))
    save_token_store(path, pos, cnt, feats, dists, write_h5=False)


def save_dense(path, acc, cnt):
    np.save(os.path.join(path, "dense_acc.npy"), acc)
    np.save(os.path.join(path, "dense_cnt.npy"), cnt)


def load_dense(path):
    return np.load(os.path.join(path, "dense_acc.npy")), np.load(os.path.join(path, "dense_cnt.npy"))
