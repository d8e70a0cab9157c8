"""SURVEY.md §8 rows a-14 / f-3 on the host: the single-floor split of load_memory (memory_2.py:202-252) against
outputs of the reference's own method (g8_floor_split.npz), and the feat.h5df adapter of bsc_nav_amd.store against the
HDF5 layout the reference's update_memory_dist_base leaves (memory_2.py:330-354), through the in-memory h5py
stand-in (tests/golden/fake_h5py.py; h5py itself is not part of this image)."""
import importlib
import os
import sys

import numpy as np
import pytest

import fake_h5py
import golden_util as gu


@pytest.fixture(scope="module")
def z():
    return gu.load("g8_floor_split")


def _pkg(name):
    # host-only modules of the package (no GPU, no libbscnav needed)
    import bsc_nav_amd  # noqa: F401
    return importlib.import_module("bsc_nav_amd." + name)


def test_floor_split_matches_reference(z):
    floors = _pkg("floors")
    gs, cs = int(z["grid"][0]), float(z["grid"][1])
    assert len(z["cases"]) >= 7
    for name in z["cases"]:
        name = str(name)
        sel = floors.select_floor(z[f"{name}_base_height"], z[f"{name}_pos"], cs, float(z[f"{name}_current_height"]))
        assert sel["num_floors"] == int(z[f"{name}_num_floors"]), name
        np.testing.assert_array_equal(np.array(sel["levels"]), z[f"{name}_floor_heights"])      # same means, bit for bit
        assert sel["current_floor"] == int(z[f"{name}_current_floor"]), name
        assert sel["zrange"] == z[f"{name}_range"].tolist(), name
        assert np.array_equal(z[f"{name}_pos"][sel["mask"]], z[f"{name}_floor_pos"]), name
        assert np.array_equal(z[f"{name}_rgb"][sel["mask"]], z[f"{name}_floor_rgb"]), name


def test_floor_split_without_any_floor_raises_like_the_reference():
    floors = _pkg("floors")
    # 10 spread-out heights, min_samples 2: every sample is noise -> the reference's argmin over an empty list raises
    with pytest.raises(ValueError):
        floors.select_floor(np.arange(10) * 5.0, np.zeros((4, 3), np.int32), 0.1, 0.0)


@pytest.fixture()
def h5(monkeypatch):
    saved = sys.modules.get("h5py")
    fake_h5py.install()
    fake_h5py.File._stores.clear()
    yield fake_h5py
    if saved is None:
        sys.modules.pop("h5py", None)
    else:
        sys.modules["h5py"] = saved


def _reference_written_file(z, path, rs):
    """Recreate the file the reference wrote: same groups and datasets, created in a scrambled order (creation order
    must not matter: HDF5 iterates links by name)."""
    off = np.concatenate([[0], np.cumsum(z["h5_cnt"])])
    with fake_h5py.File(path, "a") as f:
        for i in rs.permutation(len(z["h5_cnt"])):
            g = f.create_group(str(z["h5_group_names"][i]))
            g.create_dataset("features", data=z["h5_feats"][off[i]:off[i + 1]], maxshape=(None, z["h5_feats"].shape[1]), chunks=True)
            g.create_dataset("distances", data=z["h5_dists"][off[i]:off[i + 1]], maxshape=(None,), chunks=True)


def test_read_h5_store_returns_reference_iteration_order(z, h5):
    store = _pkg("store")
    assert store.have_h5py()
    _reference_written_file(z, "mem://ref", np.random.RandomState(0))
    pos, cnt, feats, dists = store.read_h5_store("mem://ref")
    # group names as the reference formed them <-> positions; name order has grid_1x_ before grid_1_ ('0'..'9' < '_')
    names = [f"grid_{p[0]}_{p[1]}_{p[2]}" for p in pos]
    assert names == [str(n) for n in z["h5_group_names"]] == sorted(names)
    assert any(a.split("_")[1] != b.split("_")[1] and int(a.split("_")[1]) > int(b.split("_")[1]) for a, b in zip(names, names[1:]))
    assert np.array_equal(pos, z["h5_pos"]) and np.array_equal(cnt, z["h5_cnt"])
    assert feats.dtype == np.float32 and np.array_equal(feats, z["h5_feats"]) and np.array_equal(dists, z["h5_dists"])
    assert pos[0].tolist() == [0, 0, 0] and not feats[:cnt[0]].any()          # the grid_0_0_0 zero-row quirk group


def test_write_h5_store_produces_the_reference_layout(z, h5):
    store = _pkg("store")
    rs = np.random.RandomState(1)
    # hand the voxels over in a scrambled order: the file must still iterate in name order
    order = rs.permutation(len(z["h5_cnt"]))
    off = np.concatenate([[0], np.cumsum(z["h5_cnt"])])
    feats = np.concatenate([z["h5_feats"][off[i]:off[i + 1]] for i in order])
    dists = np.concatenate([z["h5_dists"][off[i]:off[i + 1]] for i in order])
    store.write_h5_store("mem://ours", z["h5_pos"][order], z["h5_cnt"][order], feats, dists)
    man = fake_h5py.manifest("mem://ours")
    assert [g for g, _ in man] == [str(n) for n in z["h5_group_names"]]
    for (g, ds), fshape, dlen in zip(man, z["h5_feature_shapes"], z["h5_distance_shapes"]):
        assert [d for d, _, _, _ in ds] == [str(n) for n in z["h5_dataset_names"]] == ["distances", "features"]
        by = {d: (s, dt, r) for d, s, dt, r in ds}
        assert by["features"] == (tuple(fshape), "float32", True) and by["distances"] == ((int(dlen),), "float32", True)
    assert bool(z["h5_all_resizable"]) and [str(d) for d in z["h5_dataset_dtypes"]] == ["float32"]
    got = store.read_h5_store("mem://ours")
    for a, b in zip(got, (z["h5_pos"], z["h5_cnt"], z["h5_feats"], z["h5_dists"])):
        assert np.array_equal(a, b)
    # a reference-style append to one of our groups works (datasets are resizable like the reference's, :345-349)
    with fake_h5py.File("mem://ours", "a") as f:
        g = f[str(z["h5_group_names"][3])]
        n = g["features"].shape[0]
        g["features"].resize((n + 1, g["features"].shape[1]))
        g["distances"].resize((n + 1,))


def test_memory_dir_token_store_flat_and_h5_paths(z, h5, tmp_path, monkeypatch):
    store = _pkg("store")
    d = str(tmp_path)
    # the stand-in is keyed by path: route <dir>/feat.h5df through it
    store.save_token_store(d, z["h5_pos"], z["h5_cnt"], z["h5_feats"], z["h5_dists"])
    assert os.path.join(d, "feat.h5df") in fake_h5py.File._stores            # written because "h5py" is importable
    flat = store.load_token_store(d)
    for f in ("feat_voxel_keys.npy", "feat_token_offsets.npy", "feat_features.npy", "feat_distances.npy"):
        os.remove(os.path.join(d, f))
    open(os.path.join(d, "feat.h5df"), "w").close()                          # a reference-built dir has only this file
    via_h5 = store.load_token_store(d)
    for a, b, c in zip(flat, via_h5, (z["h5_pos"], z["h5_cnt"], z["h5_feats"], z["h5_dists"])):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    store.convert_h5_store(d)                                                # h5 -> flat arrays, for boxes without h5py
    assert os.path.exists(os.path.join(d, "feat_features.npy"))
    for a, b in zip(store.load_token_store(d), flat):
        assert np.array_equal(a, b)


def test_h5_store_without_h5py_fails_loudly(tmp_path):
    store = _pkg("store")
    if store.have_h5py():
        pytest.skip("real h5py present")
    open(os.path.join(str(tmp_path), "feat.h5df"), "w").close()
    with pytest.raises(RuntimeError, match="h5py"):
        store.load_token_store(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        store.load_token_store(str(tmp_path / "nowhere"))
