"""The CPU oracle (oracle/) against golden vectors produced by the reference's own code.

This is what pins the oracle: every fixture under tests/golden/ was written by
tests/golden/gen_golden.py, which imports /root/reference and runs its functions on the seeded
inputs of tests/golden/synth.py.  No GPU needed.
"""
import random

import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as orc


def _cfg(c, mode=0):
    return orc.make_config(c["H"], c["W"], c["gs"], c["cs"], c["floor_height"], c["map_height"], c["g"], c["D"],
                           iter_size=c.get("iter_size", 50000), mode=mode)


@pytest.mark.parametrize("name", gu.GEOMETRY_FIXTURES)
def test_geometry_bit_exact(name):
    import synth
    z = gu.load(name)
    H, W = int(z["H"]), int(z["W"])
    _, depth, poses = synth.make_frames(int(z["seed"]), 2, H, W, str(z["kind"]), start_yaw_steps=1)
    assert synth.checksum(depth[1], poses) == str(z["input_sha"])
    cfg = orc.make_config(H, W, int(z["gs"]), float(z["cs"]), 0.0, 0.0, int(z["g"]), 8)
    cfg.min_h, cfg.max_h = int(z["minh"]), int(z["maxh"])
    assert np.array_equal(np.array(cfg.K[:]).reshape(3, 3), z["K"])
    assert np.array_equal(np.array(cfg.Kinv[:]).reshape(3, 3), z["Kinv"])
    assert np.array_equal(np.array(cfg.Kpatch[:]).reshape(3, 3), z["Kp"])
    o = orc.geometry(cfg, depth[1], z["pick"], z["pc_tf"])
    m = z["mask"].astype(bool)
    assert np.array_equal(o["valid"].astype(bool), m)
    assert m.sum() > 1000
    assert np.array_equal(o["pc"][m], z["pc"].T[m])          # camera points: bit-exact doubles
    assert np.array_equal(o["pg"][m], z["pg"].T[m])          # map-frame points
    assert np.array_equal(o["vox"][m], z["vox"][m])          # row, col, h
    assert np.array_equal(o["pix"][m], z["pix"][m])          # knife-edge recovered pixel
    assert np.array_equal(o["pat"][m], z["pat"][m])          # patch coordinates
    assert np.array_equal(o["r2"][m], z["r2"][m])
    # alpha: libm exp vs NumPy's exp may differ in the last ulp (documented)
    ulp = np.abs(o["alpha"][m] - z["alpha"][m]) / np.spacing(z["alpha"][m])
    assert ulp.max() <= 1.0
    # the fixture really exercises the knife edge: some recovered pixels are one column left of the source
    assert (z["pix"][m][:, 0] != (z["pick"][m] % W)).sum() > 100


def _run_oracle(z, alpha_mode="numpy"):
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    c = _cfg(cfg)
    mem = orc.OracleMemory(c)
    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    N = cfg["H"] * cfg["W"]
    per_frame = []
    for f in range(cfg["F"]):
        T = mem.chain.pc_transform(poses[f])
        idx = orc.sample_indices(N, cfg["s"])
        alpha = None
        if alpha_mode == "numpy":   # exactly the reference's expression, memory_2.py:873-875
            g = orc.geometry(c, depth[f], idx, T)
            alpha = np.array([np.exp(-r / (2 * 0.6)) for r in g["r2"]], dtype=np.float64)
        mem.ingest_frame(depth[f], rgb[f], idx, T, tokens[f], alpha)
        k = mem.counters()
        per_frame.append((k["iter_id"], k["max_id"]))
    return cfg, mem, np.array(per_frame, np.int64)


@pytest.mark.parametrize("name", gu.INGEST_FIXTURES)
def test_ingest_state_bit_exact(name):
    import synth
    z = gu.load(name)
    cfg, mem, per_frame = _run_oracle(z)
    assert np.array_equal(per_frame, z["per_frame_iter_max"])
    k = mem.counters()
    assert k["max_id"] == int(z["max_id"]) and k["iter_id"] == int(z["iter_id"])
    # token cache before the final flush
    f, p, d = mem.export_cache()
    assert np.array_equal(p, z["cache_pos"])
    assert np.array_equal(f[:, 0].astype(np.int32), z["cache_src"])
    assert np.array_equal(d, z["cache_dis"])
    assert synth.checksum(f) == str(z["cache_sha"])
    # rgb voxels: ids (first-touch order), positions, truncating weighted mean, weights
    pos, rgb, w = mem.export_rgb()
    assert np.array_equal(pos, z["grid_rgb_pos"])
    assert np.array_equal(rgb, z["grid_rgb"])
    assert np.array_equal(w, z["weight"])
    occ = mem.export_occupied()
    assert int((occ >= 0).sum()) == int(z["occ_nnz"])
    assert np.array_equal(occ[pos[:, 0], pos[:, 1], pos[:, 2]], np.arange(len(pos)))
    # top-down map
    mh, cv = mem.export_heightmap()
    rc = np.argwhere(np.isfinite(mh)).astype(np.int32)
    assert np.array_equal(rc, z["map_rc"])
    assert np.array_equal(mh[rc[:, 0], rc[:, 1]].astype(np.int32), z["map_h"])
    assert np.array_equal(cv[rc[:, 0], rc[:, 1]], z["map_rgb"])


@pytest.mark.parametrize("name", gu.INGEST_FIXTURES)
def test_flush_and_query(name):
    import synth
    z = gu.load(name)
    cfg, mem, _ = _run_oracle(z)
    mem.flush()
    pos, cnt, feats, dists = mem.export_store()
    assert np.array_equal(pos, z["store_pos"])          # HDF5 name order incl. the grid_0_0_0 zero-row quirk
    assert np.array_equal(cnt, z["store_cnt"])
    assert np.array_equal(feats[:, 0].astype(np.int32), z["store_src"])
    assert np.array_equal(dists, z["store_dis"])
    assert synth.checksum(feats) == str(z["store_sha"])
    assert cnt.max() <= 10
    for q in gu.query_specs(z):
        qtok = gu.query_tokens(q, cfg["seed"], cfg["D"], feats)
        pooled = orc.pool_query(qtok)
        np.testing.assert_allclose(pooled, q["pooled"].reshape(-1), rtol=2e-6, atol=2e-6)
        p, s = mem.localize(q["pooled"], K=q["K"], radius=q["radius"], curr=q["curr"], floor=q["floor"])
        gu.assert_topk_matches(p, s, q["pos"], q["sim"])
        assert np.array_equal(p[0], q["top1"].reshape(-1))


def test_flush_small_cache_exercises_replacement():
    z = gu.load("g3_flush_small_cache")
    assert (z["store_cnt"] == 10).sum() > 50           # saturated voxels => random replacement happened
    assert int(z["per_frame_iter_max"][-1][0]) < 1500    # in-loop flushes wrapped the cache


def test_libm_alpha_close_to_numpy():
    """With libm's exp instead of NumPy's the state still matches except where the last ulp matters."""
    z = gu.load("g2_mini_s1")
    _, mem, _ = _run_oracle(z, alpha_mode="libm")
    pos, rgb, w = mem.export_rgb()
    assert np.array_equal(pos, z["grid_rgb_pos"])
    assert (rgb != z["grid_rgb"]).mean() < 1e-3
    np.testing.assert_allclose(w, z["weight"], rtol=3e-7, atol=0)


def test_name_key_is_bytewise_string_order():
    rs = np.random.RandomState(0)
    trip = np.concatenate([rs.randint(0, 1200, size=(3000, 3)), rs.randint(0, 25, size=(500, 3)),
                           [[0, 0, 0], [1, 0, 0], [10, 0, 0], [100, 0, 0], [1, 9, 0], [1, 10, 0], [19, 1, 1], [1, 1, 1],
                            [1, 1, 11], [1, 1, 2], [99999, 99999, 99999]]])
    trip = np.unique(trip, axis=0)
    by_name = sorted(range(len(trip)), key=lambda i: f"grid_{trip[i][0]}_{trip[i][1]}_{trip[i][2]}")
    by_key = sorted(range(len(trip)), key=lambda i: orc.name_key(*trip[i]))
    assert by_name == by_key


def test_localize_output_format_matches_reference_exemplars():
    """localize_results/*.npy in the reference are (100,3) int64 [row,col,h]."""
    z = gu.load("g2_mini_s1")
    q = next(x for x in gu.query_specs(z) if x["K"] == 100)
    assert q["pos"].shape == (100, 3) and q["pos"].dtype == np.int64 and q["sim"].dtype == np.float64


@pytest.mark.parametrize("case", ["c1", "c2", "c3", "c4", "c5", "c6", "c7"])
def test_cluster_centers_match_reference(case):
    """weighted_cluster_centers (BSCAgent.py:479-497): labels exact, sizes exact, centres to 1e-12 relative."""
    z = gu.load("g5_cluster_centers")
    centers, labels, sizes = orc.cluster_centers(z[f"{case}_pos"], z[f"{case}_sim"])
    assert np.array_equal(labels, z[f"{case}_labels"])
    assert np.array_equal(sizes, z[f"{case}_sizes"])
    assert centers.shape == z[f"{case}_centers"].shape
    np.testing.assert_allclose(centers, z[f"{case}_centers"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("case", ["f1", "f2", "f3", "f4", "f5", "f6", "f7"])
def test_frontier_helpers_match_reference(case):
    """FrontierExplorer helpers (memory_2.py:1165-1311): frontier cells, clusters, centres, gains, selected target."""
    z = gu.load("g7_frontier")
    cv, nav = z[f"{case}_cv_map"], z[f"{case}_nav"]
    gs, min_size, radius = (int(v) for v in z[f"{case}_params"])
    mask = orc.frontier_mask(cv, nav)
    assert np.array_equal((mask & 1).astype(bool) & nav, z[f"{case}_navigable_mask"])       # build_navigable_mask
    assert np.array_equal(np.argwhere(mask & 2), z[f"{case}_frontiers"])                    # find_frontiers, row-major
    assert np.array_equal(orc.frontier_mask(cv, z[f"{case}_navigable_mask"]), mask)         # either mask form
    r = orc.frontier_clusters(cv, mask & 2, min_size, radius)
    assert r["n"] == len(z[f"{case}_sizes"])
    assert np.array_equal(r["labels"], z[f"{case}_labels"])
    assert np.array_equal(r["sizes"], z[f"{case}_sizes"]) and np.array_equal(r["first"], z[f"{case}_first"])
    assert np.array_equal(r["centers"], z[f"{case}_centers"]) and np.array_equal(r["gains"], z[f"{case}_gains"])
    want = z[f"{case}_best"]
    if np.isnan(want[0]):
        assert r["best"] == -1
    else:
        assert np.array_equal(r["centers"][r["best"]], want)


@pytest.mark.parametrize("name", ["g2_mini_s7_yaw", "g2_c1_s1000"])
def test_numpy_loop_restatement_matches_the_reference_goldens(name):
    """oracle/numpy_loop.py (the reference-style per-point Python loop that bench.py times as `cpu_baseline_numpy_loop`)
    against goldens produced by the reference's own obs2voxeltoken: cache rows, ids, rgb bytes, weights, top-down map."""
    from oracle import oracle as orc
    from oracle.numpy_loop import NumpyLoopMemory
    z = gu.load(name)
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    N = cfg["H"] * cfg["W"]
    mem = NumpyLoopMemory(cfg["H"], cfg["W"], cfg["gs"], cfg["cs"], cfg["floor_height"], cfg["map_height"], cfg["g"], cfg["D"])
    chain = orc.PoseChain()
    np.random.seed(cfg["seed"])
    for f in range(cfg["F"]):
        T = chain.pc_transform(poses[f])
        idx = orc.sample_indices(N, cfg["s"])
        mem.ingest_frame(depth[f], rgb[f], idx, T, tokens[f])
    assert mem.max_id == int(z["max_id"]) and mem.iter_id == int(z["iter_id"])
    assert np.array_equal(mem.grid_feat_pos[:mem.iter_id], z["cache_pos"])
    assert np.array_equal(mem.grid_feat[:mem.iter_id, 0].astype(np.int32), z["cache_src"])
    assert np.array_equal(mem.grid_feat_dis[:mem.iter_id], z["cache_dis"])
    assert np.array_equal(mem.grid_rgb_pos[:mem.max_id], z["grid_rgb_pos"])
    assert np.array_equal(mem.grid_rgb[:mem.max_id], z["grid_rgb"])
    assert np.array_equal(mem.weight[:mem.max_id], z["weight"])
    rc = np.argwhere(np.isfinite(mem.max_height)).astype(np.int32)
    assert np.array_equal(rc, z["map_rc"]) and np.array_equal(mem.cv_map[rc[:, 0], rc[:, 1]], z["map_rgb"])
