"""HIP path (through the C-ABI of libbscnav.so) against the golden vectors and the CPU oracle.

Bit-exact: voxel indices, ids, counts, token-cache rows, rgb bytes, weights, top-down map, token store.
Floating point (dense feature sums, similarities): tolerance written at each assert.
"""
import random

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (no CPU fallback exists)")
    return torch


def _engine(cfg, mode="exact", **kw):
    import bsc_nav_amd as B
    return B.VoxelEngine(cfg["H"], cfg["W"], cfg["gs"], cfg["cs"], cfg["floor_height"], cfg["map_height"], cfg["g"],
                         cfg["D"], mode=mode, iter_size=cfg.get("iter_size", 50000), **kw)


def _oracle(cfg, mode=0, voxel_capacity=None):
    from oracle import oracle as orc
    c = orc.make_config(cfg["H"], cfg["W"], cfg["gs"], cfg["cs"], cfg["floor_height"], cfg["map_height"], cfg["g"],
                        cfg["D"], iter_size=cfg.get("iter_size", 50000), mode=mode)
    return orc, c, orc.OracleMemory(c, voxel_capacity)


@pytest.mark.parametrize("name", gu.GEOMETRY_FIXTURES)
def test_geometry_bit_exact(torch_cuda, name):
    import synth
    torch = torch_cuda
    z = gu.load(name)
    H, W = int(z["H"]), int(z["W"])
    _, depth, poses = synth.make_frames(int(z["seed"]), 2, H, W, str(z["kind"]), start_yaw_steps=1)
    assert synth.checksum(depth[1], poses) == str(z["input_sha"])
    import bsc_nav_amd as B
    eng = B.VoxelEngine(H, W, int(z["gs"]), float(z["cs"]), 0.0, 1.0, int(z["g"]), 8, mode="mean", voxel_capacity=16,
                        min_h=int(z["minh"]), max_h=int(z["maxh"]))
    d = torch.from_numpy(depth[1]).cuda()
    idx = torch.from_numpy(z["pick"]).cuda()
    o = eng.geometry(d, z["pc_tf"], idx)
    m = z["mask"].astype(bool)
    assert np.array_equal((o["flags"] & 1).astype(bool), m)
    assert np.array_equal(o["pc"][m], z["pc"].T[m])
    assert np.array_equal(o["pg"][m], z["pg"].T[m])
    assert np.array_equal(o["vox"][m], z["vox"][m])
    assert np.array_equal(o["pix"][m], z["pix"][m])
    assert np.array_equal(o["pat"][m], z["pat"][m])
    assert np.array_equal(o["r2"][m], z["r2"][m])
    ulp = np.abs(o["alpha"][m] - z["alpha"][m]) / np.spacing(z["alpha"][m])
    assert ulp.max() <= 1.0            # device exp vs NumPy exp: last-ulp only
    eng.close()


def _numpy_alpha(orc, c, depth, idx, T):
    g = orc.geometry(c, depth, idx, T)
    return np.array([np.exp(-r / (2 * 0.6)) for r in g["r2"]], dtype=np.float64)   # memory_2.py:873-875


def _run_engine(torch, z, batched=False, alpha_mode="numpy"):
    """Drive the HIP engine like obs2voxeltoken does, frame by frame (or all frames in one call)."""
    import bsc_nav_amd as B
    from oracle import oracle as orc
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    N = cfg["H"] * cfg["W"]
    P_max = sum(len(range(0, N, cfg["s"])) for _ in range(cfg["F"]))
    eng = _engine(cfg, max_points=max(P_max, N))
    oc = orc.make_config(cfg["H"], cfg["W"], cfg["gs"], cfg["cs"], cfg["floor_height"], cfg["map_height"], cfg["g"],
                         cfg["D"])
    chain = B.PoseChain()
    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    d_depth = torch.from_numpy(depth).cuda()
    d_rgb = torch.from_numpy(rgb).cuda()
    d_tok = torch.from_numpy(tokens).cuda()
    Ts, idxs, alphas, per_frame = [], [], [], []
    for f in range(cfg["F"]):
        T = chain.pc_transform(poses[f])
        idx = B.sample_indices(N, cfg["s"])
        alpha = _numpy_alpha(orc, oc, depth[f], idx, T) if alpha_mode == "numpy" else None
        if batched:
            Ts.append(T); idxs.append(idx); alphas.append(alpha)
            continue
        eng.ingest(d_depth[f:f + 1], d_rgb[f:f + 1], d_tok[f:f + 1], T[None], torch.from_numpy(idx).cuda(),
                   np.array([0, len(idx)]), None if alpha is None else torch.from_numpy(alpha).cuda())
        k = eng.counters()
        per_frame.append((k["iter_id"], k["max_id"]))
    if batched:
        off = np.concatenate([[0], np.cumsum([len(i) for i in idxs])])
        eng.ingest(d_depth, d_rgb, d_tok, np.stack(Ts), torch.from_numpy(np.concatenate(idxs)).cuda(), off,
                   None if alpha_mode != "numpy" else torch.from_numpy(np.concatenate(alphas)).cuda())
    return cfg, eng, np.array(per_frame, np.int64)


def _assert_state_matches_golden(eng, z):
    import synth
    k = eng.counters()
    assert k["max_id"] == int(z["max_id"]) and k["iter_id"] == int(z["iter_id"])
    f, p, d = eng.export_cache()
    assert np.array_equal(p, z["cache_pos"])
    assert np.array_equal(f[:, 0].astype(np.int32), z["cache_src"])
    assert np.array_equal(d, z["cache_dis"])
    assert synth.checksum(f) == str(z["cache_sha"])
    pos, rgb, w = eng.export_rgb()
    assert np.array_equal(pos, z["grid_rgb_pos"])
    assert np.array_equal(rgb, z["grid_rgb"])
    assert np.array_equal(w, z["weight"])
    occ = eng.export_occupied()
    assert int((occ >= 0).sum()) == int(z["occ_nnz"])
    assert np.array_equal(occ[pos[:, 0], pos[:, 1], pos[:, 2]], np.arange(len(pos)))
    mh, cv = eng.export_heightmap()
    rc = np.argwhere(np.isfinite(mh)).astype(np.int32)
    assert np.array_equal(rc, z["map_rc"])
    assert np.array_equal(mh[rc[:, 0], rc[:, 1]].astype(np.int32), z["map_h"])
    assert np.array_equal(cv[rc[:, 0], rc[:, 1]], z["map_rgb"])


@pytest.mark.parametrize("name", gu.INGEST_FIXTURES)
def test_ingest_exact_matches_reference(torch_cuda, name):
    z = gu.load(name)
    cfg, eng, per_frame = _run_engine(torch_cuda, z)
    assert np.array_equal(per_frame, z["per_frame_iter_max"])
    _assert_state_matches_golden(eng, z)
    eng.close()


@pytest.mark.parametrize("name", ["g2_mini_s1", "g2_c1_s50_iid", "g3_flush_small_cache"])
def test_ingest_one_batched_call_equals_frame_by_frame(torch_cuda, name):
    z = gu.load(name)
    cfg, eng, _ = _run_engine(torch_cuda, z, batched=True)
    _assert_state_matches_golden(eng, z)
    eng.close()


@pytest.mark.parametrize("name", gu.INGEST_FIXTURES)
def test_flush_store_and_localize_match_reference(torch_cuda, name):
    import synth
    torch = torch_cuda
    z = gu.load(name)
    cfg, eng, _ = _run_engine(torch, z)
    eng.flush()
    pos, cnt, feats, dists = eng.export_store()
    assert np.array_equal(pos, z["store_pos"])
    assert np.array_equal(cnt, z["store_cnt"])
    assert np.array_equal(feats[:, 0].astype(np.int32), z["store_src"])
    assert np.array_equal(dists, z["store_dis"])
    assert synth.checksum(feats) == str(z["store_sha"])
    for q in gu.query_specs(z):
        qtok = gu.query_tokens(q, cfg["seed"], cfg["D"], feats)
        pooled = eng.pool_query(torch.from_numpy(qtok).cuda())
        # f32 pooling, different summation order than torch: 2e-6 relative
        np.testing.assert_allclose(pooled.cpu().numpy(), q["pooled"].reshape(-1), rtol=2e-6, atol=2e-6)
        p, s, n = eng.localize(torch.from_numpy(q["pooled"].reshape(1, -1)).cuda(), K=q["K"], radius=q["radius"],
                               curr=q["curr"], floor=q["floor"])
        assert n[0] == len(q["pos"])
        gu.assert_topk_matches(p[0, :n[0]], s[0, :n[0]], q["pos"], q["sim"])   # scores within 2e-6 (bar: 1e-3)
    eng.close()


def test_store_roundtrip_import_export(torch_cuda):
    torch = torch_cuda
    z = gu.load("g2_mini_s7_yaw")
    cfg, eng, _ = _run_engine(torch, z)
    eng.flush()
    rgbs = eng.export_rgb()
    store = eng.export_store()
    eng2 = _engine(cfg)
    eng2.import_rgb(*rgbs)
    eng2.import_store(*store)
    for a, b in zip(eng2.export_store(), store):
        assert np.array_equal(a, b)
    for a, b in zip(eng2.export_rgb(), rgbs):
        assert np.array_equal(a, b)
    assert np.array_equal(eng2.export_occupied(), eng.export_occupied())
    q = next(gu.query_specs(z))
    qd = torch.from_numpy(q["pooled"].reshape(1, -1)).cuda()
    p1, s1, _ = eng.localize(qd, K=q["K"])
    p2, s2, _ = eng2.localize(qd, K=q["K"])
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    eng.close(); eng2.close()


def test_device_alpha_within_last_ulp_effects(torch_cuda):
    """Without host alpha the device exp() may differ from NumPy's in the last ulp."""
    z = gu.load("g2_mini_s1")
    _, eng, _ = _run_engine(torch_cuda, z, alpha_mode="device")
    pos, rgb, w = eng.export_rgb()
    assert np.array_equal(pos, z["grid_rgb_pos"])
    assert (rgb != z["grid_rgb"]).mean() < 1e-3
    np.testing.assert_allclose(w, z["weight"], rtol=3e-7, atol=0)
    eng.close()


def test_device_exp_within_one_ulp_of_numpy(torch_cuda):
    """bsc_exp (geometry_dev.h: table of 2^(j/64), degree-5 polynomial, constants in scalar registers) against NumPy's exp over
    the whole range of alpha = exp(-r^2 / 1.2): depths from min_depth to max_depth over a 640x480 frame (r^2 up to ~300), through
    the geometry export (generic chain) — never more than one ulp apart, most results identical."""
    import bsc_nav_amd as B
    torch = torch_cuda
    H, W = 480, 640
    eng = B.VoxelEngine(H, W, 256, 0.1, -12.8, 12.8, 14, 32, mode="mean", max_points=H * W)
    rs = np.random.RandomState(4)
    depth = np.exp(rs.uniform(np.log(0.1001), np.log(9.999), size=(H, W))).astype(np.float32)
    T = np.eye(4)
    o = eng.geometry(torch.from_numpy(depth).cuda(), T)
    ok = (o["flags"] & 1) != 0
    want = np.exp(-o["r2"][ok] / (2 * 0.6))
    got = o["alpha"][ok]
    assert ok.mean() > 0.99 and (got > 0).all()
    ulps = np.abs(got.view(np.int64) - want.view(np.int64))
    assert ulps.max() <= 1, ulps.max()
    assert (ulps == 0).mean() > 0.7, (ulps == 0).mean()
    # points that fail the depth check keep a defined alpha (0) in the export — infinite, huge and NaN depths included — and
    # bsc_exp itself returns 0 for -inf like np.exp (its range reduction alone made NaN of it)
    odd = np.full((H, W), 5.0, np.float32)
    odd[0, :4] = [np.inf, 3.0e38, np.nan, 1.0e30]
    o = eng.geometry(torch.from_numpy(odd).cuda(), T)
    assert not (o["flags"][:4] & 1).any() and (o["alpha"][:4] == 0.0).all() and np.isfinite(o["alpha"]).all()
    eng.close()


@pytest.mark.parametrize("mode", ["mean", "max"])
@pytest.mark.parametrize("name", ["g2_c1_s1000", "g2_mini_s7_yaw", "g2_c1_s50_iid"])      # one flush, at the end: no point loses its token
def test_dense_rows_equal_reduction_of_the_reference_store(torch_cuda, name, mode):
    """The dense modes do not exist in the reference, but what they reduce does: a voxel of the reference's token store
    that never filled up (fewer than cache_size rows, so no random replacement) holds exactly the tokens of the points the
    reference put into that voxel, one row per point (the goldens carry the source token of every stored row).  The dense
    mean / max of the HIP path over the same sampled points must be the mean / max of those rows, voxel by voxel — pinned to
    the reference's own point-to-voxel-to-token assignment, not to the oracle's extension."""
    import bsc_nav_amd as B
    torch = torch_cuda
    z = gu.load(name)
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    N = cfg["H"] * cfg["W"]
    eng = _engine(cfg, mode=mode, max_points=N, voxel_capacity=max(4096, int(z["max_id"]) + 16))
    chain = B.PoseChain()
    np.random.seed(cfg["seed"])                          # the reference's sampling stream (memory_2.py:747-749)
    d_depth, d_rgb, d_tok = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    for f in range(cfg["F"]):
        idx = B.sample_indices(N, cfg["s"])
        eng.ingest(d_depth[f:f + 1], d_rgb[f:f + 1], d_tok[f:f + 1], chain.pc_transform(poses[f])[None],
                   torch.from_numpy(idx).cuda(), np.array([0, len(idx)]))
    acc, cnt = eng.export_dense()
    pos = eng.export_rgb()[0]
    assert np.array_equal(pos, z["grid_rgb_pos"])
    vid_of = {tuple(p): i for i, p in enumerate(pos.tolist())}
    flat = tokens.reshape(-1, cfg["D"])
    cache_size, checked, start = int(cfg.get("cache_size", 10)), 0, 0
    for p, c in zip(z["store_pos"].tolist(), z["store_cnt"].tolist()):
        rows = flat[z["store_src"][start:start + c]]
        start += c
        if c >= cache_size or tuple(p) not in vid_of or not rows[:, 1:].any():      # full voxels and the grid_0_0_0 quirk group
            continue
        v = vid_of[tuple(p)]
        assert cnt[v] == c
        if mode == "max":
            assert np.array_equal(acc[v], rows.max(axis=0))
        else:
            np.testing.assert_allclose(acc[v] / c, rows.astype(np.float64).mean(axis=0), rtol=1e-3, atol=1e-3)
        checked += 1
    assert checked > 0.5 * len(pos)
    eng.close()


@pytest.mark.parametrize("passes", [1, 3])
@pytest.mark.parametrize("mode,omode", [("mean", 1), ("max", 2)])
@pytest.mark.parametrize("name", ["g2_mini_s1", "g2_c1_s50_iid"])
def test_dense_modes_match_oracle(torch_cuda, name, mode, omode, passes, monkeypatch):
    """North-star dense reduce: ids/positions/counts bit-exact, feature rows within 1e-3 (fp32 order).  passes 3: the call's
    reduce forced into passes over slices of its frames (what a token tile beyond the MALL size takes) — same results."""
    import bsc_nav_amd as B
    torch = torch_cuda
    if passes > 1:
        z0 = gu.load(name)
        c0 = gu.ingest_inputs(z0)[0]
        per_frame = c0["g"] * c0["g"] * c0["D"] * 4
        monkeypatch.setenv("BSC_REDUCE_PASS_BYTES", str(max(per_frame, (c0["F"] // 2) * per_frame // passes)))
    z = gu.load(name)
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    N = cfg["H"] * cfg["W"]
    vcap = min(N * cfg["F"], cfg["gs"] * cfg["gs"] * 64)     # every pixel ingested: more voxels than gs*gs
    orc, oc, om = _oracle(cfg, omode, vcap)
    eng = _engine(cfg, mode=mode, max_points=N * cfg["F"], voxel_capacity=vcap)
    chain = B.PoseChain()
    Ts = [chain.pc_transform(p) for p in poses]
    for f in range(cfg["F"]):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f])
    half = cfg["F"] // 2          # two multi-frame batches, every pixel (sample rate 1, row-major order)
    for a, b in ((0, half), (half, cfg["F"])):
        eng.ingest(torch.from_numpy(depth[a:b]).cuda(), torch.from_numpy(rgb[a:b]).cuda(),
                   torch.from_numpy(tokens[a:b]).cuda(), np.stack(Ts[a:b]))
    acc, cnt = eng.export_dense()
    oacc, ocnt = om.export_dense()
    opos, orgb, ow = om.export_rgb()
    pos, rgbv, w = eng.export_rgb()
    assert np.array_equal(pos, opos)                 # first-touch ids bit-exact
    assert np.array_equal(cnt, ocnt)                 # counts bit-exact
    assert (rgbv != orgb).mean() < 1e-3              # device exp vs libm exp, last ulp
    if mode == "max":
        assert np.array_equal(acc, oacc)             # max is order independent: bit-exact
    else:
        np.testing.assert_allclose(acc, oacc, rtol=1e-3, atol=1e-3)
        mean, omean = acc / cnt[:, None], oacc / ocnt[:, None]
        np.testing.assert_allclose(mean, omean, rtol=1e-3, atol=1e-3)
    # localize over the dense map
    qtok = gu.synth.make_query_tokens(5, 1, 256, cfg["D"])
    q = orc.pool_query(qtok)
    p, s, n = eng.localize(torch.from_numpy(q.reshape(1, -1)).cuda(), K=50)
    op, os_ = om.localize(q, K=50)
    gu.assert_topk_matches(p[0, :n[0]], s[0, :n[0]], op, os_, tol=5e-6)
    k = eng.counters()
    assert k["points_seen"] == N * cfg["F"] and k["max_id"] == len(opos)
    eng.close()


def test_batched_queries_equal_single_queries(torch_cuda):
    torch = torch_cuda
    z = gu.load("g2_c1_s50_iid")
    cfg, eng, _ = _run_engine(torch, z)
    eng.flush()
    rs = np.random.RandomState(3)
    Q = rs.standard_normal((11, cfg["D"])).astype(np.float32)
    singles = [eng.localize(torch.from_numpy(Q[i:i + 1]).cuda(), K=30) for i in range(len(Q))]
    # up to 4 queries share the one-wavefront-per-row scan of a single query: bit-identical sums
    for lo, hi in ((0, 4), (4, 7), (7, 9)):
        pb, sb, nb = eng.localize(torch.from_numpy(Q[lo:hi]).cuda(), K=30)
        for i in range(lo, hi):
            p1, s1, n1 = singles[i]
            assert np.array_equal(pb[i - lo], p1[0]) and np.array_equal(sb[i - lo], s1[0]) and nb[i - lo] == n1[0]
    # 5 queries and more go through the fp32 MFMA scan: same scores up to summation order, same top-K
    pb, sb, nb = eng.localize(torch.from_numpy(Q).cuda(), K=30)
    for i in range(len(Q)):
        p1, s1, n1 = singles[i]
        assert nb[i] == n1[0]
        gu.assert_topk_matches(pb[i, :nb[i]], sb[i, :nb[i]], p1[0, :n1[0]], s1[0, :n1[0]], tol=2e-6)
    eng.close()


def test_capacity_error_is_loud(torch_cuda):
    import bsc_nav_amd as B
    z = gu.load("g2_mini_s1")
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    eng = _engine(cfg, mode="mean", voxel_capacity=100)
    T = B.PoseChain().pc_transform(poses[0])
    torch = torch_cuda
    with pytest.raises(B._lib.BscError, match="capacity"):
        eng.ingest(torch.from_numpy(depth[:1]).cuda(), torch.from_numpy(rgb[:1]).cuda(),
                   torch.from_numpy(tokens[:1]).cuda(), T[None])
        eng.counters()
    eng.close()


@pytest.mark.parametrize("mode", ["mean", "max"])
def test_two_shard_merge_on_one_gpu_equals_single_map(torch_cuda, mode):
    """The multi-GPU merge (dist.merge_dense_maps) uses bsc_dense_gather / bsc_dense_replace; exercise the same
    calls with two engines on one GPU standing in for two ranks: union of keys, rows in union order, reduce,
    replace — and compare with one engine that ingested every frame."""
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bd
    torch = torch_cuda
    z = gu.load("g2_mini_s1")
    cfg, rgb, depth, poses, tokens = gu.ingest_inputs(z)
    N = cfg["H"] * cfg["W"]
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    d, c, t = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    full = _engine(cfg, mode=mode, max_points=N * cfg["F"], voxel_capacity=200000)
    full.ingest(d, c, t, Ts)
    half = cfg["F"] // 2
    shards = []
    for a, b in ((0, half), (half, cfg["F"])):
        e = _engine(cfg, mode=mode, max_points=N * cfg["F"], voxel_capacity=200000)
        e.ingest(d[a:b].contiguous(), c[a:b].contiguous(), t[a:b].contiguous(), Ts[a:b])
        shards.append(e)
    codes = [bd.pack_keys(e.keys_tensor()) for e in shards]
    union = torch.unique(torch.cat(codes))
    ukeys = bd.unpack_keys(union)
    rows = [e.dense_gather(ukeys) for e in shards]
    if mode == "max":
        acc = torch.maximum(rows[0][0], rows[1][0])
    else:
        acc = rows[0][0] + rows[1][0]
    cnt = rows[0][1] + rows[1][1]
    shards[0].dense_replace(ukeys, acc, cnt)                     # "rank 0" now holds the merged map
    macc, mcnt = shards[0].export_dense()
    mpos = shards[0].export_rgb()[0]
    facc, fcnt = full.export_dense()
    fpos = full.export_rgb()[0]
    order_m = np.lexsort((mpos[:, 2], mpos[:, 1], mpos[:, 0]))
    order_f = np.lexsort((fpos[:, 2], fpos[:, 1], fpos[:, 0]))
    assert np.array_equal(mpos[order_m], fpos[order_f])          # same voxel set
    assert np.array_equal(mcnt[order_m], fcnt[order_f])          # counts exact
    if mode == "max":
        assert np.array_equal(macc[order_m], facc[order_f])
    else:
        np.testing.assert_allclose(macc[order_m], facc[order_f], rtol=1e-3, atol=1e-3)
    # the merged map answers queries like the single map (scores 5e-6, ties in name order)
    q = torch.from_numpy(gu.synth.make_query_tokens(9, 1, 4, cfg["D"])[0, :1]).cuda()
    p1, s1, n1 = shards[0].localize(q, K=40)
    p2, s2, n2 = full.localize(q, K=40)
    gu.assert_topk_matches(p1[0, :n1[0]], s1[0, :n1[0]], p2[0, :n2[0]], s2[0, :n2[0]], tol=5e-6)
    # sentinel (padding) keys of the union give neutral rows
    pad = torch.tensor([[-1, -1, -1]], dtype=torch.int32, device="cuda")
    a0, c0 = shards[1].dense_gather(pad)
    assert int(c0[0]) == 0 and (torch.isinf(a0).all() if mode == "max" else (a0 == 0).all())
    for e in shards + [full]:
        e.close()


def test_mfma_batched_queries_match_single_query_path(torch_cuda):
    """Q >= 16 takes the fp32-MFMA GEMM kernel (k_cosine_mfma); Q < 16 the wavefront-per-row kernel.  Same
    similarities within fp32 summation-order noise, same top-K (ties in name order)."""
    torch = torch_cuda
    z = gu.load("g2_c1_s50_iid")            # D = 32: one K chunk of the MFMA kernel, ~11k voxels, tie-heavy tokens
    cfg, eng, _ = _run_engine(torch, z)
    eng.flush()
    rs = np.random.RandomState(4)
    for Q in (16, 40, 70, 130):
        Qm = rs.standard_normal((Q, cfg["D"])).astype(np.float32)
        pb, sb, nb = eng.localize(torch.from_numpy(Qm).cuda(), K=64)
        for i in (0, 7, Q - 1):
            p1, s1, n1 = eng.localize(torch.from_numpy(Qm[i:i + 1]).cuda(), K=64)
            assert nb[i] == n1[0]
            gu.assert_topk_matches(pb[i, :nb[i]], sb[i, :nb[i]], p1[0, :n1[0]], s1[0, :n1[0]], tol=2e-6)
    eng.close()


def test_large_map_localize_filter_path_matches_numpy(torch_cuda):
    """> 256 k candidates take the sample-threshold filter; compare with a NumPy top-K on the same map, incl. an
    adversarial layout (best rows last) that forces the fallback."""
    import bsc_nav_amd as B
    torch = torch_cuda
    V, D, gs = 300_000, 32, 128
    eng = B.VoxelEngine(48, 64, gs, 0.1, -6.4, 6.4, 16, D, mode="mean", voxel_capacity=V + 8, max_points=4096)
    rs = np.random.RandomState(0)
    codes = rs.permutation(gs ** 3)[:V]
    keys = np.stack([codes // (gs * gs), (codes // gs) % gs, codes % gs], 1).astype(np.int32)
    rows = rs.standard_normal((V, D)).astype(np.float32)
    q = rs.standard_normal((20, D)).astype(np.float32)      # >= 16 queries: MFMA cosine + filtered selection
    for adversarial in (False, True):
        if adversarial:      # similarity to q[0] ascending with the index: the sample threshold is useless
            order = np.argsort(rows @ q[0] / np.linalg.norm(rows, axis=1))
            rows = rows[order]
        eng.dense_replace(torch.from_numpy(keys).cuda(), torch.from_numpy(rows).cuda(),
                          torch.ones(V, dtype=torch.int32, device="cuda"))
        pos, sim, n = eng.localize(torch.from_numpy(q).cuda(), K=100)
        rn = rows / np.maximum(np.linalg.norm(rows, axis=1, keepdims=True), 1e-8)
        qn = q / np.linalg.norm(q, axis=1, keepdims=True)
        ref = qn @ rn.T
        for i in range(len(q)):
            top = np.argsort(-ref[i], kind="stable")[:100]
            assert n[i] == 100
            np.testing.assert_allclose(sim[i], ref[i][top], rtol=0, atol=3e-6)
            assert set(map(tuple, pos[i].tolist())) == set(map(tuple, keys[top].tolist()))
    eng.close()


@pytest.mark.parametrize("case", ["c1", "c2", "c3", "c4", "c5", "c6", "c7"])
def test_cluster_centers_match_reference(torch_cuda, case):
    """bsc_cluster_centers == GESObjectNavRobot.weighted_cluster_centers (BSCAgent.py:479-497) on the reference's own
    outputs: DBSCAN labels and sizes exact, centres 1e-12 relative (f64 sums in the same index order)."""
    import bsc_nav_amd as B
    z = gu.load("g5_cluster_centers")
    eng = B.VoxelEngine(48, 64, 64, 0.1, -3.2, 3.2, 16, 16, mode="mean", voxel_capacity=64, max_points=4096)
    centers, labels, sizes = eng.cluster_centers(z[f"{case}_pos"], z[f"{case}_sim"])
    assert np.array_equal(labels, z[f"{case}_labels"])
    assert sizes == [int(v) for v in z[f"{case}_sizes"]]
    assert centers.shape == z[f"{case}_centers"].shape and centers.dtype == np.float64
    np.testing.assert_allclose(centers, z[f"{case}_centers"], rtol=1e-12, atol=0)
    eng.close()


def test_cluster_centers_on_resident_topk(torch_cuda):
    """Clustering straight from the last localize call's device-resident top-K == clustering its host copy."""
    from oracle import oracle as orc
    torch = torch_cuda
    z = gu.load("g2_mini_s1")
    cfg, eng, _ = _run_engine(torch, z)
    eng.flush()
    q = next(x for x in gu.query_specs(z) if x["K"] == 100)
    pos, sim, n = eng.localize(torch.from_numpy(q["pooled"].reshape(1, -1)).cuda(), K=100)
    c_dev, l_dev, s_dev = eng.cluster_centers(K=int(n[0]))
    c_host, l_host, s_host = eng.cluster_centers(pos[0, :n[0]], sim[0, :n[0]])
    c_orc, l_orc, s_orc = orc.cluster_centers(pos[0, :n[0]], sim[0, :n[0]].astype(np.float64))
    assert np.array_equal(l_dev, l_host) and np.array_equal(l_dev, l_orc) and s_dev == s_host == list(s_orc)
    assert np.array_equal(c_dev, c_host)
    np.testing.assert_allclose(c_dev, c_orc, rtol=1e-12, atol=0)
    eng.close()
