"""Edge cases and BASELINE-size properties of the HIP path (empty / ragged inputs, maximum sizes,
size-independent invariants where the sequential oracle would take minutes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eng(mode="mean", H=48, W=64, gs=128, D=16, g=16, **kw):
    import bsc_nav_amd as B
    return B.VoxelEngine(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=mode, **kw)


def _frames(F, H, W, seed=0, kind="room"):
    import synth
    return synth.make_frames(seed, F, H, W, kind)


@pytest.mark.parametrize("mode", ["mean", "exact"])
def test_frames_without_a_single_valid_point(mode):
    import torch
    import bsc_nav_amd as B
    H, W, D, g = 48, 64, 16, 16
    eng = _eng(mode, max_points=4 * H * W)
    rgb = torch.zeros((2, H, W, 3), dtype=torch.uint8, device="cuda")
    tok = torch.randn((2, g, g, D), device="cuda")
    T = np.stack([np.eye(4)] * 2)
    for depth_val in (0.0, 50.0, float("nan")):            # below min_depth, above max_depth, NaN
        depth = torch.full((2, H, W), depth_val, device="cuda")
        eng.ingest(depth, rgb, tok, T)
    far = torch.full((2, H, W), 9.9, device="cuda")          # valid depth but outside a 12.8 m grid after a big shift
    Tfar = T.copy(); Tfar[:, 0, 3] = 500.0
    eng.ingest(far, rgb, tok, Tfar)
    k = eng.counters()
    assert k["max_id"] == 0 and k["points_passed"] == 0 and k["points_seen"] == 4 * 2 * H * W
    if mode == "exact":
        eng.flush()                                          # zero rows only: the grid_0_0_0 quirk group
        pos, cnt, feats, dists = eng.export_store()
        assert pos.tolist() == [[0, 0, 0]] and cnt.tolist() == [10] and not feats.any()
    p, s, n = eng.localize(torch.randn(1, D, device="cuda"), K=5)
    assert n[0] == (1 if mode == "exact" else 0)             # the reference would also find only grid_0_0_0
    eng.close()


def test_ragged_batch_with_empty_and_single_point_frames():
    import torch
    import bsc_nav_amd as B
    from oracle import oracle as orc
    H, W, D, g, F = 48, 64, 16, 16, 4
    rgb, depth, poses = _frames(F, H, W, seed=3)
    import synth
    tokens = synth.make_tokens(3, F, g, D)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    rs = np.random.RandomState(0)
    idxs = [rs.permutation(H * W)[:300].astype(np.int32), np.zeros(0, np.int32), rs.permutation(H * W)[:1].astype(np.int32),
            rs.permutation(H * W)[:777].astype(np.int32)]
    off = np.concatenate([[0], np.cumsum([len(i) for i in idxs])]).astype(np.int64)
    eng = _eng("mean", max_points=4096)
    eng.ingest(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(tokens).cuda(), Ts,
               torch.from_numpy(np.concatenate(idxs)).cuda(), off)
    oc = orc.make_config(H, W, 128, 0.1, -6.4, 6.4, g, D, mode=1)
    om = orc.OracleMemory(oc)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], idxs[f], Ts[f], tokens[f])
    acc, cnt = eng.export_dense()
    oacc, ocnt = om.export_dense()
    assert np.array_equal(eng.export_rgb()[0], om.export_rgb()[0]) and np.array_equal(cnt, ocnt)
    np.testing.assert_allclose(acc, oacc, rtol=1e-3, atol=1e-3)
    eng.close()


def test_localize_k_larger_than_map_and_large_k():
    import torch
    eng = _eng("mean", voxel_capacity=5000, max_points=4096)
    rs = np.random.RandomState(1)
    keys = np.unique(rs.randint(0, 128, size=(700, 3)), axis=0).astype(np.int32)
    rows = rs.standard_normal((len(keys), 16)).astype(np.float32)
    eng.dense_replace(torch.from_numpy(keys).cuda(), torch.from_numpy(rows).cuda(),
                      torch.ones(len(keys), dtype=torch.int32, device="cuda"))
    q = rs.standard_normal((2, 16)).astype(np.float32)
    ref = (q / np.linalg.norm(q, axis=1, keepdims=True)) @ (rows / np.linalg.norm(rows, axis=1, keepdims=True)).T
    for K in (5, 600, 1000, 4096):                           # 600 > 512: device-wide sort path; 1000, 4096 > voxels
        pos, sim, n = eng.localize(torch.from_numpy(q).cuda(), K=K)
        for i in range(2):
            m = min(K, len(keys))
            assert n[i] == m
            order = np.argsort(-ref[i], kind="stable")[:m]
            np.testing.assert_allclose(sim[i, :m], ref[i][order], atol=3e-6, rtol=0)
            assert np.array_equal(pos[i, :m], keys[order]) or set(map(tuple, pos[i, :m].tolist())) == set(map(tuple, keys[order].tolist()))
    eng.close()


def test_full_size_invariants_640x480_d768():
    """BASELINE configs[1] size (640x480, 768-D, 256^3, every pixel): properties that do not need the sequential oracle."""
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    H, W, D, g, gs, F = 480, 640, 768, 14, 256, 8
    poses = synthetic.random_walk_poses(11, F)
    rgb, depth, _ = synthetic.make_frames(11, F, H, W, "room", poses=poses)
    tok = torch.randn((F, g, g, D), device="cuda")
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])

    def build(splits, repeat=1):
        e = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="mean", voxel_capacity=400_000, max_points=F * H * W)
        for _ in range(repeat):
            for a, b in splits:
                e.ingest(depth[a:b].contiguous(), rgb[a:b].contiguous(), tok[a:b].contiguous(), Ts[a:b])
        return e

    one = build([(0, F)])
    four = build([(0, 2), (2, 4), (4, 6), (6, 8)])
    twice = build([(0, F)], repeat=2)
    p1, r1, w1 = one.export_rgb()
    p4, r4, w4 = four.export_rgb()
    a1, c1 = one.export_dense()
    a4, c4 = four.export_dense()
    a2, c2 = twice.export_dense()
    k = one.counters()
    # (1) batching does not change the order-defined state: ids, rgb bytes, weights are identical for 1 call vs 4 calls
    assert np.array_equal(p1, p4) and np.array_equal(r1, r4) and np.array_equal(w1, w4) and np.array_equal(c1, c4)
    np.testing.assert_allclose(a1, a4, rtol=1e-3, atol=1e-3)                 # fp32 sums, different association
    # (2) conservation: every passing point is counted exactly once
    assert int(c1.astype(np.int64).sum()) == k["points_passed"] and k["points_seen"] == F * H * W
    assert np.array_equal(one.export_occupied()[p1[:, 0], p1[:, 1], p1[:, 2]], np.arange(len(p1)))
    # (3) linearity: the same frames twice double counts and sums (sums exactly: x + x)
    assert np.array_equal(c2, 2 * c1)
    np.testing.assert_allclose(a2, 2 * a1, rtol=1e-5, atol=1e-4)
    # (4) the top-down map holds the highest voxel of every column
    mh, _ = one.export_heightmap()
    col_max = np.full((gs, gs), -np.inf)
    np.maximum.at(col_max, (p1[:, 0], p1[:, 1]), p1[:, 2].astype(np.float64))
    assert np.array_equal(mh, col_max)
    # (5) a voxel's own mean row localizes to that voxel with similarity 1
    v = int(np.argmax(c1))
    q = torch.from_numpy((a1[v] / c1[v]).astype(np.float32)).cuda().reshape(1, -1)
    pos, sim, n = one.localize(q, K=3)
    assert abs(sim[0, 0] - 1.0) < 1e-5 and sim[0, 0] >= sim[0, 1]
    ties = np.isclose(sim[0, :n[0]], sim[0, 0], atol=1e-6)
    assert any(np.array_equal(pos[0, i], p1[v]) for i in np.nonzero(ties)[0])
    for e in (one, four, twice):
        e.close()


def test_exact_mode_at_scale_against_oracle():
    """~1e6 points, 40 frames, a token cache that wraps ~50 times with random replacement: the whole exact-mode state
    (ids, rgb chain with host alpha, top-down map, token store incl. replacement draws, top-K) equals the sequential oracle."""
    import random
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    H, W, g, D, gs, F, s, iter_size = 240, 320, 14, 24, 128, 40, 3, 20000
    rgb, depth, poses = synth.make_frames(21, F, H, W, "room")
    tokens = synth.make_tokens(21, F, g, D)
    eng = B.VoxelEngine(H, W, gs, 0.1, -2.0, 4.4, g, D, mode="exact", iter_size=iter_size, voxel_capacity=300_000,
                        token_capacity=1_500_000, max_points=8 * H * W)
    oc = orc.make_config(H, W, gs, 0.1, -2.0, 4.4, g, D, iter_size=iter_size)
    om = orc.OracleMemory(oc, voxel_capacity=300_000)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    np.random.seed(5)
    idxs = [B.sample_indices(H * W, s) for _ in range(F)]
    alphas = []
    for f in range(F):                                   # one alpha array for both sides (libm exp of the oracle's r^2)
        gm = orc.geometry(oc, depth[f], idxs[f], Ts[f])
        alphas.append(np.exp(-gm["r2"] / (2 * 0.6)))
    random.seed(77)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], idxs[f], Ts[f], tokens[f], alphas[f])
    om.flush()
    random.seed(77)                                      # same Python RNG stream for the replacement draws
    d_depth, d_rgb, d_tok = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    for a in range(0, F, 8):                             # 8 frames per call: several in-call flushes each
        off = np.concatenate([[0], np.cumsum([len(i) for i in idxs[a:a + 8]])]).astype(np.int64)
        eng.ingest(d_depth[a:a + 8], d_rgb[a:a + 8], d_tok[a:a + 8], Ts[a:a + 8],
                   torch.from_numpy(np.concatenate(idxs[a:a + 8])).cuda(), off,
                   torch.from_numpy(np.concatenate(alphas[a:a + 8])).cuda())
    eng.flush()
    k, ok = eng.counters(), om.counters()
    assert k["flushes"] == ok["flushes"] and k["flushes"] > 40
    assert (k["max_id"], k["store_voxels"], k["store_tokens"]) == (ok["max_id"], ok["store_voxels"], ok["store_tokens"])
    for a, b in zip(eng.export_rgb(), om.export_rgb()):
        assert np.array_equal(a, b)
    for a, b in zip(eng.export_heightmap(), om.export_heightmap()):
        assert np.array_equal(a, b)
    for a, b in zip(eng.export_store(), om.export_store()):
        assert np.array_equal(a, b)
    assert (eng.export_store()[1] == 10).sum() > 1000     # thousands of saturated voxels: replacement really happened
    q = orc.pool_query(synth.make_query_tokens(2, 1, 196, D))
    p, sim, n = eng.localize(torch.from_numpy(q[None]).cuda(), K=100)
    op, osim = om.localize(q, K=100)
    import golden_util as gu
    gu.assert_topk_matches(p[0, :n[0]], sim[0, :n[0]], op, osim, tol=5e-6)
    eng.close()


@pytest.mark.parametrize("mode", ["mean", "max", "exact"])
def test_bf16_tokens_equal_widened_f32_tokens(mode):
    """bsc_ingest_typed(BSC_TOK_BF16) widens rows exactly: every state array equals the f32 call on the same values."""
    import torch
    H, W, D, g, F = 48, 64, 32, 16, 6
    rgb, depth, poses = _frames(F, H, W, seed=5)
    import bsc_nav_amd as B
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    tok16 = torch.randn((F, g, g, D), device="cuda").to(torch.bfloat16)
    tok32 = tok16.float()
    out = []
    import random
    for tok in (tok32, tok16):
        random.seed(3)                                       # replacement draws of the flush (memory_2.py:352)
        eng = _eng(mode, D=D, g=g, max_points=F * H * W, iter_size=1000)
        r, d = torch.as_tensor(rgb).cuda(), torch.as_tensor(depth).cuda()
        eng.ingest(d[:3].contiguous(), r[:3].contiguous(), tok[:3].contiguous(), Ts[:3])
        eng.ingest(d[3:].contiguous(), r[3:].contiguous(), tok[3:].contiguous(), Ts[3:])
        if mode == "exact":
            eng.flush()
            out.append(eng.export_store())
        else:
            out.append(eng.export_dense())
        eng.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("D", [256, 768, 1024, 520])
def test_bf16_rows_with_large_multiplicities_equal_f32_rows(D):
    """Dense mean with bf16 token rows where a (voxel, frame, patch) pair holds many hundreds of points (34x46-pixel patches,
    0.5 m cells): the reduce feeds the multiplicity to v_dot2c_f32_bf16 byte by byte (exact in bf16), and the sums must be
    those of the f32 rows (within an ulp-level tolerance: two accumulation steps instead of one above 255), counts equal.
    D = 256 / 768 / 1024 / 520: one, three, four and a ragged number of 4-column accumulators per lane — 8 consecutive columns
    of a bf16 row come in with one 16-byte load and feed two accumulators, an odd last accumulator keeps its 8-byte load."""
    import torch
    import bsc_nav_amd as B
    H, W, g, F = 240, 320, 7, 4
    rgb, depth, poses = _frames(F, H, W, seed=8)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    tok16 = torch.randn((F, g, g, D), device="cuda").to(torch.bfloat16)
    out = []
    for tok in (tok16.float(), tok16):
        eng = B.VoxelEngine(H, W, 64, 0.5, -8.0, 8.0, g, D, mode="mean", max_points=F * H * W)
        eng.ingest(torch.as_tensor(depth).cuda(), torch.as_tensor(rgb).cuda(), tok.contiguous(), Ts)
        out.append(eng.export_dense())
        assert eng.counters()["pairs_last_call"] * 100 < F * H * W          # > 100 points per pair on average, whole tiles in one cell
        eng.close()
    (a32, c32), (a16, c16) = out
    assert np.array_equal(c32, c16) and c32.max() > 2000      # voxels of > 2000 points out of 4 frames x a few patches: pairs far above 255
    np.testing.assert_allclose(a16, a32, rtol=2e-6, atol=1e-4)


def _dense_vs_oracle(H, W, g, D, gs, cs, lo, hi, F, per_call, seed, vcap, kind="room", depth_override=None,
                     alpha_override=None):
    """Dense mean mode, every pixel, host alpha: ids, positions, rgb bytes, weights, top-down map, counts bit-exact and
    feature sums within 1e-3 against the sequential oracle."""
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    rgb, depth, poses = synth.make_frames(seed, F, H, W, kind)
    if depth_override is not None:
        depth = depth_override(depth)
    tokens = synth.make_tokens(seed, F, g, D)
    eng = B.VoxelEngine(H, W, gs, cs, lo, hi, g, D, mode="mean", voxel_capacity=vcap, max_points=per_call * H * W)
    oc = orc.make_config(H, W, gs, cs, lo, hi, g, D, mode=1)
    om = orc.OracleMemory(oc, voxel_capacity=vcap)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    alphas = []
    for f in range(F):
        gm = orc.geometry(oc, depth[f], None, Ts[f])
        al = np.exp(-gm["r2"] / (2 * 0.6))
        if alpha_override is not None:
            al = alpha_override(f, al)
        alphas.append(al)
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f], al)
    d_depth, d_rgb, d_tok = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    for a in range(0, F, per_call):
        eng.ingest(d_depth[a:a + per_call].contiguous(), d_rgb[a:a + per_call].contiguous(),
                   d_tok[a:a + per_call].contiguous(), Ts[a:a + per_call], None, None,
                   torch.from_numpy(np.concatenate(alphas[a:a + per_call])).cuda())
    k, ok = eng.counters(), om.counters()
    assert k["max_id"] == ok["max_id"]
    for a, b in zip(eng.export_rgb(), om.export_rgb()):
        assert np.array_equal(a, b)
    for a, b in zip(eng.export_heightmap(), om.export_heightmap()):
        assert np.array_equal(a, b)
    (acc, cnt), (oacc, ocnt) = eng.export_dense(), om.export_dense()
    assert np.array_equal(cnt, ocnt) and int(ocnt.astype(np.int64).sum()) == k["points_passed"]
    # feature values = per-voxel means (the oracle adds the points one by one in f32, the device adds multiplicity x row)
    c = np.maximum(cnt, 1)[:, None].astype(np.float64)
    np.testing.assert_allclose(acc / c, oacc / c, rtol=1e-3, atol=1e-3)
    eng.close()
    return int(ocnt.max()), k["max_id"]


def test_full_size_dense_against_oracle():
    """BASELINE configs[1] frame and token size (640x480, 14x14x768, 256^3 grid), 6 frames in two calls, vs the oracle."""
    longest, n_vox = _dense_vs_oracle(480, 640, 14, 768, 256, 0.1, -12.8, 12.8, F=6, per_call=3, seed=31, vcap=400_000)
    assert n_vox > 5000


@pytest.mark.parametrize("D,bf16,mode,kind", [(768, True, "mean", "iid"), (768, True, "max", "room"), (1024, False, "mean", "room"),
                                               (128, False, "max", "iid"), (1024, True, "mean", "room"), (384, True, "mean", "iid")])
def test_column_sliced_reduce_against_oracle(monkeypatch, D, bf16, mode, kind):
    """k_dense_reduce_sliced (one column slice of the token rows per XCD; taken for very long pair lists, forced here): 4 slices
    of 24 lanes (768-D bf16), 8 of 32 / 16 (1024-D f32 / bf16), 4 of 8 lanes (128-D f32), 2 of 24 (384-D bf16) — counts exact,
    max rows bit-exact, means within 1e-3 of the sequential oracle, over two calls (old voxels take the read-modify-write)."""
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    monkeypatch.setenv("BSC_SLICED_MIN_PAIRS", "1")
    H, W, g, gs, F = 240, 320, 14, 128, 4
    rgb, depth, poses = synth.make_frames(51, F, H, W, kind)
    tokens = synth.make_tokens(51, F, g, D)
    d_tok = torch.from_numpy(tokens).cuda()
    if bf16:
        d_tok = d_tok.bfloat16()
        tokens = d_tok.float().cpu().numpy()
    eng = B.VoxelEngine(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=mode, voxel_capacity=400_000, max_points=2 * H * W)
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=1 if mode == "mean" else 2), voxel_capacity=400_000)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f])
    for a in range(0, F, 2):
        eng.ingest(torch.from_numpy(depth[a:a + 2]).cuda(), torch.from_numpy(rgb[a:a + 2]).cuda(), d_tok[a:a + 2].contiguous(), Ts[a:a + 2])
    (acc, cnt), (oacc, ocnt) = eng.export_dense(), om.export_dense()
    assert np.array_equal(eng.export_rgb()[0], om.export_rgb()[0]) and np.array_equal(cnt, ocnt) and len(cnt) > 2000
    if mode == "max":
        assert np.array_equal(acc, oacc)
    else:
        c = np.maximum(cnt, 1)[:, None].astype(np.float64)
        np.testing.assert_allclose(acc / c, oacc / c, rtol=1e-3, atol=1e-3)
    eng.close()


@pytest.mark.parametrize("D,bf16,mode,kind", [(768, False, "mean", "room"), (768, True, "mean", "iid"), (1024, False, "max", "room"),
                                               (128, True, "max", "iid")])
def test_dense_reduce_in_passes_over_frame_slices(monkeypatch, D, bf16, mode, kind):
    """A call whose token tile is larger than the MALL is reduced in passes over slices of its frames (forced here: 4-frame calls
    in passes of 1 or 2 frames — the last pass takes the remainder): counts exact, max rows bit-exact, means within 1e-3 of the
    sequential oracle and equal to a one-pass reduce up to f32 summation order; voxels that first appear in a later slice and
    voxels whose pairs all sit in one slice included."""
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    H, W, g, gs, F = 240, 320, 14, 128, 8
    rgb, depth, poses = synth.make_frames(77, F, H, W, kind)
    tokens = synth.make_tokens(77, F, g, D)
    d_tok = torch.from_numpy(tokens).cuda()
    if bf16:
        d_tok = d_tok.bfloat16()
        tokens = d_tok.float().cpu().numpy()
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=1 if mode == "mean" else 2), voxel_capacity=400_000)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f])
    oacc, ocnt = om.export_dense()
    res = {}
    per_frame = g * g * D * (2 if bf16 else 4)
    for label, pass_bytes in (("one", 0), ("per_frame", per_frame), ("three_then_one", 3 * per_frame)):
        monkeypatch.setenv("BSC_REDUCE_PASS_BYTES", str(pass_bytes))
        eng = B.VoxelEngine(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=mode, voxel_capacity=400_000, max_points=4 * H * W)
        for a in range(0, F, 4):
            eng.ingest(torch.from_numpy(depth[a:a + 4]).cuda(), torch.from_numpy(rgb[a:a + 4]).cuda(), d_tok[a:a + 4].contiguous(), Ts[a:a + 4])
        acc, cnt = eng.export_dense()
        assert np.array_equal(eng.export_rgb()[0], om.export_rgb()[0]) and np.array_equal(cnt, ocnt) and len(cnt) > 2000
        if mode == "max":
            assert np.array_equal(acc, oacc)
        else:
            c = np.maximum(cnt, 1)[:, None].astype(np.float64)
            np.testing.assert_allclose(acc / c, oacc / c, rtol=1e-3, atol=1e-3)
        res[label] = acc
        eng.close()
    if mode == "mean":
        c = np.maximum(ocnt, 1)[:, None].astype(np.float64)
        assert np.abs((res["one"] - res["per_frame"]) / c).max() < 2e-5
        assert np.abs((res["one"] - res["three_then_one"]) / c).max() < 2e-5


def test_groups_longer_than_the_run_length_field_are_cut_into_runs():
    """A voxel capacity of 2^26 leaves 6 bits for a run's length beside the voxel id in the sort key: k_points has to cut every
    block-local group of more than 64 points into several runs (1 m cells: groups of up to 2048 points), and the chain must still
    see every voxel's points in order j — rgb bytes, weights, ids bit-exact against the sequential oracle."""
    longest, n_vox = _dense_vs_oracle(240, 320, 16, 4, 32, 1.0, -16.0, 16.0, F=4, per_call=2, seed=21, vcap=(1 << 26) - 2)
    assert longest > 20_000 and n_vox < 2000


def test_very_long_voxel_chains_against_oracle():
    """1 m cells: single voxels collect > 10^5 points of a call, so the rgb chain walks thousands of 64-point chunks per
    segment (prefetch pipeline, longest-first queue, segments far longer than the wavefront count) — bit-exact rgb."""
    longest, n_vox = _dense_vs_oracle(480, 640, 14, 32, 32, 1.0, -16.0, 16.0, F=6, per_call=6, seed=9, vcap=40_000)
    assert longest > 100_000 and n_vox < 2000


def test_long_chains_of_old_voxels_over_several_calls():
    """The same scene in three calls: from the second call on the long segments belong to voxels that already carry a
    large weight (no early part on the quad chain; the speculating kernel starts from the stored state, crosses weight
    binades inside a round and keeps colours that almost never change)."""
    longest, n_vox = _dense_vs_oracle(480, 640, 14, 32, 32, 1.0, -16.0, 16.0, F=6, per_call=2, seed=10, vcap=40_000)
    assert longest > 100_000


@pytest.mark.parametrize("knob", ["BSC_ORDER_MAIN=1", "BSC_QUAD_CHAIN_ONLY=1", "BSC_CHAIN_EAGER=1", "BSC_NO_HOT_SPLIT=1",
                                  "BSC_LONG_NWV=8", "BSC_HOT_LOG2=10", "BSC_LONG_LOG2=8", "BSC_GROUP_RPW=4",
                                  "BSC_CHAIN_SPLIT=0", "BSC_LONG_NWV=16", "BSC_TOTALS_UNFUSED=1", "BSC_NO_MAILBOX=1", "BSC_PROJ_DIVIDE=1",
                                  "BSC_REC12=1", "BSC_SORT_ROCPRIM=1"])
def test_chain_and_order_knobs_keep_the_result(knob):
    """The library's A/B switches move work between streams and kernels (order stage on the main stream, quad chain only, chain
    right behind its order stage, no hot split, 8-wavefront hot tiles, other length classes, 1024-point blocks; round 6: the long
    chain as one launch, 16-wavefront hot tiles, block scans + totals as separate launches, scalars by copy instead of the host
    mailbox, source pixel by division, 12-byte records, rocPRIM sorts) — never the result: the long-chain scene in three calls, bit-exact against the oracle, under each of them.  Most are read once per
    process, so every case runs in its own interpreter."""
    import os, subprocess, sys
    name, val = knob.split("=")
    code = ("import sys; sys.path[:0] = ['.', 'tests', 'tests/golden']; import test_gpu_edges as t; "
            "longest, n = t._dense_vs_oracle(240, 320, 16, 8, 32, 1.0, -16.0, 16.0, F=6, per_call=2, seed=10, vcap=40_000); "
            "assert longest > 20_000, longest; print('ok', longest, n)")
    env = dict(os.environ, **{name: val})
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_weight_ties_and_saturation_against_oracle():
    """A handful of 50 m cells take every point.  alpha = 1 until the f32 weights pass 2^23 (one ulp = 1), then values
    from {0.25, 0.5, 0.75, 1}: 0.5 is an exact tie whose rounding depends on the parity of the running weight (the
    speculating kernel's prefix prediction cannot know it, its check must catch it), and past 2^24 every addition is lost
    or a tie (memory_2.py:896-899 in f32).  rgb bytes and weights bit-exact against the sequential oracle."""
    rs = np.random.RandomState(4)
    H, W = 240, 320

    def alphas(f, al):
        if f < 125:
            return np.ones_like(al)
        return rs.choice(np.array([0.25, 0.5, 0.75, 1.0]), size=al.shape)

    longest, n_vox = _dense_vs_oracle(H, W, 16, 16, 4, 50.0, -100.0, 100.0, F=260, per_call=65, seed=12, vcap=64,
                                      alpha_override=alphas)
    assert longest > (1 << 24) and n_vox <= 8         # a voxel collected more points than an f32 weight can count


@pytest.mark.parametrize("hw,g", [((480, 640), 14), ((240, 320), 16), ((680, 680), 16), ((97, 131), 7)])
def test_fast_geometry_equals_generic_chain(hw, g):
    """k_points' fast path (pinhole collapse of the fma chains, x / cs through the exact reciprocal form, shared
    reciprocal of p2, per-pixel patch tables) against the generic per-point chains (BSC_GENERIC_GEOMETRY=1 forces them;
    they are what the geometry goldens pin): identical ids, rgb bytes, weights, top-down map, counts and feature rows,
    every pixel and sub-sampled."""
    import os
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import synthetic
    H, W = hw
    D, gs, F = 16, 256, 3
    poses = synthetic.random_walk_poses(5, F)
    rgb, depth, _ = synthetic.make_frames(5, F, H, W, "room", poses=poses)
    depth[1] = synthetic.make_frames(6, 1, H, W, "iid")[1][0]          # one frame of worst-case depth
    # one frame of depths on which a quotient q / z rounds most delicately: powers of two, short mantissas and their float
    # neighbours (the division-free source-pixel test of the fast path decides on the sign of an exact residual there)
    base = np.array([0.125, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.5], np.float32)
    special = np.concatenate([base, np.nextafter(base, np.float32(0)), np.nextafter(base, np.float32(100)),
                              np.float32(1.0) / np.arange(1, 9, dtype=np.float32) * np.float32(3.0)])
    depth[2] = torch.from_numpy(np.tile(special, H * W // len(special) + 1)[:H * W].reshape(H, W).copy()).to(depth.device)
    tok = torch.randn((F, g, g, D), device="cuda")
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    rs = np.random.RandomState(0)
    idxs = [np.sort(rs.permutation(H * W)[:H * W // 3]).astype(np.int32) for _ in range(F)]
    off = np.concatenate([[0], np.cumsum([len(i) for i in idxs])]).astype(np.int64)
    idx = torch.from_numpy(np.concatenate(idxs)).cuda()

    def build(generic):
        if generic:
            os.environ["BSC_GENERIC_GEOMETRY"] = "1"
        try:
            e = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="max", voxel_capacity=500_000, max_points=F * H * W)
        finally:
            os.environ.pop("BSC_GENERIC_GEOMETRY", None)
        e.ingest(depth, rgb, tok, Ts)
        e.ingest(depth, rgb, tok, Ts, idx, off)
        out = e.export_rgb() + e.export_heightmap() + e.export_dense()
        e.close()
        return out

    a, b = build(False), build(True)
    assert len(a[0]) > 1000
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_cold_start_every_voxel_new_and_capacity_boundary():
    """A first call in which every voxel is new (claims, new-cell list, id ranking) against the oracle at a size where
    thousands of points race for each cell, then a call that hits the voxel capacity exactly at its boundary."""
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    H, W, g, D, gs, F = 240, 320, 14, 8, 64, 4
    rgb, depth, poses = synth.make_frames(12, F, H, W, "iid")
    tokens = synth.make_tokens(12, F, g, D)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.2, -6.4, 6.4, g, D, mode=1), voxel_capacity=300_000)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f])
    n = om.counters()["max_id"]
    d, c, t = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    eng = B.VoxelEngine(H, W, gs, 0.2, -6.4, 6.4, g, D, mode="mean", voxel_capacity=n, max_points=F * H * W)   # exactly enough
    eng.ingest(d, c, t, Ts)
    assert eng.counters()["max_id"] == n and np.array_equal(eng.export_rgb()[0], om.export_rgb()[0])
    assert np.array_equal(eng.export_dense()[1], om.export_dense()[1])
    eng.close()
    small = B.VoxelEngine(H, W, gs, 0.2, -6.4, 6.4, g, D, mode="mean", voxel_capacity=n - 1, max_points=F * H * W)
    with pytest.raises(B._lib.BscError, match="capacity"):
        small.ingest(d, c, t, Ts)
    # the refused call leaves the handle in its error state (bsc_counters keeps reporting the capacity flag) until bsc_reset; the
    # ids of the clipped new voxels were still assigned (no cell is left with a provisional claim), so a reset handle is as new
    with pytest.raises(B._lib.BscError, match="capacity"):
        small.counters()
    small.reset()
    assert small.counters()["max_id"] == 0
    om1 = orc.OracleMemory(orc.make_config(H, W, gs, 0.2, -6.4, 6.4, g, D, mode=1), voxel_capacity=300_000)
    om1.ingest_frame(depth[0], rgb[0], None, Ts[0], tokens[0])
    assert om1.counters()["max_id"] < n - 1
    small.ingest(d[:1], c[:1], t[:1], Ts[:1])
    assert np.array_equal(small.export_rgb()[0], om1.export_rgb()[0]) and np.array_equal(small.export_dense()[1], om1.export_dense()[1])
    small.close()


def test_token_store_grows_like_the_unbounded_reference_store():
    """The reference's HDF5 store is unbounded (memory_2.py:330-354).  A flush — also one triggered from the middle of an
    ingest call, memory_2.py:880-881 — that would overflow the token pool grows it in place: the store equals the one of
    a context created with ample capacity, frame by frame; the same holds after a store was imported into a context
    that was sized for exactly the imported rows (load_memory followed by more exploration)."""
    import random
    import torch
    import bsc_nav_amd as B
    import synth
    H, W, g, D, F = 48, 64, 16, 16, 4
    rgb, depth, poses = synth.make_frames(4, F, H, W, "room")
    tokens = synth.make_tokens(4, F, g, D)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    d, c, t = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))

    def build(cap, frames, seed_state=None, start=None):
        eng = B.VoxelEngine(H, W, 128, 0.1, -6.4, 6.4, g, D, mode="exact", iter_size=2000, voxel_capacity=20_000,
                            token_capacity=cap, max_points=H * W)
        if start is not None:
            eng.import_rgb(*start[0])
            eng.import_store(*start[1])
        random.seed(0) if seed_state is None else random.setstate(seed_state)
        for f in frames:
            eng.ingest(d[f:f + 1], c[f:f + 1], t[f:f + 1], Ts[f:f + 1])
        return eng

    big = build(1_000_000, range(F))
    small = build(2500, range(F))                    # every in-call flush after the first has to grow the pool
    assert small.counters()["flushes"] == big.counters()["flushes"] >= 4
    for a, b in zip(big.export_store(), small.export_store()):
        assert np.array_equal(a, b)
    assert all(np.array_equal(a, b) for a, b in zip(big.export_rgb(), small.export_rgb()))
    # a loaded store in a context with no spare rows, then more frames
    half = build(1_000_000, range(2))
    half.flush()
    st = random.getstate()
    state = (half.export_rgb(), half.export_store())
    n_tok = len(state[1][2])
    cont_big = build(1_000_000, range(2, F), seed_state=st, start=state)
    cont_small = build(n_tok, range(2, F), seed_state=st, start=state)
    for a, b in zip(cont_big.export_store(), cont_small.export_store()):
        assert np.array_equal(a, b)
    p, s_, n = cont_small.localize(torch.randn(1, D, device="cuda"), K=5)
    assert n[0] == 5
    for e in (big, small, half, cont_big, cont_small):
        e.close()


def test_selection_filter_overflow_falls_back_to_the_exact_rounds():
    """The batched top-K takes its per-query threshold from a sample of 16 blocks spread over the map and keeps the candidates that
    beat it.  When the sample is unrepresentative — here every sampled block holds rows orthogonal to the queries while all other
    rows score high, so ~7/8 of 2^17 candidates beat the threshold and the survivor list (32768) overflows — the call must notice
    and fall back to the direct rounds: same answer as an fp64 scan."""
    import torch
    import bsc_nav_amd as B
    import golden_util as gu
    V, D, gs, K, Q = 1 << 17, 64, 64, 100, 8
    gen = torch.Generator(device="cuda").manual_seed(3)
    codes = torch.randperm(gs ** 3, device="cuda", generator=gen)[:V]
    keys = torch.stack([codes // (gs * gs), (codes // gs) % gs, codes % gs], dim=1).to(torch.int32).contiguous()
    q = torch.randn((Q, D), device="cuda", generator=gen)
    q[:, D // 2:] = 0                                                    # queries live in the first half of the dimensions
    rows = torch.randn((V, D), device="cuda", generator=gen)
    blk = torch.arange(V, device="cuda") // 1024                         # the sample takes blocks 0, 8, 16, ... of the 128
    sampled = (blk % 8) == 0
    rows[sampled, :D // 2] = 0                                           # sampled rows: orthogonal to every query (similarity 0)
    rows[~sampled, :D // 2] = rows[~sampled, :D // 2].abs() * torch.sign(q[0, :D // 2])    # the others: aligned with query 0
    eng = B.VoxelEngine(48, 64, gs, 0.1, -3.2, 3.2, 16, D, mode="mean", voxel_capacity=V + 8, max_points=4096)
    eng.dense_replace(keys, rows.contiguous(), torch.ones(V, dtype=torch.int32, device="cuda"))
    pos, sim, n = eng.localize(q, K=K)
    rn = rows.double() / rows.double().norm(dim=1, keepdim=True).clamp_min(1e-8)
    ref = (q.double() / q.double().norm(dim=1, keepdim=True)) @ rn.T
    top = torch.topk(ref, K, dim=1)
    kk = keys.cpu().numpy()
    for i in range(Q):
        assert n[i] == K
        gu.assert_topk_near(pos[i], sim[i], kk[top.indices[i].cpu().numpy()], top.values[i].cpu().numpy(), tol=2e-6)
    assert sim[0, K - 1] > 0.5                                           # query 0's answers come from the unsampled rows
    eng.close()


def test_point_log_capacity_and_empty_replay():
    """The point log refuses a call that would overflow it before anything is changed; bsc_replay_colour leaves voxels without
    records at zero."""
    import torch
    import bsc_nav_amd as B
    import synth
    H, W, g, D, F = 48, 64, 16, 16, 2
    rgb, depth, poses = synth.make_frames(9, F, H, W, "room")
    tokens = synth.make_tokens(9, F, g, D)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    eng = B.VoxelEngine(H, W, 128, 0.1, -6.4, 6.4, g, D, mode="mean", voxel_capacity=50_000, max_points=H * W)
    eng.point_log_enable(H * W + 10)
    d, c, t = (torch.from_numpy(a).cuda() for a in (depth, rgb, tokens))
    eng.ingest(d[:1], c[:1], t[:1], Ts[:1])
    before = (eng.export_rgb(), eng.export_dense())
    with pytest.raises(B._lib.BscError, match="point log full"):
        eng.ingest(d[1:2], c[1:2], t[1:2], Ts[1:2])
    after = (eng.export_rgb(), eng.export_dense())
    for a, b in zip(before, after):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    cells, recs = eng.point_log()
    assert cells.numel() == H * W and recs.shape == (H * W, 3)
    # colour state imported over a logging engine: the log no longer describes the map and says so instead of replaying wrong colours
    pos0, rgb0, w0 = eng.export_rgb()
    eng.import_rgb(pos0, rgb0, w0)
    with pytest.raises(B._lib.BscError, match="does not describe"):
        eng.point_log()
    eng.point_log_enable(H * W + 10)                                     # a new log starts clean
    assert eng.point_log()[0].numel() == 0
    vox = torch.tensor([1, 1, 3], dtype=torch.int32, device="cuda")      # voxels 0, 2, 4 have no record
    r = torch.tensor([[0, 0x3ff00000, 0x0a0b0c], [0, 0x3fe00000, 0x010203], [0, 0x3fd00000, 0x040506]], dtype=torch.int32, device="cuda")
    out_rgb, out_w = eng.replay_colour(vox, r, 5)
    assert out_rgb.cpu().tolist()[0] == [0, 0, 0] and out_rgb.cpu().tolist()[2] == [0, 0, 0] and out_rgb.cpu().tolist()[4] == [0, 0, 0]
    assert out_rgb.cpu().tolist()[3] == [6, 5, 4] and float(out_w[3]) == 0.25 and float(out_w[0]) == 0.0
    # voxel 1: first point (alpha 1: colour 0x0a0b0c, weight 1), then alpha 0.5 with colour 0x010203
    c0 = [0x0c, 0x0b, 0x0a]
    exp = [int((np.float64(np.float32(c0[k]) * np.float32(1.0)) + [3, 2, 1][k] * 0.5) / 1.5) for k in range(3)]
    assert out_rgb.cpu().tolist()[1] == exp and float(out_w[1]) == 1.5
    eng.close()
