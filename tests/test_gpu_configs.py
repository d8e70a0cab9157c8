"""BASELINE.json configs at their real sizes on the HIP path (C1: one 320x240 frame, ViT-B/16 tokens projected to 512-D,
128^3 grid — the reference's CPU-runnable case; C3: ViT-L/14 tokens 16x16x1024 into a 512^3 grid;
C4: localize top-K over a 512^3 x 1024-D map; C5: 2^20 voxels x 1024-D, 256 batched queries, voxel-sharded).

The sequential oracle is affordable for a few full-size frames; the 2^20 x 1024 scans are checked against an
independent fp64 scan (torch matmul in float64 on the same device — not the library), through the sample-threshold
filter path the library takes for large maps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c1_single_frame_vit_b16_512d_grid128():
    """configs[0]: a single 320x240 RGB-D frame, random-weight ViT-B/16 with a 512-D output head, 128^3 grid.  The frame goes
    through the library's encoder (u8 frame -> fused preprocessing -> ViT -> 14x14x512 tokens) and the same tokens through
    the sequential oracle: reference-exact mode at depth_sample_rate 1 and 50 (token cache rows, ids, rgb with host alpha,
    top-down map, then flush + top-K) and the dense mean mode."""
    import random
    import torch
    import bsc_nav_amd as B
    import golden_util as gu
    import synth
    from bsc_nav_amd import encoder
    from oracle import oracle as orc
    H, W, g, D, gs = 240, 320, 14, 512, 128
    rgb, depth, poses = synth.make_frames(41, 1, H, W, "room")
    vit = encoder.RandomViT("vit_b16", image_size=224, out_dim=D, seed=2).cuda()
    tok = vit.patch_tokens(torch.from_numpy(rgb).cuda()).contiguous()              # (1, 14, 14, 512) f32
    assert tok.shape == (1, g, g, D) and tok.dtype == torch.float32
    tokens = tok.cpu().numpy()
    T = B.PoseChain().pc_transform(poses[0])
    for rate in (1, 50):
        idx = np.random.RandomState(rate).permutation(H * W)[::rate].astype(np.int32)
        kw = dict(iter_size=5000)
        eng = B.VoxelEngine(H, W, gs, 0.1, -6.4, 6.4, g, D, mode="exact", voxel_capacity=60_000, token_capacity=200_000,
                            max_points=H * W, **kw)
        oc = orc.make_config(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=0, **kw)
        om = orc.OracleMemory(oc, voxel_capacity=60_000)
        al = np.exp(-orc.geometry(oc, depth[0], idx, T)["r2"] / (2 * 0.6))
        random.seed(7); om.ingest_frame(depth[0], rgb[0], idx, T, tokens[0], al); om.flush()
        random.seed(7)
        eng.ingest(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), tok, T[None], torch.from_numpy(idx).cuda(),
                   np.array([0, len(idx)], np.int64), torch.from_numpy(al).cuda())
        eng.flush()
        assert eng.counters()["max_id"] == om.counters()["max_id"] > 500
        for a, b in zip(eng.export_rgb(), om.export_rgb()):
            assert np.array_equal(a, b)
        for a, b in zip(eng.export_heightmap(), om.export_heightmap()):
            assert np.array_equal(a, b)
        for a, b in zip(eng.export_store(), om.export_store()):
            assert np.array_equal(a, b)
        q = torch.from_numpy(tokens[0, 5, 7].copy()).cuda()[None]
        pos, sim, n = eng.localize(q, K=100)
        opos, osim = om.localize(tokens[0, 5, 7], K=100)
        gu.assert_topk_matches(pos[0, :n[0]], sim[0, :n[0]], opos, osim, tol=5e-6)
        eng.close()
    eng = B.VoxelEngine(H, W, gs, 0.1, -6.4, 6.4, g, D, mode="mean", voxel_capacity=60_000, max_points=H * W)
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -6.4, 6.4, g, D, mode=1), voxel_capacity=60_000)
    om.ingest_frame(depth[0], rgb[0], None, T, tokens[0])
    eng.ingest(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), tok, T[None])
    (acc, cnt), (oacc, ocnt) = eng.export_dense(), om.export_dense()
    assert np.array_equal(eng.export_rgb()[0], om.export_rgb()[0]) and np.array_equal(cnt, ocnt)
    c = np.maximum(cnt, 1)[:, None].astype(np.float64)
    np.testing.assert_allclose(acc / c, oacc / c, rtol=1e-3, atol=1e-3)
    eng.close()


def test_c3_dense_ingest_vit_l14_tokens_grid512_against_oracle():
    """640x480 frames, 16x16x1024 tokens (ViT-L/14 @224, memory_2.py:107 token_dim 1024), 512^3 grid of 0.1 m cells,
    every pixel, 4 frames in two calls: ids / rgb / weights / top-down map / counts bit-exact, means within 1e-3."""
    from test_gpu_edges import _dense_vs_oracle
    longest, n_vox = _dense_vs_oracle(480, 640, 16, 1024, 512, 0.1, -25.6, 25.6, F=4, per_call=2, seed=33, vcap=400_000)
    assert n_vox > 5000


def test_c3_dense_max_mode_1024d_against_oracle():
    """Same sizes, element-wise max reduce (the scatter-max of the north star): rows bit-exact."""
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    H, W, g, D, gs, F = 480, 640, 16, 1024, 512, 2
    rgb, depth, poses = synth.make_frames(34, F, H, W, "room")
    tokens = synth.make_tokens(34, F, g, D)
    eng = B.VoxelEngine(H, W, gs, 0.1, -25.6, 25.6, g, D, mode="max", voxel_capacity=300_000, max_points=F * H * W)
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -25.6, 25.6, g, D, mode=2), voxel_capacity=300_000)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    eng.ingest(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(tokens).cuda(), Ts)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], None, Ts[f], tokens[f])
    (acc, cnt), (oacc, ocnt) = eng.export_dense(), om.export_dense()
    assert np.array_equal(eng.export_rgb()[0], om.export_rgb()[0]) and np.array_equal(cnt, ocnt)
    assert np.array_equal(acc, oacc)
    eng.close()


def test_c3_exact_mode_1024d_tokens_against_oracle():
    """Reference-exact token cache at D=1024, g=16 (the reference's own encoder shape): cache rows, flush with
    replacement draws, store — bit-exact against the oracle over several in-loop flushes."""
    import random
    import torch
    import bsc_nav_amd as B
    import synth
    from oracle import oracle as orc
    H, W, g, D, gs, F, s = 240, 320, 16, 1024, 256, 6, 5
    rgb, depth, poses = synth.make_frames(35, F, H, W, "room")
    tokens = synth.make_tokens(35, F, g, D)
    kw = dict(iter_size=6000)
    eng = B.VoxelEngine(H, W, gs, 0.1, -12.8, 12.8, g, D, mode="exact", voxel_capacity=100_000, token_capacity=400_000,
                        max_points=H * W, **kw)
    om = orc.OracleMemory(orc.make_config(H, W, gs, 0.1, -12.8, 12.8, g, D, mode=0, **kw), voxel_capacity=100_000)
    chain = B.PoseChain()
    Ts = np.stack([chain.pc_transform(p) for p in poses])
    rs = np.random.RandomState(1)
    idxs = [np.ascontiguousarray(rs.permutation(H * W)[::s].astype(np.int32)) for _ in range(F)]
    random.seed(123)
    for f in range(F):
        eng.ingest(torch.from_numpy(depth[f:f + 1]).cuda(), torch.from_numpy(rgb[f:f + 1]).cuda(),
                   torch.from_numpy(tokens[f:f + 1]).cuda(), Ts[f:f + 1], torch.from_numpy(idxs[f]).cuda(),
                   np.array([0, len(idxs[f])], np.int64))
    eng.flush()
    st_dev = eng.export_store()
    k = eng.counters()
    random.seed(123)
    for f in range(F):
        om.ingest_frame(depth[f], rgb[f], idxs[f], Ts[f], tokens[f])
    om.flush()
    ok = om.counters()
    assert k["flushes"] == ok["flushes"] and k["flushes"] >= 10 and k["max_id"] == ok["max_id"]
    for a, b in zip(st_dev, om.export_store()):
        assert np.array_equal(a, b)
    eng.close()


def _big_map(torch, V, D, gs, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    codes = torch.randperm(gs ** 3, device="cuda", generator=gen)[:V]
    keys = torch.stack([codes // (gs * gs), (codes // gs) % gs, codes % gs], dim=1).to(torch.int32).contiguous()
    rows = torch.randn((V, D), device="cuda", generator=gen)
    return keys, rows, gen


def _fp64_topk(torch, rows, q, K, mask=None, chunk=1 << 17):
    """Independent scan: cosine in float64, chunked; -> (idx (Q,K), sim (Q,K)) by descending similarity."""
    qn = (q.double() / q.double().norm(dim=1, keepdim=True).clamp_min(1e-8))
    sims = torch.empty((q.shape[0], rows.shape[0]), dtype=torch.float64, device=rows.device)
    for a in range(0, rows.shape[0], chunk):
        r = rows[a:a + chunk].double()
        sims[:, a:a + chunk] = qn @ (r / r.norm(dim=1, keepdim=True).clamp_min(1e-8)).T
    if mask is not None:
        sims[:, ~mask] = -2.0
    top = torch.topk(sims, K, dim=1)
    return top.indices, top.values


@pytest.mark.parametrize("Q", [1, 8, 256])
def test_c4_c5_localize_2pow20_x_1024_matches_fp64_scan(Q):
    """C4 / C5 size: 2^20 voxels x 1024-D inside a 512^3 grid, K=100, Q = 1 (text/patch query), 8, 256 (image-goal
    batch): the library's scan + sample-threshold filter + top-K against the fp64 scan; scores within 2e-6 (north star: 1e-3), same voxels in the same order up to fp32 near-ties."""
    import torch
    import bsc_nav_amd as B
    V, D, gs, K = 1 << 20, 1024, 512, 100
    keys, rows, gen = _big_map(torch, V, D, gs, 7)
    eng = B.VoxelEngine(48, 64, gs, 0.1, -25.6, 25.6, 16, D, mode="mean", voxel_capacity=V + 8, max_points=4096)
    eng.dense_replace(keys, rows, torch.ones(V, dtype=torch.int32, device="cuda"))
    q = torch.randn((Q, D), device="cuda", generator=gen)
    pos, sim, n = eng.localize(q, K=K)
    idx, ref = _fp64_topk(torch, rows, q, K)
    kk = keys.cpu().numpy()
    import golden_util as gu
    for i in range(Q):
        assert n[i] == K
        gu.assert_topk_near(pos[i], sim[i], kk[idx[i].cpu().numpy()], ref[i].cpu().numpy(), tol=2e-6)
    # region + floor filters at this size (memory_2.py:624-640): candidates within a radius of curr and a z-range
    curr, radius, floor = [256, 256, 256], 120.0, (100, 400)
    pos, sim, n = eng.localize(q[:min(Q, 8)], K=K, radius=radius, curr=curr, floor=floor)
    k64 = keys.to(torch.int64)
    mask = (((k64 - torch.tensor(curr, device="cuda")) ** 2).sum(1) <= radius * radius) & (k64[:, 2] >= floor[0]) & (k64[:, 2] <= floor[1])
    idx, ref = _fp64_topk(torch, rows, q[:min(Q, 8)], K, mask)
    for i in range(min(Q, 8)):
        assert n[i] == K
        gu.assert_topk_near(pos[i], sim[i], kk[idx[i].cpu().numpy()], ref[i].cpu().numpy(), tol=2e-6)
    eng.close()


def test_c5_voxel_sharded_localize_equals_single_map():
    """C5's partitioning on one GPU: the 2^20 x 1024 map split into 8 voxel shards (one engine each, as 8 ranks would
    hold them), 256 queries scanned per shard, K-way merge in the reference's tie order (dist.merge_topk) == the scan of
    the whole map."""
    import torch
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bd
    V, D, gs, K, Q, R = 1 << 20, 1024, 512, 100, 256, 8
    keys, rows, gen = _big_map(torch, V, D, gs, 8)
    q = torch.randn((Q, D), device="cuda", generator=gen)
    ones = torch.ones(V, dtype=torch.int32, device="cuda")
    full = B.VoxelEngine(48, 64, gs, 0.1, -25.6, 25.6, 16, D, mode="mean", voxel_capacity=V + 8, max_points=4096)
    full.dense_replace(keys, rows, ones)
    p0, s0, n0 = full.localize(q, K=K)
    full.close()
    per = V // R
    shard = B.VoxelEngine(48, 64, gs, 0.1, -25.6, 25.6, 16, D, mode="mean", voxel_capacity=per + 8, max_points=4096)
    parts = []
    for r in range(R):
        sl = slice(r * per, (r + 1) * per)
        shard.dense_replace(keys[sl].contiguous(), rows[sl].contiguous(), ones[sl].contiguous())
        parts.append(shard.localize(q, K=K))
    shard.close()
    for i in range(Q):
        p, s = bd.merge_topk([pp[0][i, :pp[2][i]] for pp in parts], [pp[1][i, :pp[2][i]] for pp in parts], K)
        assert np.array_equal(p, p0[i]) and np.array_equal(s, s0[i])


def _store_shape_map(torch, V, D, gs, seed, n_dup=48):
    """C4 in the reference's store shape: V voxels with M ~ U{1..10} raw tokens each (memory_2.py:642-663
    takes the max over a voxel's tokens).  `n_dup` voxels carry one and the same token among theirs, so a query near that
    token meets a block of EXACT ties at the head of the ranking (the reference's name-order case, memory_2.py:665)."""
    gen = torch.Generator(device="cuda").manual_seed(seed)
    codes = torch.randperm(gs ** 3, device="cuda", generator=gen)[:V]
    keys = torch.stack([codes // (gs * gs), (codes // gs) % gs, codes % gs], dim=1).to(torch.int32).contiguous()
    cnt = torch.randint(1, 11, (V,), device="cuda", generator=gen, dtype=torch.int32)
    off = torch.zeros(V + 1, dtype=torch.int64, device="cuda")
    off[1:] = torch.cumsum(cnt.to(torch.int64), 0)
    T = int(off[-1].item())
    rows = torch.empty((T, D), dtype=torch.float32, device="cuda")
    for a in range(0, T, 1 << 20):
        rows[a:a + (1 << 20)] = torch.randn((min(1 << 20, T - a), D), device="cuda", generator=gen)
    dup_vox = torch.randperm(V, device="cuda", generator=gen)[:n_dup]
    special = torch.randn(D, device="cuda", generator=gen)
    rows[off[dup_vox] + (cnt[dup_vox].to(torch.int64) - 1)] = special           # the last token of each of these voxels
    seg = torch.repeat_interleave(torch.arange(V, device="cuda"), cnt.to(torch.int64))
    return keys, cnt, off, rows, seg, special, gen


def _fp64_voxel_ranking(torch, rows, seg, V, q, keys_np, K, mask=None, slack=64, chunk=1 << 18):
    """Independent scan in the reference's terms: cosine in float64 per token, max per voxel (memory_2.py:655-661), stable
    descending sort over name-ordered candidates (:665) -> (pos (K,3), sim (K,)) per query."""
    from bsc_nav_amd import dist as bd
    qn = q.double() / q.double().norm(dim=1, keepdim=True).clamp_min(1e-8)
    best = torch.full((q.shape[0], V), -2.0, dtype=torch.float64, device=rows.device)
    for a in range(0, rows.shape[0], chunk):
        r = rows[a:a + chunk].double()
        s = qn @ (r / r.norm(dim=1, keepdim=True).clamp_min(1e-8)).T
        best.scatter_reduce_(1, seg[a:a + chunk].expand(q.shape[0], -1), s, reduce="amax")
    if mask is not None:
        best[:, ~mask] = -3.0
    top = torch.topk(best, K + slack, dim=1)
    out = []
    for i in range(q.shape[0]):
        idx, sim = top.indices[i].cpu().numpy(), top.values[i].cpu().numpy()
        k0, k1, k2 = bd.name_keys_np(keys_np[idx])
        order = np.lexsort((k2, k1, k0, -sim))
        assert sim[order][K - 1] > sim[order][-1], "tie group runs past the candidate slack"
        out.append((keys_np[idx[order[:K]]], sim[order[:K]]))
    return out


def test_c4_store_shape_2pow20_voxels_ragged_tokens_matches_fp64_scan():
    """BASELINE configs[3] / SURVEY.md §8d "C4": localize over V = 2^20 voxels x 1024-D inside a 512^3 grid in the
    REFERENCE'S store shape — M ~ U{1..10} raw tokens per voxel (sum M ~ 5.8 M rows, ~23.6 GB), loaded through
    bsc_import_store like a reference-built memory — Q = 1 and 8, with and without the region + floor filters
    (memory_2.py:624-640): cosine per token, per-voxel max, stable top-K in HDF5 name order, against the fp64 scan.
    Scores within 2e-6 (north star: 1e-3); the block of exact ties at the head keeps name order."""
    import torch
    import bsc_nav_amd as B
    import golden_util as gu
    V, D, gs, K = 1 << 20, 1024, 512, 100
    keys, cnt, off, rows, seg, special, gen = _store_shape_map(torch, V, D, gs, 11)
    T = rows.shape[0]
    assert 5_000_000 < T < 6_500_000
    eng = B.VoxelEngine(48, 64, gs, 0.1, -25.6, 25.6, 16, D, mode="exact", iter_size=256, voxel_capacity=V + 8,
                        token_capacity=T, max_points=4096)
    kk = keys.cpu().numpy()
    eng.import_rgb(kk, np.zeros((V, 3), np.uint8), np.ones(V, np.float32))
    eng.import_store(kk, cnt.cpu().numpy(), rows.cpu().numpy(), np.zeros(T, np.float32))
    c = eng.counters()
    assert c["store_voxels"] == V and c["store_tokens"] == T
    q = torch.randn((8, D), device="cuda", generator=gen)
    q[0] = special + 0.02 * q[0]                                   # heads the ranking with the 48 tied voxels
    for Q in (1, 8):
        pos, sim, n = eng.localize(q[:Q], K=K)
        ref = _fp64_voxel_ranking(torch, rows, seg, V, q[:Q], kk, K)
        for i in range(Q):
            assert n[i] == K
            gu.assert_topk_matches(pos[i], sim[i], ref[i][0], ref[i][1], tol=2e-6)
        # the block of exact ties heads query 0's ranking, in HDF5 link-name order ('grid_19_..' < 'grid_1_..', memory_2.py:665)
        assert len(set(sim[0][:48].tolist())) == 1 and sim[0][48] < sim[0][47]
        names = ["grid_%d_%d_%d" % tuple(r) for r in pos[0][:48].tolist()]
        assert names == sorted(names)
    curr, radius, floor = [256, 256, 256], 150.0, (80, 420)
    k64 = keys.to(torch.int64)
    mask = (((k64 - torch.tensor(curr, device="cuda")) ** 2).sum(1) <= radius * radius) & (k64[:, 2] >= floor[0]) & (k64[:, 2] <= floor[1])
    pos, sim, n = eng.localize(q, K=K, radius=radius, curr=curr, floor=floor)
    ref = _fp64_voxel_ranking(torch, rows, seg, V, q, kk, K, mask)
    for i in range(8):
        assert n[i] == K
        gu.assert_topk_matches(pos[i], sim[i], ref[i][0], ref[i][1], tol=2e-6)
    eng.close()


@pytest.mark.parametrize("Q", [70, 130, 300])
def test_batched_cosine_on_bf16_pieces_keeps_f32_accuracy(Q):
    """From 65 queries on the scan runs on the 16-bit matrix cores with every f32 operand split into pieces (round 5: two fp16
    pieces, three MFMAs per product, per-row power-of-two scales; rounds 3-4: three bf16 pieces, six MFMAs), f32 accumulation:
    rows whose scale spans eight decades (cosine is scale-free, the split must be too) and elements of mixed magnitude, against
    the fp64 scan — scores within 3e-6, same voxels in the same order up to near-ties."""
    import torch
    import bsc_nav_amd as B
    import golden_util as gu
    V, D, gs, K = 1 << 16, 1024, 128, 100
    keys, rows, gen = _big_map(torch, V, D, gs, 21)
    rows = rows * torch.pow(10.0, torch.rand((V, 1), device="cuda", generator=gen) * 8 - 3)          # per-row scale 1e-3 .. 1e5
    rows = rows * torch.pow(10.0, -3 * torch.rand((1, D), device="cuda", generator=gen))              # per-column 1e-3 .. 1
    eng = B.VoxelEngine(48, 64, gs, 0.1, -6.4, 6.4, 16, D, mode="mean", voxel_capacity=V + 8, max_points=4096)
    eng.dense_replace(keys, rows.contiguous(), torch.ones(V, dtype=torch.int32, device="cuda"))
    q = torch.randn((Q, D), device="cuda", generator=gen) * torch.pow(10.0, -2 * torch.rand((1, D), device="cuda", generator=gen))
    pos, sim, n = eng.localize(q, K=K)
    idx, ref = _fp64_topk(torch, rows, q, K)
    kk = keys.cpu().numpy()
    for i in range(Q):
        assert n[i] == K
        gu.assert_topk_near(pos[i], sim[i], kk[idx[i].cpu().numpy()], ref[i].cpu().numpy(), tol=3e-6)
    eng.close()


def test_batched_cosine_row_scales_follow_the_map(monkeypatch):
    """The fp16-piece scan keeps per-row scales / inverse norms until the rows change: a replaced map, an imported one and a map that
    grew by an ingest must each be scanned with fresh scales (scores against the fp64 scan of the CURRENT rows), zero
    rows score 0 like the other scans, rows from 1e-15 to 1e15 keep f32 accuracy, and the six-product bf16 scan (A/B switch)
    agrees."""
    import torch
    import bsc_nav_amd as B
    import golden_util as gu
    V, D, gs, K, Q = 20000, 768, 64, 50, 130
    keys, rows, gen = _big_map(torch, V, D, gs, 33)
    eng = B.VoxelEngine(48, 64, gs, 0.1, -3.2, 3.2, 14, D, mode="mean", voxel_capacity=V + 4096, max_points=4096)
    ones = torch.ones(V, dtype=torch.int32, device="cuda")
    q = torch.randn((Q, D), device="cuda", generator=gen)
    kk = keys.cpu().numpy()

    def check(r, tol=3e-6, kk=kk):
        pos, sim, n = eng.localize(q, K=K)
        idx, ref = _fp64_topk(torch, r, q, K)
        for i in range(0, Q, 13):
            assert n[i] == K
            gu.assert_topk_near(pos[i], sim[i], kk[idx[i].cpu().numpy()], ref[i].cpu().numpy(), tol=tol)
        return sim

    eng.dense_replace(keys, rows, ones)
    check(rows)
    # the same voxels, every row rescaled by its own factor spread over thirty decades, some rows zero
    r2 = rows * torch.pow(10.0, torch.rand((V, 1), device="cuda", generator=gen) * 30 - 15)
    r2[::97] = 0
    r2 = r2[torch.randperm(V, device="cuda", generator=gen)].contiguous()      # other rows under the same keys: stale scales would be wrong
    eng.dense_replace(keys, r2, ones)
    s_f16 = check(r2)
    monkeypatch.setenv("BSC_COSINE_BF16", "1")
    s_bf16 = check(r2)
    monkeypatch.delenv("BSC_COSINE_BF16")
    assert np.abs(s_f16 - s_bf16).max() < 3e-6
    # import through the host path
    r3 = torch.randn((V, D), device="cuda", generator=gen) * 1e-3
    eng.import_rgb(kk, np.zeros((V, 3), np.uint8), np.ones(V, np.float32))
    eng.import_dense(r3.cpu().numpy(), np.ones(V, np.int32))
    check(r3)
    # an ingest on top: touched rows change, new voxels appear
    import synth
    rgb, depth, poses = synth.make_frames(7, 1, 48, 64, "room")
    T = B.PoseChain().pc_transform(poses[0])[None]
    tok = torch.randn((1, 14, 14, D), device="cuda", generator=gen)
    eng.ingest(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), tok, T)
    acc, cnt = eng.export_dense()
    assert (cnt > 0).all() and len(cnt) > V
    check(torch.from_numpy(acc).cuda(), kk=eng.export_rgb()[0])
    eng.close()
