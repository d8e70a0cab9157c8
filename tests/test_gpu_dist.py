"""Frame-sharded build with REAL engines: two processes (gloo, 127.0.0.1) share the one GPU of the test box, each ingests
its contiguous frame block through libbscnav, the maps are merged (dist.merge_dense_maps), collected on rank 0
(dist.gather_merged_to_root), saved as a memory directory — and that directory, loaded into a fresh single-process
object, equals the memory one process builds from all frames: ids / positions / occupied_ids / counts / top-down map
bit-exact, features within 1e-3, weights within f32 rounding, rgb by the documented merge rule, identical top-K
(the reference itself is single-process, memory_2.py:888-903 / :1136-1145 define the state)."""
import os
import socket

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

H, W, G, D, GS, F = 96, 128, 8, 32, 128, 8


def _args(tmp, name):
    import bsc_nav_amd as B
    return B.MemoryArgs(width=W, height=H, grid_size=GS, cell_size=0.1, floor_height=-6.4, map_height=6.4,
                        depth_sample_rate=1, query_width=G * 14, query_height=G * 14, token_dim=D, memory_path=str(tmp),
                        scene_name=name)


def _inputs(nf=F):
    import synth
    rgb, depth, poses = synth.make_frames(41, nf, H, W, "room")
    tokens = synth.make_tokens(41, nf, G, D)
    return rgb, depth, poses, tokens


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir, nf=F):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bd
    rgb, depth, poses, tokens = _inputs(nf)
    mem = B.VoxelTokenMemory(_args(out_dir, "merged"), need_diffusion=False, feature_mode=mode, max_frames_per_call=nf,
                             voxel_capacity=100_000)
    mem.set_map_origin(poses[0])                                   # every rank writes into the scene's map frame
    a, b = bd.shard_frames(nf)
    dev = lambda x: torch.from_numpy(x[a:b]).cuda().contiguous()   # noqa: E731
    mem.ingest_frames(dev(rgb), dev(depth), poses[a:b], tokens=dev(tokens))
    mem.base_height.append(float(rank))
    mem.long_memory_dict.append({"label": "chair", "loc": [10 * rank, 5, 5], "confidence": 0.5 + 0.1 * rank})
    local_voxels = mem.max_id
    lpos, lrgb, lwgt = mem.engine.export_rgb()                     # this rank's own colour state, before the merge
    info = bd.merge_dense_maps(mem.engine)                         # rows now distributed: slice `rank` of the global order
    q = torch.from_numpy(np.random.RandomState(5).standard_normal((2, D)).astype(np.float32)).cuda()
    sp, ss = bd.localize_sharded(mem.engine, q, K=50)              # every rank scans its slice, K-way merge
    # the public entry: merge (again: merging an already merged map must change nothing) + collect on rank 0
    is_root = mem.merge_shards(root=0)
    if is_root:
        mem.initial_memory()
        mem.save_memory(original_pos=np.asarray(poses[0][:3], np.float32))
        mh, cv = mem.engine.export_heightmap()
        np.savez(f"{out_dir}/root.npz", path=np.array(mem.memory_save_path), mh=mh, cv=cv, n_union=info["n_union"])
    np.savez(f"{out_dir}/r{rank}.npz", local_voxels=local_voxels, n_local=info["n_local"], per=info["per_rank"],
             sp0=sp[0], ss0=ss[0], sp1=sp[1], ss1=ss[1], frames=b - a, lpos=lpos, lrgb=lrgb, lwgt=lwgt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["mean", "max"])
def test_two_process_sharded_build_equals_single_process(tmp_path, mode):
    import torch
    import torch.multiprocessing as mp
    import bsc_nav_amd as B
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    root = np.load(f"{tmp_path}/root.npz")
    ranks = [np.load(f"{tmp_path}/r{r}.npz") for r in range(world)]
    # ---- the same frames through ONE process ----
    rgb, depth, poses, tokens = _inputs()
    one = B.VoxelTokenMemory(_args(tmp_path, "single"), need_diffusion=False, feature_mode=mode, max_frames_per_call=F,
                             voxel_capacity=100_000)
    one.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
    n = one.max_id
    assert int(root["n_union"]) == n and sum(int(r["n_local"]) for r in ranks) == n
    assert all(int(r["local_voxels"]) < n for r in ranks)          # the shards really saw different voxel sets
    # ---- the saved merged directory, loaded like any memory ----
    args = _args(tmp_path, "loaded")
    args.load_memory_path = str(root["path"])
    got = B.VoxelTokenMemory(args, need_diffusion=False, feature_mode=mode, voxel_capacity=100_000)
    got.load_memory()
    assert got.max_id == n
    assert np.array_equal(got.grid_rgb_pos, one.grid_rgb_pos)       # ids in the single-process first-touch order
    assert np.array_equal(got.occupied_ids, one.occupied_ids)
    (gacc, gcnt), (oacc, ocnt) = got.engine.export_dense(), one.engine.export_dense()
    assert np.array_equal(gcnt, ocnt)                               # counts exact
    if mode == "max":
        assert np.array_equal(gacc, oacc)
    else:
        c = np.maximum(ocnt, 1)[:, None].astype(np.float64)
        np.testing.assert_allclose(gacc / c, oacc / c, rtol=1e-3, atol=1e-3)
    omh, ocv = one.engine.export_heightmap()
    assert np.array_equal(root["mh"], omh) and np.array_equal(root["cv"], ocv)      # top-down map exact
    np.testing.assert_allclose(got.weight, one.weight, rtol=2e-6)                  # f32 sums in a different order
    # rgb: documented dense-mode rule (a rank's state enters the running mean as one observation), not the sequential
    # truncating chain; voxels seen by one rank only are bit-exact, the others stay close
    drgb = np.abs(got.grid_rgb.astype(np.int32) - one.grid_rgb.astype(np.int32))
    assert (drgb == 0).mean() > 0.5 and drgb.mean() < 2.0 and np.percentile(drgb, 99) <= 16
    # the analytic bound of the rule: every merge step is the chain's own update — a truncated convex combination of
    # integer colours whose product c*w is rounded in f32, which can land an ulp under c and truncate to c - 1 — so the merged
    # colour lies between (the smallest of the ranks' own colours of that voxel) - (merge steps) and the largest, channel by
    # channel; it equals the rank's colour where only one rank saw the voxel; the weights are the sum of the ranks' weights
    code = lambda p: (p[:, 0].astype(np.int64) << 42) | (p[:, 1].astype(np.int64) << 21) | p[:, 2].astype(np.int64)   # noqa: E731
    gcode = code(got.grid_rgb_pos)
    order = np.argsort(gcode)
    lo = np.full((n, 3), 255, np.int32); hi = np.zeros((n, 3), np.int32); seen = np.zeros(n, np.int32); wsum = np.zeros(n, np.float64)
    for r in ranks:
        at = order[np.searchsorted(gcode[order], code(r["lpos"]))]
        assert np.array_equal(gcode[at], code(r["lpos"]))
        lo[at] = np.minimum(lo[at], r["lrgb"]); hi[at] = np.maximum(hi[at], r["lrgb"]); seen[at] += 1; wsum[at] += r["lwgt"]
    g = got.grid_rgb.astype(np.int32)
    assert (seen >= 1).all() and ((g >= lo - (seen[:, None] - 1)) & (g <= hi)).all()
    assert (seen == 1).any() and np.array_equal(g[seen == 1], one.grid_rgb[seen == 1].astype(np.int32))
    np.testing.assert_allclose(got.weight, wsum, rtol=2e-6)
    assert np.load(str(root["path"]) + "/base_height.npy").tolist() == [0.0, 1.0]       # rank order
    assert sorted(o["loc"][0] for o in got.long_memory_dict) == [0, 10]
    # ---- identical top-K: sharded scan (before the gather), the loaded merged memory, the single-process memory ----
    q = torch.from_numpy(np.random.RandomState(5).standard_normal((2, D)).astype(np.float32)).cuda()
    p1, s1, n1 = one.engine.localize(q, K=50)
    p2, s2, n2 = got.engine.localize(q, K=50)
    for qi in range(2):
        assert n1[qi] == n2[qi] == 50
        gu.assert_topk_near(p2[qi], s2[qi], p1[qi], s1[qi], tol=5e-6)
        for r in ranks:                                             # both ranks hold the same merged answer
            gu.assert_topk_near(r[f"sp{qi}"], r[f"ss{qi}"], p1[qi], s1[qi], tol=5e-6)


def test_eight_process_sharded_build_with_uneven_shards(tmp_path):
    """The target world size: EIGHT ranks (gloo stand-in, all on the test box's one GPU) with 11 frames — shards of 2 and 1
    frames — and a voxel union that is not a multiple of 8 (sentinel rows in the reduce-scatter layout): merge_dense_maps +
    gather_merged_to_root + localize_sharded give the single-process memory (ids, positions, counts, top-down map bit-exact,
    features 1e-3, the same top-K on every rank)."""
    import torch
    import torch.multiprocessing as mp
    import bsc_nav_amd as B
    world, mode = 8, "mean"
    for nf in (11, 13, 10, 12, 14):        # a frame count whose voxel union is not a multiple of 8 (11 frames: 11 952 = 8 x 1 494)
        rgb, depth, poses, tokens = _inputs(nf)
        one = B.VoxelTokenMemory(_args(tmp_path, f"single8_{nf}"), need_diffusion=False, feature_mode=mode, max_frames_per_call=nf,
                                 voxel_capacity=100_000)
        one.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
        n = one.max_id
        if n % world != 0 and nf % world != 0:
            break
        one.engine.close()
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), nf), nprocs=world, join=True)
    root = np.load(f"{tmp_path}/root.npz")
    ranks = [np.load(f"{tmp_path}/r{r}.npz") for r in range(world)]
    assert sorted(int(r["frames"]) for r in ranks) == sorted(nf // world + (1 if r < nf % world else 0) for r in range(world))
    assert len({int(r["frames"]) for r in ranks}) == 2, "uneven shards"
    per = int(ranks[0]["per"])
    assert int(root["n_union"]) == n and sum(int(r["n_local"]) for r in ranks) == n
    assert n % world != 0 and per * world > n, "the case must exercise the sentinel rows of the last slice"
    assert all(int(r["per"]) == per for r in ranks) and [int(r["n_local"]) for r in ranks][:-1] == [per] * (world - 1)
    args = _args(tmp_path, "loaded8")
    args.load_memory_path = str(root["path"])
    got = B.VoxelTokenMemory(args, need_diffusion=False, feature_mode=mode, voxel_capacity=100_000)
    got.load_memory()
    assert got.max_id == n
    assert np.array_equal(got.grid_rgb_pos, one.grid_rgb_pos) and np.array_equal(got.occupied_ids, one.occupied_ids)
    (gacc, gcnt), (oacc, ocnt) = got.engine.export_dense(), one.engine.export_dense()
    assert np.array_equal(gcnt, ocnt)
    c = np.maximum(ocnt, 1)[:, None].astype(np.float64)
    np.testing.assert_allclose(gacc / c, oacc / c, rtol=1e-3, atol=1e-3)
    omh, ocv = one.engine.export_heightmap()
    assert np.array_equal(root["mh"], omh) and np.array_equal(root["cv"], ocv)
    np.testing.assert_allclose(got.weight, one.weight, rtol=4e-6)
    q = torch.from_numpy(np.random.RandomState(5).standard_normal((2, D)).astype(np.float32)).cuda()
    p1, s1, n1 = one.engine.localize(q, K=50)
    for qi in range(2):
        assert n1[qi] == 50
        for r in ranks:                                             # all eight ranks hold the same merged answer
            gu.assert_topk_near(r[f"sp{qi}"], r[f"ss{qi}"], p1[qi], s1[qi], tol=5e-6)


@pytest.mark.parametrize("mode", ["mean", "max"])
def test_rccl_world1_merge_gather_localize(tmp_path, mode):
    """The RCCL branches of dist.py on the one GPU of the test box: a `nccl` process group of ONE rank drives
    merge_dense_maps / gather_merged_to_root / localize_sharded through `reduce_scatter_tensor` (f32 SUM or MAX, int32 SUM),
    device `all_gather`, `dist.gather` and the ragged key all-gather — no gloo fallback is taken (asserted on the backend).
    With one rank every exchange is the identity, so the merged memory must equal the memory before the merge bit for bit
    (ids, positions, feature rows, counts, rgb, weights, top-down map) and the sharded top-K the plain top-K."""
    import torch
    import torch.distributed as dist
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bd
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl" and bd._active()
        rgb, depth, poses, tokens = _inputs()
        mem = B.VoxelTokenMemory(_args(tmp_path, "w1"), need_diffusion=False, feature_mode=mode, max_frames_per_call=F,
                                 voxel_capacity=100_000)
        mem.set_map_origin(poses[0])
        mem.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
        bd.warmup_collectives(torch.device("cuda", 0))
        before = (mem.engine.export_rgb(), mem.engine.export_dense(), mem.engine.export_heightmap(), mem.engine.export_occupied())
        q = torch.from_numpy(np.random.RandomState(5).standard_normal((3, D)).astype(np.float32)).cuda()
        p0, s0, n0 = mem.engine.localize(q, K=40)
        info = bd.merge_dense_maps(mem.engine)
        n = before[0][0].shape[0]
        assert {k: info[k] for k in ("n_union", "per_rank", "n_local")} == dict(n_union=n, per_rank=n, n_local=n)
        sp, ss = bd.localize_sharded(mem.engine, q, K=40)
        assert bd.gather_merged_to_root(mem.engine, info) is True
        assert mem.merge_shards() is True                 # the public entry: merge + gather + long-memory lists
        after = (mem.engine.export_rgb(), mem.engine.export_dense(), mem.engine.export_heightmap(), mem.engine.export_occupied())
        for a, b in zip(before, after):
            for x, y in zip(a, b) if isinstance(a, tuple) else ((a, b),):
                assert np.array_equal(x, y)
        for qi in range(3):
            assert np.array_equal(sp[qi], p0[qi, :n0[qi]]) and np.array_equal(ss[qi], s0[qi, :n0[qi]])
        # the raw collectives on known data
        rows = torch.arange(24, dtype=torch.float32, device="cuda").reshape(6, 4)
        assert torch.equal(bd.reduce_scatter_rows(rows, dist.ReduceOp.MAX, 6), rows)
        assert torch.equal(bd.reduce_scatter_rows(rows.to(torch.int32), dist.ReduceOp.SUM, 6), rows.to(torch.int32))
        parts = bd.all_gather_ragged(torch.arange(5, dtype=torch.int64, device="cuda"))
        assert len(parts) == 1 and parts[0].tolist() == [0, 1, 2, 3, 4]
        mem.engine.close()
    finally:
        dist.destroy_process_group()


# ---- exact colour across ranks: sub-sampled build, point log, owner-side replay ---------------------------
RATE = 50


def _sampled_args(tmp, name, rate=RATE):
    a = _args(tmp, name)
    a.depth_sample_rate = rate
    return a


def _replay_worker(rank, world, port, out_dir, rate=RATE):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bsc_nav_amd as B
    from bsc_nav_amd import dist as bd
    from bsc_nav_amd.geometry import sample_indices_fast
    rgb, depth, poses, tokens = _inputs()
    mem = B.VoxelTokenMemory(_sampled_args(out_dir, "merged", rate), need_diffusion=False, feature_mode="mean", max_frames_per_call=F,
                             voxel_capacity=100_000)
    mem.enable_point_log(F * H * W)
    mem.set_map_origin(poses[0])
    a, b = bd.shard_frames(F)
    np.random.seed(3)
    for _ in range(a if rate > 1 else 0):                           # the shuffles the earlier ranks' frames consume
        sample_indices_fast(H * W, rate)
    dev = lambda x: torch.from_numpy(x[a:b]).cuda().contiguous()   # noqa: E731
    mem.ingest_frames(dev(rgb), dev(depth), poses[a:b], tokens=dev(tokens))
    if mem.merge_shards(root=0):
        pos, c, w = mem.engine.export_rgb()
        acc, cnt = mem.engine.export_dense()
        np.savez(f"{out_dir}/replay_root.npz", pos=pos, rgb=c, w=w, cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rate", [RATE, 1])
def test_two_process_colour_replay_is_bit_exact(tmp_path, rate):
    """depth_sample_rate = 50 and 1 (EVERY pixel: the dense build of the north star), two ranks, point log on: ids, counts AND rgb
    bytes / weights of the merged memory equal the single-process build bit for bit (the owner of every voxel replays its points
    in global order).  The log costs 16 bytes per ingested point, which is why the every-pixel build does not keep it by default."""
    import torch
    import torch.multiprocessing as mp
    import bsc_nav_amd as B
    mp.spawn(_replay_worker, args=(2, _free_port(), str(tmp_path), rate), nprocs=2, join=True)
    got = np.load(f"{tmp_path}/replay_root.npz")
    rgb, depth, poses, tokens = _inputs()
    one = B.VoxelTokenMemory(_sampled_args(tmp_path, "single", rate), need_diffusion=False, feature_mode="mean", max_frames_per_call=F,
                             voxel_capacity=100_000)
    np.random.seed(3)
    one.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
    pos, c, w = one.engine.export_rgb()
    assert len(pos) > 1000 and np.array_equal(got["pos"], pos)
    assert np.array_equal(got["rgb"], c) and np.array_equal(got["w"], w)           # exact, not "close"
    assert np.array_equal(got["cnt"], one.engine.export_dense()[1])
    # the replay kernel alone against the engine's own chain: all points of the single-process build, grouped by voxel
    one2 = B.VoxelTokenMemory(_sampled_args(tmp_path, "single2", rate), need_diffusion=False, feature_mode="mean", max_frames_per_call=F,
                              voxel_capacity=100_000)
    one2.enable_point_log(F * H * W)
    np.random.seed(3)
    one2.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
    from bsc_nav_amd import dist as bd
    keys = one2.engine.keys_tensor()
    r2, w2 = bd.merge_colour_replay(one2.engine, bd.pack_keys(keys), len(pos))
    assert np.array_equal(r2.cpu().numpy(), c) and np.array_equal(w2.cpu().numpy(), w)


def test_rccl_world1_colour_replay(tmp_path):
    """The all-to-all of the colour replay through RCCL (`all_to_all_single` with split sizes) on a one-rank group: the
    merged rgb / weights equal the chain's own result bit for bit."""
    import torch
    import torch.distributed as dist
    import bsc_nav_amd as B
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        rgb, depth, poses, tokens = _inputs()
        mem = B.VoxelTokenMemory(_sampled_args(tmp_path, "w1r"), need_diffusion=False, feature_mode="mean", max_frames_per_call=F,
                                 voxel_capacity=100_000)
        mem.enable_point_log(F * H * W)
        np.random.seed(3)
        mem.ingest_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), poses, tokens=torch.from_numpy(tokens).cuda())
        before = mem.engine.export_rgb()
        from bsc_nav_amd import dist as bd
        info = bd.merge_dense_maps(mem.engine)
        assert info["colour"].startswith("replay")
        after = mem.engine.export_rgb()
        assert len(before[0]) > 1000 and all(np.array_equal(a, b) for a, b in zip(before, after))
        mem.engine.close()
    finally:
        dist.destroy_process_group()
