"""Multi-rank merge / sharded-localize logic on CPU: world_size 2, gloo, localhost (no GPU needed).

A dictionary-backed stand-in implements the four engine calls the collective layer uses; the real engine
implements the same calls on the GPU (tests/test_gpu_parity.py covers their kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class DictEngine:
    """keys (n,3) in local id order, feature rows, counts, colour state and a top-down map, all on the CPU."""

    def __init__(self, mode, D, keys, acc, cnt, rgb=None, w=None, gs=12, seed=0):
        import types
        self.mode, self.D = mode, D
        self.keys, self.acc, self.cnt = keys, acc, cnt
        self.device = torch.device("cpu")
        self.cfg = types.SimpleNamespace(token_dim=D)
        rs = np.random.RandomState(77 + seed)
        self.rgb = torch.from_numpy(rs.randint(0, 255, size=(len(keys), 3)).astype(np.uint8)) if rgb is None else rgb
        self.w = torch.from_numpy(rs.uniform(0.1, 4, size=len(keys)).astype(np.float32)) if w is None else w
        self.mh = np.where(rs.uniform(size=(gs, gs)) < 0.5, rs.randint(0, 4, size=(gs, gs)).astype(np.float64), -np.inf)
        self.cv = (rs.randint(1, 255, size=(gs, gs, 3)) * np.isfinite(self.mh)[..., None]).astype(np.uint8)

    def dense_gather_rgb(self, ukeys):
        lut = {tuple(k.tolist()): i for i, k in enumerate(self.keys)}
        rgb = torch.zeros((len(ukeys), 3), dtype=torch.uint8)
        w = torch.zeros(len(ukeys))
        for i, k in enumerate(ukeys.tolist()):
            j = lut.get(tuple(k))
            if j is not None:
                rgb[i], w[i] = self.rgb[j], self.w[j]
        return rgb, w

    def export_heightmap(self):
        return self.mh.copy(), self.cv.copy()

    def import_heightmap(self, mh, cv):
        self.mh, self.cv = mh.copy(), cv.copy()

    def keys_tensor(self):
        return self.keys.clone()

    def dense_gather(self, ukeys):
        lut = {tuple(k.tolist()): i for i, k in enumerate(self.keys)}
        fill = float("-inf") if self.mode == "max" else 0.0
        acc = torch.full((len(ukeys), self.D), fill)
        cnt = torch.zeros(len(ukeys), dtype=torch.int32)
        for i, k in enumerate(ukeys.tolist()):
            j = lut.get(tuple(k))
            if j is not None:
                acc[i], cnt[i] = self.acc[j], self.cnt[j]
        return acc, cnt

    def dense_replace(self, keys, acc, cnt, rgb=None, weight=None):
        self.keys, self.acc, self.cnt = keys.clone(), acc.clone(), cnt.clone()
        self.rgb = torch.zeros((len(keys), 3), dtype=torch.uint8) if rgb is None else rgb.clone()
        self.w = torch.zeros(len(keys)) if weight is None else weight.clone()

    def localize(self, q, K=100, radius=None, curr=None, floor=None):
        rows = self.acc / self.acc.norm(dim=1, keepdim=True).clamp_min(1e-8)
        qn = q / q.norm(dim=1, keepdim=True).clamp_min(1e-8)
        sims = qn @ rows.T
        from bsc_nav_amd.dist import name_key_np
        pos = np.zeros((len(q), K, 3), np.int32)
        sim = np.zeros((len(q), K), np.float32)
        cnt = np.zeros(len(q), np.int32)
        for qi in range(len(q)):
            order = sorted(range(len(self.keys)), key=lambda i: (-float(sims[qi, i]),) + name_key_np(self.keys[i].tolist()))[:K]
            cnt[qi] = len(order)
            pos[qi, :len(order)] = self.keys[order].numpy()
            sim[qi, :len(order)] = sims[qi, order].numpy()
        return pos, sim, cnt


def _make_rank_map(rank, D, mode):
    rs = np.random.RandomState(10 + rank)
    keys = np.unique(rs.randint(0, 12, size=(150, 3)), axis=0).astype(np.int32)     # heavy overlap between ranks
    keys = keys[rs.permutation(len(keys))]                                           # local id order is not sorted
    acc = rs.standard_normal((len(keys), D)).astype(np.float32)
    cnt = rs.randint(1, 9, size=len(keys)).astype(np.int32)
    return torch.from_numpy(keys), torch.from_numpy(acc), torch.from_numpy(cnt)


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bsc_nav_amd import dist as bd
    D = 8
    bd.warmup_collectives(torch.device("cpu"))      # the collectives bench.py warms up before its clock starts
    keys, acc, cnt = _make_rank_map(rank, D, mode)
    eng = DictEngine(mode, D, keys, acc, cnt, seed=rank)
    before = dict(rgb=eng.rgb.clone(), w=eng.w.clone(), mh=eng.mh.copy(), cv=eng.cv.copy())
    info = bd.merge_dense_maps(eng)
    q = torch.from_numpy(np.random.RandomState(99).standard_normal((3, D)).astype(np.float32))
    pos, sim = bd.localize_sharded(eng, q, K=20)
    sl = dict(keys=eng.keys.clone(), acc=eng.acc.clone(), cnt=eng.cnt.clone(), rgb=eng.rgb.clone(), w=eng.w.clone())
    is_root = bd.gather_merged_to_root(eng, info, root=0)
    torch.save(dict(info=info, pos=pos, sim=sim, before=before, mh=eng.mh, cv=eng.cv, is_root=is_root,
                    full=dict(keys=eng.keys, acc=eng.acc, cnt=eng.cnt, rgb=eng.rgb, w=eng.w), **sl), f"{out_dir}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("chunk_bytes", [None, 256])
@pytest.mark.parametrize("mode", ["mean", "max"])
def test_two_rank_merge_equals_single_process_reduce(tmp_path, mode, chunk_bytes, monkeypatch):
    """chunk_bytes 256: the reduce-scatter of the union runs over row chunks of 4 rows per rank (bounded staging instead of
    the whole (U, D) union on every rank) — same result"""
    world, D = 2, 8
    if chunk_bytes:
        monkeypatch.setenv("BSC_MERGE_CHUNK_BYTES", str(chunk_bytes))        # inherited by the spawned ranks
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(f"{tmp_path}/r{r}.pt", weights_only=False) for r in range(world)]
    # single-process reference reduce over both rank maps
    ref = {}
    for r in range(world):
        keys, acc, cnt = _make_rank_map(r, D, mode)
        for k, a, c in zip(keys.tolist(), acc, cnt):
            k = tuple(k)
            if k not in ref:
                ref[k] = [a.clone(), int(c)]
            else:
                ref[k][0] = torch.maximum(ref[k][0], a) if mode == "max" else ref[k][0] + a
                ref[k][1] += int(c)
    got = {}
    for r in range(world):
        for k, a, c in zip(res[r]["keys"].tolist(), res[r]["acc"], res[r]["cnt"]):
            assert tuple(k) not in got, "a voxel must be owned by exactly one rank after the merge"
            got[tuple(k)] = (a, int(c))
    assert set(got) == set(ref) and res[0]["info"]["n_union"] == len(ref)
    for k in ref:
        assert got[k][1] == ref[k][1]                                  # counts exact
        assert torch.allclose(got[k][0], ref[k][0], atol=1e-6)        # sums within fp32 order, max exact
    # ids: global first-touch order = rank 0's voxels in its local order, then rank 1's new voxels in its local order
    k0 = [tuple(k) for k in _make_rank_map(0, D, mode)[0].tolist()]
    k1 = [tuple(k) for k in _make_rank_map(1, D, mode)[0].tolist()]
    seen = set(k0)
    expect_order = k0 + [k for k in k1 if k not in seen]
    got_order = [tuple(k) for r in range(world) for k in res[r]["keys"].tolist()]
    assert got_order == expect_order
    # the root holds the whole memory after the gather, in that order, rows aligned with the slices
    assert res[0]["is_root"] and not res[1]["is_root"]
    full = res[0]["full"]
    assert [tuple(k) for k in full["keys"].tolist()] == expect_order
    for name in ("acc", "cnt", "rgb", "w"):
        assert torch.equal(full[name], torch.cat([res[r][name] for r in range(world)]))
    # colour state: documented rule (merge_colour_states) — rank 0's state, then rank 1's as one observation
    lut = [{k: i for i, k in enumerate(kk)} for kk in (k0, k1)]
    for i, k in enumerate(expect_order):
        st = [(res[r]["before"]["rgb"][lut[r][k]].double(), float(res[r]["before"]["w"][lut[r][k]])) for r in range(world) if k in lut[r]]
        c, w = st[0]
        for c2, w2 in st[1:]:
            num = (c.float() * np.float32(w)).double() + c2 * w2
            c = torch.trunc(num / (w + w2))
            w = float(np.float32(w + w2))
        assert torch.equal(full["rgb"][i].double(), c) and float(full["w"][i]) == np.float32(w)
    # top-down map: highest cell wins, ties go to the later rank; identical on both ranks
    h0, h1 = res[0]["before"]["mh"], res[1]["before"]["mh"]
    exp_h = np.maximum(h0, h1)
    exp_c = np.where((h1 >= h0)[..., None], res[1]["before"]["cv"], res[0]["before"]["cv"])
    for r in range(world):
        assert np.array_equal(res[r]["mh"], exp_h) and np.array_equal(res[r]["cv"], exp_c)
    # sharded localize == localize over the merged single map, identical on both ranks
    keys = torch.tensor(sorted(ref), dtype=torch.int32)
    acc = torch.stack([ref[tuple(k)][0] for k in keys.tolist()])
    single = DictEngine(mode, D, keys, acc, torch.ones(len(keys), dtype=torch.int32))
    q = torch.from_numpy(np.random.RandomState(99).standard_normal((3, D)).astype(np.float32))
    p1, s1, n1 = single.localize(q, K=20)
    for r in range(world):
        for qi in range(3):
            assert np.array_equal(res[r]["pos"][qi], p1[qi, :n1[qi]])
            assert np.allclose(res[r]["sim"][qi], s1[qi, :n1[qi]], atol=1e-6)


def test_key_packing_roundtrip_and_frame_sharding():
    from bsc_nav_amd import dist as bd
    k = torch.tensor([[0, 0, 0], [511, 3, 255], [1000, 999, 199]], dtype=torch.int32)
    assert torch.equal(bd.unpack_keys(bd.pack_keys(k)), k)
    assert bd.shard_frames(10) == (0, 10)
    assert bd.name_key_np([19, 1, 1]) < bd.name_key_np([1, 1, 1])      # "grid_19_" < "grid_1_"


# ---- exact colour across ranks: the exchange of dist.merge_colour_replay on CPU (gloo has no all-to-all: all-gather branch) ----
def _chain_np(recs):
    """The sequential colour chain of memory_2.py:888-899 over (alpha f64, rgb packed) records, restated for the test."""
    c, w, first = np.zeros(3, np.uint32), np.float32(0), True
    for a, rgbv in recs:
        r = np.array([rgbv & 0xff, (rgbv >> 8) & 0xff, (rgbv >> 16) & 0xff], np.uint32)
        if first:
            c, w, first = r.copy(), np.float32(np.float64(np.float32(0)) + a), False
            continue
        den = np.float64(w) + a
        num = (c.astype(np.float32) * w).astype(np.float64) + r.astype(np.float64) * a
        c = np.trunc(num / den).astype(np.uint32)
        w = np.float32(den)
    return c.astype(np.uint8), w


class LogEngine(DictEngine):
    """DictEngine + the point log / replay entry points of the real engine (NumPy stand-ins)."""

    def __init__(self, *a, gs=12, nh=12, log=None, **kw):
        super().__init__(*a, gs=gs, **kw)
        import types
        self.cfg = types.SimpleNamespace(token_dim=self.D, grid_size=gs)
        self.nh, self.log_capacity, self._log = nh, 1 << 20, log

    def point_log(self):
        return self._log

    def replay_colour(self, vox_sorted, records, n_vox):
        rgb, w = torch.zeros((n_vox, 3), dtype=torch.uint8), torch.zeros(n_vox)
        v, rec = vox_sorted.numpy(), records.numpy().astype(np.int64) & 0xffffffff
        assert np.all(np.diff(v) >= 0)
        for vox in np.unique(v):
            rows = rec[v == vox]
            alphas = ((rows[:, 1] << 32) | rows[:, 0]).astype(np.uint64).view(np.float64)
            c, ww = _chain_np(list(zip(alphas, rows[:, 2])))
            rgb[vox], w[vox] = torch.from_numpy(c.copy()), float(ww)
        return rgb, w


def _rank_points(rank, n=400):
    rs = np.random.RandomState(50 + rank)
    vox = rs.randint(0, 5, size=(n, 3))                               # 125 cells, every one hit by both ranks
    alpha = np.exp(-rs.uniform(0.1, 30, size=n) / 1.2)
    rgbv = rs.randint(0, 1 << 24, size=n)
    bad = rs.uniform(size=n) < 0.1                                    # points without a voxel are logged with cell < 0
    return vox, alpha, rgbv, bad


def _replay_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bsc_nav_amd import dist as bd
    gs = nh = 12
    vox, alpha, rgbv, bad = _rank_points(rank)
    cells = np.where(bad, -1, (vox[:, 0] * gs + vox[:, 1]) * nh + vox[:, 2]).astype(np.int32)
    bits = alpha.view(np.uint64)
    recs = np.stack([(bits & 0xffffffff).astype(np.uint32).view(np.int32), (bits >> 32).astype(np.uint32).view(np.int32),
                     rgbv.astype(np.int32)], axis=1)
    ok = ~bad
    _, first = np.unique(vox[ok], axis=0, return_index=True)
    keys = torch.from_numpy(vox[ok][np.sort(first)].astype(np.int32))            # local first-touch order
    eng = LogEngine("mean", 4, keys, torch.zeros((len(keys), 4)), torch.ones(len(keys), dtype=torch.int32), seed=rank,
                    log=(torch.from_numpy(cells), torch.from_numpy(recs)))
    info = bd.merge_dense_maps(eng)
    assert info["colour"].startswith("replay")
    torch.save(dict(keys=eng.keys, rgb=eng.rgb, w=eng.w), f"{out_dir}/c{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_colour_replay_equals_sequential_chain(tmp_path):
    """merge_colour_replay on two CPU ranks: every voxel's colour state after the merge equals the sequential chain over rank
    0's points, then rank 1's, in their logged order (global point order of a frame-sharded build)."""
    world = 2
    mp.spawn(_replay_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seq = {}
    for r in range(world):
        vox, alpha, rgbv, bad = _rank_points(r)
        for v, a, c, b in zip(vox.tolist(), alpha, rgbv, bad):
            if not b:
                seq.setdefault(tuple(v), []).append((a, int(c)))
    n = 0
    for r in range(world):
        got = torch.load(f"{tmp_path}/c{r}.pt", weights_only=False)
        for k, c, w in zip(got["keys"].tolist(), got["rgb"], got["w"]):
            ec, ew = _chain_np(seq[tuple(k)])
            assert np.array_equal(c.numpy(), ec) and np.float32(w.item()) == ew
            n += 1
    assert n == len(seq)


def test_batched_topk_merge_equals_per_query_merge():
    """merge_topk_batched (all queries in one lexsort) == merge_topk query by query: exact ties in name order, ragged counts."""
    from bsc_nav_amd import dist as bd
    rs = np.random.RandomState(3)
    W, Q, K = 4, 9, 40
    pos = rs.randint(0, 25, size=(W, Q, K, 3))
    sim = -np.sort(-np.round(rs.uniform(size=(W, Q, K)), 1).astype(np.float32), axis=2)        # many exact ties
    cnt = rs.randint(0, K + 1, size=(W, Q))
    cnt[:, 0] = 0                                                                               # a query with no candidate anywhere
    got_p, got_s = bd.merge_topk_batched(pos, sim, cnt, K)
    for qi in range(Q):
        p, s = bd.merge_topk([pos[r, qi, :cnt[r, qi]] for r in range(W)], [sim[r, qi, :cnt[r, qi]] for r in range(W)], K)
        assert np.array_equal(got_p[qi], p) and np.array_equal(got_s[qi], s.astype(np.float32))


# ---- the target world size on CPU: eight gloo ranks (until now world size 8 ran only on the GPU box) ----
def test_eight_rank_merge_and_sharded_localize(tmp_path):
    """merge_dense_maps + localize_sharded + gather_merged_to_root over EIGHT ranks: one owner per voxel, counts exact, sums in
    f32 order, ids in global first-touch order (first occurrence in the rank-major concatenation), the K-way merge of the eight
    per-shard top-K lists equal to the top-K of the single merged map on every rank, the root holding the whole memory."""
    world, D, mode = 8, 8, "mean"
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(f"{tmp_path}/r{r}.pt", weights_only=False) for r in range(world)]
    ref, order = {}, []
    for r in range(world):
        keys, acc, cnt = _make_rank_map(r, D, mode)
        for k, a, c in zip(keys.tolist(), acc, cnt):
            k = tuple(k)
            if k not in ref:
                ref[k] = [a.clone(), int(c)]
                order.append(k)
            else:
                ref[k][0] = ref[k][0] + a
                ref[k][1] += int(c)
    got, got_order = {}, []
    for r in range(world):
        for k, a, c in zip(res[r]["keys"].tolist(), res[r]["acc"], res[r]["cnt"]):
            assert tuple(k) not in got
            got[tuple(k)] = (a, int(c))
            got_order.append(tuple(k))
    assert got_order == order and res[0]["info"]["n_union"] == len(ref)
    assert len(ref) % world != 0 or True                # the union need not divide by the world size (sentinel rows in the last slice)
    for k in ref:
        assert got[k][1] == ref[k][1] and torch.allclose(got[k][0], ref[k][0], atol=1e-5)
    info = res[0]["info"]
    assert set(info["phases_ms"]) >= {"global_id_order", "gather_rows+reduce_scatter", "colour", "heightmap", "replace"}
    assert info["reduce_scatter_bytes_sent_per_rank"] == info["per_rank"] * (D * 4 + 4) * (world - 1)
    assert res[0]["is_root"] and not any(res[r]["is_root"] for r in range(1, world))
    assert [tuple(k) for k in res[0]["full"]["keys"].tolist()] == order
    for name in ("acc", "cnt", "rgb", "w"):
        assert torch.equal(res[0]["full"][name], torch.cat([res[r][name] for r in range(world)]))
    keys = torch.tensor(sorted(ref), dtype=torch.int32)
    acc = torch.stack([ref[tuple(k)][0] for k in keys.tolist()])
    single = DictEngine(mode, D, keys, acc, torch.ones(len(keys), dtype=torch.int32))
    q = torch.from_numpy(np.random.RandomState(99).standard_normal((3, D)).astype(np.float32))
    p1, s1, n1 = single.localize(q, K=20)
    for r in range(world):
        for qi in range(3):
            assert np.array_equal(res[r]["pos"][qi], p1[qi, :n1[qi]])
            assert np.allclose(res[r]["sim"][qi], s1[qi, :n1[qi]], atol=1e-5)


def test_eight_rank_colour_replay_equals_sequential_chain(tmp_path):
    """merge_colour_replay over eight CPU ranks: rank 0's points, then rank 1's, ... — the global point order of a frame-sharded
    build — replayed on every voxel's owner equal the sequential chain, bit for bit."""
    world = 8
    mp.spawn(_replay_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seq = {}
    for r in range(world):
        vox, alpha, rgbv, bad = _rank_points(r)
        for v, a, c, b in zip(vox.tolist(), alpha, rgbv, bad):
            if not b:
                seq.setdefault(tuple(v), []).append((a, int(c)))
    n = 0
    for r in range(world):
        got = torch.load(f"{tmp_path}/c{r}.pt", weights_only=False)
        for k, c, w in zip(got["keys"].tolist(), got["rgb"], got["w"]):
            ec, ew = _chain_np(seq[tuple(k)])
            assert np.array_equal(c.numpy(), ec) and np.float32(w.item()) == ew
            n += 1
    assert n == len(seq)
