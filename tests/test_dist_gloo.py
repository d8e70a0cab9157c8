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
    def __init__(self, mode, D, keys, acc, cnt):
        self.mode, self.D = mode, D
        self.keys, self.acc, self.cnt = keys, acc, cnt

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

    def dense_replace(self, keys, acc, cnt):
        self.keys, self.acc, self.cnt = keys.clone(), acc.clone(), cnt.clone()

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
    eng = DictEngine(mode, D, keys, acc, cnt)
    info = bd.merge_dense_maps(eng)
    q = torch.from_numpy(np.random.RandomState(99).standard_normal((3, D)).astype(np.float32))
    pos, sim = bd.localize_sharded(eng, q, K=20)
    torch.save(dict(info=info, keys=eng.keys, acc=eng.acc, cnt=eng.cnt, pos=pos, sim=sim), f"{out_dir}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["mean", "max"])
def test_two_rank_merge_equals_single_process_reduce(tmp_path, mode):
    world, D = 2, 8
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
