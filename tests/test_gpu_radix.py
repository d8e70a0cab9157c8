"""The in-tree radix sort (csrc/radix.hip) behind bsc_ingest's two sorts — the run sort that gives every voxel its points in the
reference's order (memory_2.py:888-903) and the pair sort of the dense reduce — against NumPy's stable sort: bit-exact keys and
values, i.e. the same permutation, for every size class (empty, one item, around a tile of 8192, several tiles, odd tails), every
digit plan the library uses (6-bit segment classes, 15-22-bit voxel ids, 24-bit Morton cells, an offset bit range, all 32 bits)
and the key distributions that stress it (all equal, two values, already sorted, reversed, one hot key among random ones)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import bsc_nav_amd as B
    e = B.VoxelEngine(48, 64, 64, 0.1, -3.2, 3.2, 16, 16, mode="mean", max_points=3_100_000, voxel_capacity=1000)
    yield e
    e.close()


def _check(eng, keys, vals, b0, b1):
    import torch
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    ko, vo = eng.sort_pairs_u32(k, v, b0, b1)
    torch.cuda.synchronize()
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys) and np.array_equal(v.cpu().numpy().view(np.uint32), vals), "input changed"
    digits = (keys >> np.uint32(b0)) & np.uint32((1 << (b1 - b0)) - 1 if b1 - b0 < 32 else 0xffffffff)
    order = np.argsort(digits, kind="stable")
    assert np.array_equal(ko.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(vo.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("n", [0, 1, 5, 64, 1000, 8191, 8192, 8193, 16384, 100_003, 3_000_001])
@pytest.mark.parametrize("bits", [(0, 6), (0, 15), (0, 19), (0, 24), (3, 32), (0, 32)])
def test_random_keys_every_size_and_digit_plan(eng, n, bits):
    rng = np.random.default_rng(n * 37 + bits[1])
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    _check(eng, keys, np.arange(n, dtype=np.uint32), *bits)


@pytest.mark.parametrize("kind", ["equal", "two", "sorted", "reversed", "hot", "low_bits_only"])
def test_key_distributions(eng, kind):
    n = 1_000_003
    rng = np.random.default_rng(5)
    if kind == "equal":
        keys = np.full(n, 0x00abcdef, np.uint32)
    elif kind == "two":
        keys = rng.integers(0, 2, n).astype(np.uint32) * np.uint32(0x00ff00ff)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 22, n).astype(np.uint32))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 22, n).astype(np.uint32))[::-1].copy()
    elif kind == "hot":         # the voxel in front of the camera: a third of all runs carry one id
        keys = rng.integers(0, 1 << 15, n).astype(np.uint32)
        keys[rng.random(n) < 0.33] = 4242
    else:                       # key bits above the sorted range differ (run keys carry the run length there)
        keys = (rng.integers(0, 1 << 10, n).astype(np.uint32) << np.uint32(22)) | rng.integers(0, 1 << 15, n).astype(np.uint32)
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    _check(eng, keys, vals, 0, 24 if kind != "low_bits_only" else 15)


def test_back_to_back_sorts_share_the_workspace(eng):
    """Status words are tagged with an epoch instead of being cleared: sorts of different sizes and plans right after each other."""
    rng = np.random.default_rng(11)
    for n, bits in ((500_000, (0, 24)), (9000, (0, 6)), (2_000_000, (0, 17)), (8192, (0, 32)), (70_000, (0, 8))):
        keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        _check(eng, keys, np.arange(n, dtype=np.uint32), *bits)
