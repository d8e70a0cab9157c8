"""Stand-alone rate of the in-tree radix sort (bsc_sort_pairs_u32) on the key shapes bsc_ingest sorts.
usage: sort_bench.py            (prints ms per sort and G items/s; BSC_SORT_ROCPRIM has no effect here: this entry is always in-tree)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
eng = B.VoxelEngine(48, 64, 64, 0.1, -3.2, 3.2, 16, 16, mode="mean", max_points=120_000_000, voxel_capacity=1000)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for name, n, bits, hi in (("iid runs", 118_000_000, 22, 1 << 22), ("room runs", 15_000_000, 15, 30279), ("pairs", 6_000_000, 24, 1 << 24),
                          ("segments", 30_000, 6, 64), ("hot runs", 15_000_000, 15, -1)):
    if hi > 0:
        keys = torch.randint(0, hi, (n,), device="cuda", dtype=torch.int32, generator=g)
    else:
        keys = torch.randint(0, 30279, (n,), device="cuda", dtype=torch.int32, generator=g)
        keys[torch.rand(n, device="cuda", generator=g) < 0.33] = 4242
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    for _ in range(2):
        eng.sort_pairs_u32(keys, vals, 0, bits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        eng.sort_pairs_u32(keys, vals, 0, bits)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:10s} n={n:>11,d} bits={bits:2d}: {ms:8.3f} ms  {n / ms / 1e6:7.2f} G items/s  ({(bits + 7) // 8} passes)")
