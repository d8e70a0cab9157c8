"""bsc_import_store from pageable host memory: seconds and GB/s for a store of `rows` x 1024 f32 token rows (M ~ U{1..10} per voxel),
and a read-back check.  usage: import_bench.py [voxels = 2^18] [BSC_H2D_THREADS=n to vary the staging threads]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bsc_nav_amd as B
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
D, gL = 1024, 512
rs = np.random.RandomState(0)
flat = rs.choice(gL * gL * gL // 2, size=V, replace=False).astype(np.int64) + 1
keys = np.stack([flat // (gL * gL), (flat // gL) % gL, flat % gL], axis=1).astype(np.int32)
cnt = rs.randint(1, 11, size=V).astype(np.int32)
T = int(cnt.sum())
rows = np.empty((T, D), np.float32)
for lo in range(0, T, 1 << 18):
    rows[lo:lo + (1 << 18)] = rs.standard_normal((min(1 << 18, T - lo), D)).astype(np.float32)
dists = rs.uniform(0, 9, T).astype(np.float32)
eng = B.VoxelEngine(480, 640, gL, 0.1, -25.6, 25.6, 16, D, mode="exact", iter_size=256, voxel_capacity=V + 8, token_capacity=T, max_points=1024)
for rep in range(2):
    eng.reset()
    t0 = time.perf_counter()
    eng.import_rgb(keys, np.zeros((V, 3), np.uint8), np.ones(V, np.float32))
    t1 = time.perf_counter()
    eng.import_store(keys, cnt, rows, dists)
    t2 = time.perf_counter()
    print(f"rep {rep}: import_rgb {t1 - t0:.3f} s, import_store {t2 - t1:.3f} s for {T * D * 4 / 1e9:.2f} GB = {T * D * 4 / (t2 - t1) / 1e9:.1f} GB/s")
pos, c2, f2, d2 = eng.export_store()
order = np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0])); ko = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
assert np.array_equal(pos[order], keys[ko]) and np.array_equal(c2[order], cnt[ko])
off_in, off_out = np.concatenate([[0], np.cumsum(cnt)]), np.concatenate([[0], np.cumsum(c2)])
for a, b in list(zip(ko, order))[:: max(1, V // 2000)]:
    assert np.array_equal(rows[off_in[a]:off_in[a + 1]], f2[off_out[b]:off_out[b + 1]]) and np.array_equal(dists[off_in[a]:off_in[a + 1]], d2[off_out[b]:off_out[b + 1]])
print("read-back equal")
