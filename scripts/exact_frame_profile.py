"""cProfile of the frame-by-frame reference-semantics path (obs2voxeltoken, host frames in): where a frame's ~2 ms go."""
import cProfile, pstats, random, sys, tempfile, types, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bsc_nav_amd as B
from bsc_nav_amd import synthetic
H, W, g, D, gs, frames = 480, 640, 16, 1024, 256, 120
poses = synthetic.random_walk_poses(3, frames)
rgb, depth, _ = synthetic.make_frames(3, frames, H, W, "room", poses=poses)
rgb_h, depth_h = rgb.cpu().numpy(), depth.cpu().numpy()
tok = torch.randn(1, g * g, D, device="cuda")
dino = types.SimpleNamespace(forward_features=lambda x: {"x_norm_patchtokens": tok})
args = B.MemoryArgs(width=W, height=H, grid_size=gs, cell_size=0.1, floor_height=-12.8, map_height=12.8, depth_sample_rate=1000,
                    query_width=224, query_height=224, memory_path=tempfile.mkdtemp(), scene_name="p", token_dim=D)
mem = B.VoxelTokenMemory(args, preload_dino=dino, need_diffusion=False, feature_mode="exact", voxel_capacity=400_000, token_capacity=4_000_000)
np.random.seed(0); random.seed(0)
for f in range(10):
    mem.obs2voxeltoken({"rgb": rgb_h[f], "depth": depth_h[f]}, poses[f])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for f in range(10, frames):
    mem.obs2voxeltoken({"rgb": rgb_h[f], "depth": depth_h[f]}, poses[f])
mem.engine.sync(); torch.cuda.synchronize()
pr.disable()
print(f"{(time.perf_counter() - t0) / (frames - 10) * 1e3:.3f} ms per frame under the profiler")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
