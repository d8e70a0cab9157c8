"""Reference-faithful NumPy restatement of obs2voxeltoken's per-point loop — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/ and bench.py's ``cpu_baseline`` leg may import this module.  It restates, with the same NumPy calls per point
that the reference makes (small matmuls, Python ints and floats), what memory_2.py:842-903 does for one frame: the
vectorised unprojection and transform (utils.py:153-199) followed by the Python loop over the sampled points (voxel
index, two projections, token append, rgb running mean, top-down map).  It exists to time the reference's own style of
execution on the GPU box's host (SURVEY.md 8d "reference-faithful NumPy path"), since the reference files never travel there;
tests/test_oracle_golden.py pins it to the goldens produced by the reference itself.
"""
import numpy as np


class NumpyLoopMemory:
    def __init__(self, H, W, gs, cs, floor_height, map_height, g, D, iter_size=50000, min_depth=0.1, max_depth=10.0):
        self.H, self.W, self.gs, self.cs, self.g, self.D = H, W, gs, cs, g, D
        self.minh, self.maxh = int(floor_height / cs), int(map_height / cs)              # memory_2.py:122-123
        f = W / (2.0 * np.tan(np.deg2rad(90) / 2.0))                                     # utils.py:181-186 (both from W)
        self.calib = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
        self.kpatch = np.array([[g / 2.0, 0, g / 2.0], [0, g / 2.0, g / 2.0], [0, 0, 1.0]])   # utils.py:144-150
        self.min_depth, self.max_depth = min_depth, max_depth
        self.iter_size = iter_size
        self.grid_feat = np.zeros((iter_size, D), np.float32)                            # memory_2.py:708-722
        self.grid_feat_pos = np.zeros((iter_size, 3), np.int32)
        self.grid_feat_dis = np.zeros(iter_size, np.float32)
        self.iter_id = 0
        self.grid_rgb_pos = np.zeros((gs * gs, 3), np.int32)
        self.grid_rgb = np.zeros((gs * gs, 3), np.uint8)
        self.weight = np.zeros(gs * gs, np.float32)
        self.occupied_ids = -np.ones((gs, gs, self.maxh - self.minh), np.int32)
        self.max_id = 0
        self.cv_map = np.zeros((gs, gs, 3), np.uint8)                                    # memory_2.py:98-100
        self.max_height = np.full((gs, gs), -np.inf)
        self.dropped = 0

    def _depth2pc(self, depth):                                                          # utils.py:153-178
        H, W = depth.shape
        x, y = np.meshgrid(np.arange(W), np.arange(H))
        p2d = np.vstack([x.reshape(1, -1) + 0.5, y.reshape(1, -1) + 0.5, np.ones((1, H * W))])
        z = depth.reshape(1, -1)
        pc = (np.linalg.inv(self.calib) @ p2d) * z
        mask = (pc[2] > self.min_depth) & (pc[2] < self.max_depth)
        return pc, mask

    def ingest_frame(self, depth, rgb, idx, T, tokens, max_points=None):
        """depth (H,W) f32, rgb (H,W,>=3) u8, idx: sampled pixel indices in the reference's shuffled order (None: all, in
        order), T: pc_transform (4,4), tokens (g,g,D) f32.  Returns the number of points walked (max_points caps it: timing)."""
        pc, mask = self._depth2pc(depth)
        if idx is None:
            idx = np.arange(depth.size)
        idx = idx[mask[idx]]                                                             # memory_2.py:750-752
        pc = pc[:, idx]
        pcg = (T @ np.vstack([pc, np.ones((1, pc.shape[1]))]))[:3]                       # utils.py:189-199
        n = pc.shape[1] if max_points is None else min(pc.shape[1], max_points)
        gs, cs = self.gs, self.cs
        for i in range(n):                                                               # memory_2.py:863-903
            p, pg = pc[:, i], pcg[:, i]
            row = int(gs / 2 - int(pg[0] / cs)); col = int(gs / 2 - int(pg[1] / cs)); h = int(pg[2] / cs)   # utils.py:201-205
            if col >= gs or row >= gs or h >= self.maxh or col < 0 or row < 0 or h < self.minh:             # :755-756
                continue
            h -= self.minh
            q = self.calib @ p; q = q / q[2]
            px, py = int(q[0] - 0.5), int(q[1] - 0.5)                                    # utils.py:208-214
            rgb_v = rgb[py, px, :3]                                                      # negative indices wrap
            q = self.kpatch @ p; q = q / q[2]
            tx, ty = int(q[0] - 0.5), int(q[1] - 0.5)
            r2 = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]
            alpha = np.exp(-r2 / (2 * 0.6))
            if tx < 0 or ty < 0 or tx >= self.g or ty >= self.g:                         # :878
                continue
            if self.iter_id >= self.iter_size:                                           # :880-881 (flush, token dropped)
                self.iter_id = 0
                self.dropped += 1
            else:
                self.grid_feat[self.iter_id] = tokens[ty, tx]
                self.grid_feat_pos[self.iter_id] = [row, col, h]
                self.grid_feat_dis[self.iter_id] = r2
                self.iter_id += 1
            vid = self.occupied_ids[row, col, h]
            if vid == -1:                                                                # :888-894
                vid = self.max_id
                self.occupied_ids[row, col, h] = vid
                self.grid_rgb_pos[vid] = [row, col, h]
                self.grid_rgb[vid] = rgb_v
                self.weight[vid] = self.weight[vid] + alpha
                self.max_id += 1
            else:                                                                        # :895-899
                self.grid_rgb[vid] = (self.grid_rgb[vid] * self.weight[vid] + rgb_v * alpha) / (self.weight[vid] + alpha)
                self.weight[vid] += alpha
            if h >= self.max_height[row, col]:                                           # :901-903
                self.max_height[row, col] = h
                self.cv_map[row, col] = rgb_v
        return n
