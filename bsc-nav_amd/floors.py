"""Single-floor selection of a loaded memory (`--load_single_floor`, memory_2.py:202-252) — host logic.

The reference clusters the agent heights recorded while the memory was built (`base_height.npy`) with
DBSCAN(eps=0.4, min_samples=max(1, n // 5)); every cluster is a floor at the mean height of its samples.
The voxel z-extent [zmin, zmax] of the map is then cut at the floor-to-floor height differences measured from the
lowest floor: floor i spans (zmin + (f_i - f_0)/cs, zmin + (f_{i+1} - f_0)/cs) with the map's own extent at both
ends, and every span is shrunk by one cell per side through `int(lo) + 1`, `int(hi) - 1`.  A single floor keeps
the full extent unchanged.  Pinned by tests/golden/g8_floor_split.npz (outputs of the reference's own method).
"""
import numpy as np


def floor_levels(base_height):
    """Sorted mean heights of the DBSCAN clusters of the recorded agent heights (memory_2.py:205-217)."""
    from sklearn.cluster import DBSCAN
    h = np.asarray(base_height, dtype=np.float64).reshape(-1, 1)
    labels = DBSCAN(eps=0.4, min_samples=max(1, len(h) // 5)).fit(h).labels_
    return sorted(float(np.mean(h[labels == k])) for k in set(labels.tolist()) if k != -1)


def floor_ranges(levels, z_extent, cell_size):
    """Per-floor inclusive voxel z-ranges (memory_2.py:224-241)."""
    zmin, zmax = z_extent
    if len(levels) == 1:
        return [[zmin, zmax]]
    cuts = [zmin + (f - levels[0]) / cell_size for f in levels]     # cuts[0] == zmin
    out = []
    for i in range(len(levels)):
        lo = zmin if i == 0 else cuts[i]
        hi = zmax if i == len(levels) - 1 else cuts[i + 1]
        out.append([int(lo) + 1, int(hi) - 1])
    return out


def select_floor(base_height, grid_rgb_pos, cell_size, current_height):
    """-> dict(levels, num_floors, current_floor, zrange [lo, hi], mask over grid_rgb_pos rows)."""
    levels = floor_levels(base_height)
    current = int(np.argmin(np.abs(np.array(levels) - current_height)))      # ValueError on no floor, as the reference
    z = grid_rgb_pos[:, 2]
    lo, hi = floor_ranges(levels, [z.min(), z.max()], cell_size)[current]
    lo, hi = int(lo), int(hi)
    return dict(levels=levels, num_floors=len(levels), current_floor=current, zrange=[lo, hi],
                mask=np.logical_and(z >= lo, z <= hi))
