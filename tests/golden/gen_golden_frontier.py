#!/usr/bin/env python3
"""Golden vectors for the FrontierExplorer helpers of VoxelTokenMemory (memory_2.py:1147-1311): known / unknown /
frontier cells of the top-down map, 4-connected frontier clusters, centres, information gain and the selected target,
produced by calling the reference's own methods on seeded synthetic top-down maps (build container only).

The simulator only enters through `Env.plnner.pathfinder.is_navigable(loc)`; the stub answers from a seeded mask by
inverting the reference's own grid2loc_2d."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402


def make_map(seed, gs, kind):
    """Top-down colour map (gs,gs,3) u8 (all-zero = unknown) and a navigability mask (gs,gs) bool."""
    rs = np.random.RandomState(seed)
    cv = np.zeros((gs, gs, 3), np.uint8)
    yy, xx = np.mgrid[0:gs, 0:gs]
    known = np.zeros((gs, gs), bool)
    if kind == "rooms":                       # a few explored discs / boxes with unexplored holes and corridors
        for _ in range(6):
            cy, cx, r = rs.randint(gs // 6, gs - gs // 6, 2).tolist() + [rs.randint(gs // 12, gs // 5)]
            known |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        for _ in range(4):
            y0, x0 = rs.randint(0, gs - 10, 2)
            known[y0:y0 + rs.randint(4, gs // 4), x0:x0 + rs.randint(4, gs // 4)] = True
        for _ in range(5):                    # unexplored holes
            cy, cx, r = rs.randint(0, gs, 2).tolist() + [rs.randint(2, gs // 14 + 3)]
            known &= ~((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
    elif kind == "noise":                     # salt-and-pepper: many tiny clusters, most below min_cluster_size
        known = rs.uniform(size=(gs, gs)) < 0.55
    elif kind == "full":                      # everything known: no frontier at all
        known[:] = True
    elif kind == "border":                    # known region touching the map border (in_bounds checks)
        known[:gs // 3, :] = True
        known[:, -gs // 4:] = True
    cv[known] = rs.randint(1, 256, size=(int(known.sum()), 3)).astype(np.uint8)
    if kind == "rooms":                       # known cells whose colour is black still count as unknown (:1165)
        ky, kx = np.nonzero(known)
        sel = rs.choice(len(ky), size=len(ky) // 50, replace=False)
        cv[ky[sel], kx[sel]] = 0
    nav = rs.uniform(size=(gs, gs)) < (0.9 if kind != "noise" else 0.7)
    return cv, nav


def main():
    ref_utils, ref_mem = import_reference("/root/reference")
    data = {}
    cases = [("f1", 1, 96, "rooms", 10, 5), ("f2", 2, 160, "rooms", 10, 5), ("f3", 3, 64, "noise", 3, 2),
             ("f4", 4, 48, "full", 10, 5), ("f5", 5, 80, "border", 5, 7), ("f6", 6, 256, "rooms", 10, 5),
             ("f7", 7, 40, "noise", 1, 0)]
    for name, seed, gs, kind, min_size, radius in cases:
        cv, nav = make_map(seed, gs, kind)
        M = object.__new__(ref_mem.VoxelTokenMemory)
        M.gs, M.cs = gs, 0.1
        M.cv_map = cv
        M.min_cluster_size, M.ig_radius = min_size, radius
        origin = np.array([1.5, 0.25, -2.0])
        M.Env = types.SimpleNamespace(original_state=types.SimpleNamespace(position=origin))

        def is_navigable(loc, gs=gs, nav=nav, origin=origin, cs=0.1):
            col = int(round((loc[0] - origin[0]) / cs)) + gs // 2
            row = int(round((loc[2] - origin[2]) / cs)) + gs // 2
            return bool(nav[row, col])
        M.Env.plnner = types.SimpleNamespace(pathfinder=types.SimpleNamespace(is_navigable=is_navigable))
        mask = M.build_navigable_mask()
        frontiers = M.find_frontiers(mask)
        clusters = M.cluster_frontiers(frontiers)
        centers = [M.compute_cluster_center(c) for c in clusters]
        gains = [M.compute_information_gain(cx, cy) for cx, cy in centers]
        best = M.select_best_cluster_center_by_ig(clusters)
        labels = -np.ones((gs, gs), np.int32)
        for k, c in enumerate(clusters):
            for (x, y) in c:
                labels[x, y] = k
        data[f"{name}_cv_map"], data[f"{name}_nav"] = cv, nav
        data[f"{name}_params"] = np.array([gs, min_size, radius], np.int64)
        data[f"{name}_navigable_mask"] = mask
        data[f"{name}_frontiers"] = np.asarray(frontiers, np.int64).reshape(-1, 2)
        data[f"{name}_labels"] = labels
        data[f"{name}_sizes"] = np.asarray([len(c) for c in clusters], np.int64)
        data[f"{name}_first"] = np.asarray([c[0] for c in clusters], np.int64).reshape(-1, 2)
        data[f"{name}_centers"] = np.asarray(centers, np.float64).reshape(-1, 2)
        data[f"{name}_gains"] = np.asarray(gains, np.float64)
        data[f"{name}_best"] = np.asarray(best if best is not None else [np.nan, np.nan], np.float64)
        data[f"{name}_loc"] = np.stack([M.grid2loc_2d(3, 7), M.grid2loc_2d(gs - 1, 0)])
        data[f"{name}_l2g"] = np.asarray([M.loc2grid_2d(0.37, -1.21), M.loc2grid_2d(-2.05, 0.0)], np.int64)
        print(name, kind, "gs", gs, "known", int((cv.sum(-1) != 0).sum()), "frontier cells", len(frontiers), "clusters",
              len(clusters), "gains", gains[:6], "best", best)
    np.savez_compressed(os.path.join(HERE, "g7_frontier.npz"), **data)


if __name__ == "__main__":
    main()
