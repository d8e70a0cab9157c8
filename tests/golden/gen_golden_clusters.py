#!/usr/bin/env python3
"""Golden vectors for the step right after localize: GESObjectNavRobot.weighted_cluster_centers
(BSCAgent.py:479-497), produced by calling the reference's own method (build container only)."""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def import_agent(ref):
    def stub(name):
        m = MagicMock(name=name)
        m.__spec__ = types.SimpleNamespace(name=name, loader=None, origin=None, submodule_search_locations=[])
        m.__path__ = []
        return m
    for name in ["cv2", "habitat_sim", "habitat_sim.utils", "habitat_sim.utils.common", "kneed", "open3d", "diffusers",
                 "ultralytics", "torchvision", "torchvision.transforms", "magnum", "habitat", "habitat.utils",
                 "habitat.utils.visualizations", "habitat.utils.visualizations.maps", "transformers", "matplotlib",
                 "matplotlib.pyplot", "matplotlib.colors", "h5py", "open_clip", "openai", "imageio", "memory_2"]:
        sys.modules[name] = stub(name)
    for name in ["env", "LLMAgent", "vlnce_maps"]:       # `from x import *` needs real (empty) modules
        sys.modules[name] = types.ModuleType(name)
    sys.modules["vlnce_maps"].colorize_draw_agent_and_fit_to_height_vlnce = None
    sys.path.insert(0, ref)
    import BSCAgent
    return BSCAgent


def cases():
    rs = np.random.RandomState(0)
    out = []
    # three blobs of different sizes + scattered noise, like a top-100 of voxel positions
    for seed, K, blobs in [(1, 100, [(40, 4.0), (30, 3.0), (12, 2.0)]), (2, 100, [(60, 8.0), (25, 2.5)]),
                           (3, 64, [(20, 1.5), (20, 1.5), (4, 1.0)]), (4, 100, []), (5, 10, [(6, 1.0)]),
                           (6, 100, [(50, 5.0), (45, 5.0)])]:
        rs = np.random.RandomState(seed)
        pts = []
        for n, sd in blobs:
            c = rs.randint(60, 440, size=3)
            pts.append(np.round(c + rs.standard_normal((n, 3)) * sd))
        n_noise = K - sum(n for n, _ in blobs)
        pts.append(rs.randint(0, 500, size=(n_noise, 3)))
        pos = np.concatenate(pts).astype(np.int64)[:K]
        if seed == 6:                   # two touching blobs: border points reachable from both clusters
            pos[50:95] = pos[:45] + np.array([9, 0, 0])
        pos = pos[rs.permutation(len(pos))]
        sim = np.sort(rs.uniform(0.2, 0.9, size=len(pos)))[::-1].astype(np.float32).astype(np.float64)
        out.append((f"c{seed}", pos, sim))
    # 40 points where a border point is within eps of core points of TWO clusters (found by search, seed 290):
    # sklearn hands it to the cluster that is expanded first
    rs = np.random.RandomState(290)
    pos = rs.randint(0, 60, size=(40, 3)).astype(np.int64)
    sim = np.sort(rs.uniform(0.2, 0.9, size=40))[::-1].astype(np.float32).astype(np.float64)
    out.append(("c7", pos, sim))
    return out


def main():
    B = import_agent("/root/reference")
    fn = B.GESObjectNavRobot.weighted_cluster_centers
    data = {}
    for name, pos, sim in cases():
        centers, labels, sizes = fn(None, pos, sim)
        data[f"{name}_pos"], data[f"{name}_sim"] = pos, sim
        data[f"{name}_centers"] = np.asarray(centers, dtype=np.float64).reshape(-1, 3)
        data[f"{name}_labels"] = np.asarray(labels, dtype=np.int64)
        data[f"{name}_sizes"] = np.asarray(sizes, dtype=np.int64)
        print(name, "clusters", len(sizes), "sizes", sizes, "noise", int((labels == -1).sum()))
    np.savez_compressed(os.path.join(HERE, "g5_cluster_centers.npz"), **data)


if __name__ == "__main__":
    main()
