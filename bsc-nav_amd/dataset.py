"""create_memory_for_dataset-shaped driver (create_memory_for_dataset.py:54-137) over a frame source.

The reference loops over habitat episodes, derives a memory directory per scene island, skips scenes whose
directory exists (load instead of build) and otherwise explores the scene with the simulator.  The simulator
is out of scope here; the same call order runs over any `FrameSource`:
    initial_memory() -> per frame obs2voxeltoken / ingest_frames -> final flush -> save.
`EnvExplorer` is the frame source for callers that DO hold a simulator: it turns an injected NavEnv-like collaborator
(env.py:49: sims / agent / plnner.pathfinder / move2point) into the stream of (observation, pose) events that
VoxelTokenMemory.excute / exploring_create_memory / explore_entire_space consume (memory_2.py:1086-1145, 1347-1391).
"""
import os

import numpy as np

from . import synthetic
from .memory import VoxelTokenMemory


class EnvExplorer:
    """Event stream over an injected simulator.  Events: ("frame", obs, pose7) after every executed action, ("height", y)
    after a goal is reached (memory_2.py:1122 base_height), ("skipped", exception) for a goal whose move failed (the
    reference swallows those, memory_2.py:1126-1128)."""

    def __init__(self, env):
        self.env = env
        self.last_obs = None

    def pose(self):
        st = self.env.agent.get_state()
        p, q = st.position, st.rotation
        return np.array([p[0], p[1], p[2], q.x, q.y, q.z, q.w], dtype=np.float64)

    def frames(self, actions):
        """step the simulator through `actions` ("stop" entries are no-ops, memory_2.py:1088)"""
        for action in actions:
            if action == "stop":
                continue
            self.last_obs = self.env.sims.step(action)
            yield "frame", self.last_obs, self.pose()

    def random_goal(self):
        """a random navigable point on the agent's own island (memory_2.py:1112-1118)"""
        pf = self.env.plnner.pathfinder
        here = pf.get_island(self.env.agent.get_state().position)
        while True:
            goal = pf.get_random_navigable_point()
            if pf.is_navigable(goal) and pf.get_island(goal) == here:
                return goal

    def sweep(self, turn_deg):
        return ["turn_left"] * int(360 / turn_deg)

    def tour(self, n_goals, turn_deg):
        """n_goals x (walk to a random goal, then look around once)"""
        for _ in range(int(n_goals)):
            try:
                path, _goal = self.env.move2point(self.random_goal())
                yield from self.frames(path)
                yield "height", float(self.env.agent.get_state().position[1])
                yield from self.frames(self.sweep(turn_deg))
            except Exception as e:      # noqa: BLE001 — a failed move costs its goal, not the build
                yield "skipped", e


class SyntheticScene:
    """Seeded stand-in for one scene island: yields batches of (rgb, depth, poses) resident on the device."""

    def __init__(self, name, seed, n_frames, height, width, kind="room", batch=16):
        self.name, self.seed, self.n_frames, self.kind, self.batch = name, seed, n_frames, kind, batch
        self.height, self.width = height, width
        self.poses = synthetic.random_walk_poses(seed, n_frames)

    def __iter__(self):
        for s in range(0, self.n_frames, self.batch):
            p = self.poses[s:s + self.batch]
            rgb, depth, _ = synthetic.make_frames(self.seed * 7919 + s, len(p), self.height, self.width, self.kind,
                                                  poses=p)
            yield rgb, depth, p


def create_memory_for_dataset(args, scenes, encoder, feature_mode="mean", root=None, **memory_kwargs):
    """For every scene: build the memory (or load it when its directory exists, create_memory_for_dataset.py:103).

    Returns {scene name: memory directory}."""
    root = root or args.memory_path
    out = {}
    memory = None
    for scene in scenes:
        memory_path = os.path.join(root, scene.name)                     # :97-99
        args.load_memory_path = memory_path
        if memory is None:
            memory = VoxelTokenMemory(args, memory_path=memory_path, preload_dino=encoder, need_diffusion=False,
                                      feature_mode=feature_mode, max_frames_per_call=scene.batch, **memory_kwargs)
        if os.path.exists(memory_path):                                  # :103-109 load instead of rebuilding
            memory.load_memory(init_state=None)
            out[scene.name] = memory_path
            continue
        memory.memory_save_path = memory_path
        memory.load_memory(build_map=True)                               # fresh caches (memory_2.py:172-184)
        memory.memory_save_path = memory_path
        memory.initial_memory()
        for rgb, depth, poses in scene:
            memory.ingest_frames(rgb, depth, poses)
            memory.base_height.append(float(poses[-1][1]))
        if feature_mode == "exact":
            memory.update_memory_dist_base()                             # memory_2.py:1135 final flush
        memory.save_memory(original_pos=np.asarray(scene.poses[0][:3], dtype=np.float32))
        out[scene.name] = memory.memory_save_path
    return out
