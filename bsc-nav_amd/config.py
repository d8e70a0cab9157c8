"""Configuration of the memory path: every constant the reference takes from args.py or hard-codes.

The reference mutates one argparse namespace at run time (args.py:3-114, create_memory_for_dataset.py:93-99);
`MemoryArgs` is that namespace restricted to the flags the memory path reads, with the reference defaults.
Hard-coded class constants of VoxelTokenMemory (memory_2.py:80-83,102,107-111) are explicit fields here.
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class MemoryArgs:
    # sensor / frame (args.py:24-28,84)
    width: int = 680
    height: int = 680
    sensor_height: float = 1.5
    image_hfov: int = 90
    # query / encoder (args.py:38-39,50,46)
    query_width: int = 224
    query_height: int = 224
    dino_size: str = "dinov2_vitl14_reg"
    imagenary_num: int = 3
    # grid (args.py:54-58)
    floor_height: float = -10.0
    map_height: float = 10.0
    cell_size: float = 0.1
    grid_size: int = 1000
    # axes (args.py:60-63)
    base_forward_axis: List[int] = field(default_factory=lambda: [0, 0, -1])
    base_left_axis: List[int] = field(default_factory=lambda: [-1, 0, 0])
    base_up_axis: List[int] = field(default_factory=lambda: [0, 1, 0])
    base2cam_rot: List[int] = field(default_factory=lambda: [1, 0, 0, 0, -1, 0, 0, 0, -1])
    # depth (args.py:65-67)
    min_depth: float = 0.1
    max_depth: float = 10
    depth_sample_rate: int = 1000
    # paths / flags
    memory_path: str = "./memory"
    scene_name: str = "scene"
    load_memory_path: str = ""
    load_single_floor: bool = False
    random_move_num: int = 30
    turn_left: int = 30
    # VoxelTokenMemory class constants (memory_2.py:80-83,107-111)
    patch_size: int = 14
    token_dim: int = 1024
    iter_size: int = 50000
    cache_size: int = 10


def from_namespace(ns):
    """Accept the reference's argparse namespace (or any object with those attributes)."""
    if isinstance(ns, MemoryArgs):
        return ns
    out = MemoryArgs()
    for k in out.__dataclass_fields__:
        if hasattr(ns, k):
            setattr(out, k, getattr(ns, k))
    return out
