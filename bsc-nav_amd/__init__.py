"""bsc_nav_amd — MI355X-native structured spatial memory for BSC-Nav (construction + query path).

The directory is named ``bsc-nav_amd`` (repository convention); import it as ``bsc_nav_amd``
through the shim package of that name at the repository root.
"""
from . import _lib
from .engine import VoxelEngine
from .geometry import PoseChain, cam_mat_fov, cam_mat_patch, pose_vec2tf, sample_indices, sample_indices_fast
from .config import MemoryArgs
from .memory import Memory, VoxelTokenMemory
from .dataset import create_memory_for_dataset

__all__ = ["VoxelEngine", "VoxelTokenMemory", "Memory", "create_memory_for_dataset", "MemoryArgs", "PoseChain", "cam_mat_fov", "cam_mat_patch", "pose_vec2tf", "sample_indices", "sample_indices_fast", "_lib"]
