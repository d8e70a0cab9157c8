"""Host-side geometry of the memory path: intrinsics, pose chain, point sub-sampling.

These few 3x3 / 4x4 operations per frame stay on the host in NumPy / SciPy, exactly as the
reference evaluates them, so that the matrices handed to the kernels are bit-identical to the
reference's:
  cam_mat_fov      utils.py:181-186   cam_mat_patch  utils.py:144-150
  pose_vec2tf      utils.py:133-141   PoseChain      memory_2.py:844-851,860
  sample_indices   memory_2.py:747-749 (global NumPy RNG, Fisher-Yates over all N pixels)
"""
import numpy as np
from scipy.spatial.transform import Rotation as R


def cam_mat_fov(h, w, fov=90):
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / (2.0 * np.tan(np.deg2rad(fov / 2)))   # both focal lengths from the WIDTH
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


def cam_mat_patch(h, w):
    m = np.eye(3)
    m[0, 0] = m[1, 1] = w / 2.0
    m[0, 2] = w / 2.0
    m[1, 2] = h / 2.0
    return m


def pose_vec2tf(pose):
    """(px, py, pz, qx, qy, qz, qw) -> 4x4."""
    pose = np.asarray(pose, dtype=np.float64)
    tf = np.eye(4)
    tf[:3, 3] = pose[:3].flatten()
    tf[:3, :3] = R.from_quat(pose[3:].flatten()).as_matrix()
    return tf


class PoseChain:
    """pc_transform = inv(B T0 B^-1) (B T B^-1) B base2cam; the map frame is the first ingested pose."""

    def __init__(self, base_forward_axis=(0, 0, -1), base_left_axis=(-1, 0, 0), base_up_axis=(0, 1, 0),
                 base2cam_rot=(1, 0, 0, 0, -1, 0, 0, 0, -1), sensor_height=1.5):
        self.base_transform = np.eye(4)
        self.base_transform[0, :3] = base_forward_axis
        self.base_transform[1, :3] = base_left_axis
        self.base_transform[2, :3] = base_up_axis
        self.base2cam_tf = np.eye(4)
        self.base2cam_tf[:3, :3] = np.array([base2cam_rot]).reshape((3, 3))
        self.base2cam_tf[1, 3] = sensor_height
        self.inv_init_base_tf = None
        self.init_base_tf = None
        self.tf = None

    def reset(self):
        self.inv_init_base_tf = None

    def anchor(self, pose):
        """Fix the map origin at `pose` instead of at the first ingested frame (memory_2.py:844-847).  A frame-sharded
        build calls this on every rank with the scene's first pose, so that all ranks write into one map frame."""
        B = self.base_transform
        self.init_base_tf = B @ pose_vec2tf(np.asarray(pose, dtype=np.float64)) @ np.linalg.inv(B)
        self.inv_init_base_tf = np.linalg.inv(self.init_base_tf)

    def pc_transform(self, pose):
        B = self.base_transform
        if self.inv_init_base_tf is None:
            self.init_base_tf = B @ pose_vec2tf(pose) @ np.linalg.inv(B)
            self.inv_init_base_tf = np.linalg.inv(self.init_base_tf)
        base_pose = B @ pose_vec2tf(pose) @ np.linalg.inv(B)
        self.tf = self.inv_init_base_tf @ base_pose
        return np.ascontiguousarray(self.tf @ B @ self.base2cam_tf)


def sample_indices(n_pixels, rate):
    """The reference's shuffled sub-sampling (memory_2.py:747-749); consumes np.random's global stream identically."""
    idx = np.arange(n_pixels)
    np.random.shuffle(idx)
    return np.ascontiguousarray(idx[::rate].astype(np.int32))


_scratch = {}


def sample_indices_fast(n_pixels, rate):
    """Same permutation, same final state of np.random's global MT19937 stream, from the library's own restatement of
    NumPy's legacy shuffle (bsc_host_shuffled_sample: ~2.5x faster than np.random.shuffle at 640x480)."""
    import ctypes as C
    from . import _lib
    st = np.random.get_state()
    if st[0] != "MT19937":
        return sample_indices(n_pixels, rate)
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    scratch = _scratch.get(n_pixels)
    if scratch is None:
        scratch = _scratch[n_pixels] = np.empty(n_pixels, np.int32)
    out = np.empty((n_pixels + rate - 1) // rate, np.int32)
    _lib.check(_lib.load().bsc_host_shuffled_sample(key.ctypes.data_as(C.c_void_p), C.byref(pos), n_pixels, int(rate),
                                                    scratch.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))
    return out


class SamplePrefetcher:
    """The reference's per-frame shuffled sub-sampling (memory_2.py:747-749), drawn AHEAD of its use by a host thread.

    The permutation of frame f + 1 does not depend on frame f's data — only on the position of NumPy's global MT19937 stream —
    so a worker thread draws the index arrays one after the other from a private copy of the stream (bsc_host_shuffled_sample
    releases the GIL) while the GPU works on the current frame.  `next()` hands them out in order; `close()` puts the global
    stream where the reference would have left it after the frames actually consumed.  Valid while nothing else draws from
    np.random between the frames (the dataset loop, ingest_frames)."""

    def __init__(self, n_pixels, rate, depth=4):
        import ctypes as C
        import queue
        import threading
        from . import _lib
        st = np.random.get_state()
        self._fallback = st[0] != "MT19937"
        self.n_pixels, self.rate = int(n_pixels), int(rate)
        self._tail = st
        if self._fallback:
            return
        self._lib, self._C = _lib.load(), C
        self._check = _lib.check
        self._q = queue.Queue(maxsize=max(1, depth))
        self._stop = False
        self._key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        self._pos = C.c_int32(int(st[2]))
        self._rest = (st[3], st[4])
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _work(self):
        C = self._C
        scratch = np.empty(self.n_pixels, np.int32)
        while not self._stop:
            out = np.empty((self.n_pixels + self.rate - 1) // self.rate, np.int32)
            try:
                self._check(self._lib.bsc_host_shuffled_sample(self._key.ctypes.data_as(C.c_void_p), C.byref(self._pos), self.n_pixels,
                                                               self.rate, scratch.ctypes.data_as(C.c_void_p),
                                                               out.ctypes.data_as(C.c_void_p)))
                item = (out, ("MT19937", self._key.copy(), self._pos.value) + self._rest)
            except BaseException as e:      # noqa: BLE001 — handed to the consumer: next() re-raises it instead of waiting forever
                item = (e, None)
                self._stop = True
                self._q.put(item)
                return
            while not self._stop:
                try:
                    self._q.put(item, timeout=0.05)
                    break
                except Exception:
                    pass

    def next(self):
        if self._fallback:
            return sample_indices(self.n_pixels, self.rate)
        import queue
        while True:
            try:
                idx, state = self._q.get(timeout=1.0)
                break
            except queue.Empty:
                if not self._thread.is_alive():
                    raise RuntimeError("SamplePrefetcher: the worker thread is gone") from None
        if isinstance(idx, BaseException):
            raise idx
        self._tail = state
        return idx

    def close(self):
        if not self._fallback:
            self._stop = True
            while self._thread.is_alive():          # unblock a producer waiting on the full queue
                try:
                    self._q.get_nowait()
                except Exception:
                    pass
                self._thread.join(timeout=0.01)
            np.random.set_state(self._tail)
