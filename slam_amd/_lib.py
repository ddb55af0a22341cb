"""ctypes binding of libslam_amd.so (C ABI: include/slam_amd.h).

There is no fallback: if the shared library is missing or no gfx950 GPU is usable, the first call raises.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libslam_amd.so")

REC_COUNT, REC_ERRSUM, REC_INDEX, REC_F, REC_LEN = 0, 1, 2, 3, 12
SAMPLER_CPYTHON, SAMPLER_COUNTER = 0, 1
SEED_BASE = 0x1337BEEF
DEFAULT_THRESHOLD = 0.005
NOALLOC_MAX_H = 16384
SCORE_ALGEBRAIC, SCORE_SAMPSON = 0, 1
K_SOLVE, K_SCORE, K_SELECT, K_TRIANGULATE, K_RESECT, K_TRACK_SCORE, K_MATCH = 0, 1, 2, 3, 4, 5, 6

# every symbol include/slam_amd.h declares, with its ctypes signature
_vp, _i32, _i64, _u64, _f64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_int
SIGNATURES = {
    "slam_create": (_int, [_int, C.POINTER(_vp)]),
    "slam_destroy": (None, [_vp]),
    "slam_last_error": (C.c_char_p, [_vp]),
    "slam_build_hash": (C.c_char_p, []),
    "slam_set_stream": (_int, [_vp, _vp]),
    "slam_set_concurrency": (_int, [_vp, _int]),
    "slam_set_scoring": (_int, [_vp, _int]),
    "slam_synchronize": (_int, [_vp]),
    "slam_host_alloc": (_int, [_vp, _u64, C.POINTER(C.c_void_p)]),
    "slam_host_free": (_int, [_vp, _vp]),
    "slam_num_hypotheses": (_i64, [_i64, _i64]),
    "slam_sample_indices": (_int, [_vp, _int, _u64, _i64, _int, _int, _int, _vp]),
    "slam_eightpoint": (_int, [_vp, _vp, _vp, _int, _vp]),
    "slam_ransac_f": (_int, [_vp, _vp, _vp, _int, _vp, _int, _u64, _i64, _int, _f64, _vp, _vp, _vp, _vp, _vp]),
    "slam_ransac_f_dev": (_int, [_vp, _vp, _vp, _int, _vp, _int, _u64, _i64, _int, _f64, _vp, _vp, _vp, _vp, _vp]),
    "slam_ransac_f_dev_noalloc": (_int, [_vp, _vp, _vp, _int, _vp, _int, _u64, _i64, _int, _f64, _vp, _vp]),
    "slam_ransac_job_create": (_int, [_vp, _vp, _vp, _int, _vp, _int, _u64, _i64, _int, _f64, _vp, _vp, _vp, _vp, _vp,
                                      C.POINTER(_vp)]),
    "slam_job_run": (_int, [_vp]),
    "slam_job_destroy": (None, [_vp]),
    "slam_ransac_f_adaptive": (_int, [_vp, _vp, _vp, _int, _int, _u64, _i64, _int, _f64, _f64, _vp, _vp, _vp]),
    "slam_merge_records": (_int, [_vp, _int, _vp]),
    "slam_comm_unique_id": (_int, [_vp, _vp]),
    "slam_comm_init_rank": (_int, [_vp, _vp, _int, _int]),
    "slam_comm_count": (_int, [_vp, _vp, C.POINTER(_int)]),
    "slam_exchange_winner_dev": (_int, [_vp, _vp, _vp]),
    "slam_allreduce_counts4_dev": (_int, [_vp, _vp, _vp]),
    "slam_merge_records_dev": (_int, [_vp, _vp, _int, _vp]),
    "slam_normalize_points": (_int, [_vp, _vp, _int, _vp, _vp]),
    "slam_fundamental_to_essential": (_int, [_vp, _vp, _vp, _vp]),
    "slam_triangulate": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "slam_triangulate_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "slam_pose_from_E": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "slam_pose_candidates_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "slam_pose_select_dev": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "slam_init_pair": (_int, [_vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slam_init_pairs": (_int, [_vp, _vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slam_num_tracking_hypotheses": (_i64, [_i64, _i64]),
    "slam_pose_from_points": (_int, [_vp, _vp, _vp, _int, _vp, _vp]),
    "slam_track_pose": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slam_track_pose_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "slam_match_knn": (_int, [_vp, _vp, _int, _vp, _int, _int, _vp, _vp]),
    "slam_match_points": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "slam_match_points_dev": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "slam_frame_pair": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp]),
    "slam_profile_enable": (_int, [_vp, _int]),
    "slam_profile_read": (_int, [_vp, _int, C.POINTER(_f64), C.POINTER(_i64)]),
    "slam_profile_reset": (_int, [_vp]),
}


SOURCE_FILES = ("slam_amd.hip", "ctx.h", "exchange.h", "linalg.h", "kernels_ransac.h", "kernels_count32.h", "kernels_tri.h",
                "kernels_track.h", "kernels_match.h", "../../include/slam_amd.h")   # the Makefile's $(SRC) $(HDR), in its order


def source_hash():
    """What slam_build_hash() of a library built from this tree's sources returns (None without the sources)."""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in SOURCE_FILES:
            with open(os.path.join(_HERE, "csrc", name), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def build_hash():
    """slam_build_hash() of the library that is loaded."""
    return load().slam_build_hash().decode()


class SlamError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own copy of the HIP and HSA runtimes.  Two copies in one process do not both
    get the GPU: the one that initialises second sees no device.  When torch is installed but not imported yet, load
    its copy first, so that libslam_amd binds to it and a later `import torch` finds it already in the process
    (the order torch-then-libslam_amd needs nothing: the library then binds to torch's copy by itself).
    SLAM_AMD_SYSTEM_HIP=1 keeps the system runtime."""
    if "torch" in sys.modules or os.environ.get("SLAM_AMD_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the library and declare the signatures (does not touch the GPU)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SlamError(f"{LIB_PATH} is missing: build it with `make -C slam_amd/csrc` "
                                "(there is no CPU fallback)")
            _share_hip_runtime_with_torch()
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


class Context:
    """One GPU, one stream.  Not thread-safe; use one per thread."""

    def __init__(self, device=None):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("SLAM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = _vp()
        rc = self.lib.slam_create(int(device), C.byref(h))
        if rc != 0:
            raise SlamError(f"slam_create(device={device}) failed ({rc}): "
                            f"{self.lib.slam_last_error(None).decode()}")
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        if rc < 0:
            msg = self.lib.slam_last_error(self.handle).decode()
            if rc == -1:
                # the reference guards these preconditions with `assert`
                raise AssertionError(msg)
            raise SlamError(f"libslam_amd error {rc}: {msg}")
        return rc

    def pinned_empty(self, shape, dtype=np.float64):
        """An uninitialised numpy array in page-locked host memory (slam_host_alloc): transfers from and to it run at
        the full PCIe rate.  The memory is released when the array and all its views are gone."""
        import weakref
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(1, count * dtype.itemsize)
        p = C.c_void_p()
        self.check(self.lib.slam_host_alloc(self.handle, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
        lib, addr = self.lib, p.value
        weakref.finalize(buf, lambda: lib.slam_host_free(None, addr))
        return arr

    def close(self):
        if getattr(self, "handle", None):
            self.lib.slam_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context()
    return _default


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)
