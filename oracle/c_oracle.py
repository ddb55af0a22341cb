"""ctypes access to oracle/liboracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, the smoke check and the benchmark's CPU-baseline leg, never by slam_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_sample.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, _i32]
        L.oracle_mt_raw.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, _u32]
        L.oracle_eight_point.argtypes = [_f64, _f64, C.c_int, _f64]
        L.oracle_ransac.argtypes = [_f64, _f64, C.c_int, _i32, C.c_int, C.c_double, C.c_void_p, _i32, _f64,
                                    C.POINTER(C.c_int32), _f64, C.c_void_p, C.c_int]
        L.oracle_normalize_points.argtypes = [_f64, C.c_int, _f64, _f64]
        L.oracle_triangulate.argtypes = [_f64, _f64, _f64, _f64, C.c_int64, _f64, C.c_int]
        L.oracle_pose_candidates.argtypes = [_f64, _f64, _f64]
        L.oracle_pose_from_E.argtypes = [_f64, _f64, _f64, _f64, C.c_int64, _f64, _f64, _i32,
                                         C.POINTER(C.c_int32), C.c_int]
        L.oracle_pose_from_points.argtypes = [_f64, _f64, C.c_int, _f64, C.c_void_p]
        L.oracle_tracking_ransac.argtypes = [_f64, _f64, _f64, _f64, C.c_int, _f64, _i32, C.c_int, _f64, _i32, _f64,
                                             C.POINTER(C.c_int32)]
        _u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.oracle_match_knn.argtypes = [_u8, _u8, C.c_int, C.c_int, C.c_int, _i32, _i32]
        L.oracle_select_matches.argtypes = [_f64, _f64, _i32, _i32, C.c_int, _f64, _f64]
        _LIB = L
    return _LIB


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def sample(N, H, k=8, seed_base=0x1337BEEF, it0=0):
    idx = np.empty((H, k), dtype=np.int32)
    assert lib().oracle_sample(seed_base, it0, H, N, k, idx) == 0
    return idx


def mt_raw(H, count, seed_base=0x1337BEEF, it0=0):
    out = np.empty((H, count), dtype=np.uint32)
    lib().oracle_mt_raw(seed_base, it0, H, count, out)
    return out


def eight_point(p0, p1):
    p0, p1 = _c(p0), _c(p1)
    F = np.empty((3, 3))
    assert lib().oracle_eight_point(p0, p1, len(p0), F) == 0
    return F


def ransac(p0, p1, idx, thr=0.005, threads=1, want_F=True, want_mask=True):
    p0, p1, idx = _c(p0), _c(p1), _c(idx, np.int32)
    H, N = len(idx), len(p0)
    Fs = np.empty((H, 3, 3)) if want_F else None
    counts = np.empty(H, dtype=np.int32)
    errsums = np.empty(H)
    bi = C.c_int32(-1)
    bF = np.empty((3, 3))
    mask = np.empty(N, dtype=np.uint8) if want_mask else None
    rc = lib().oracle_ransac(p0, p1, N, idx, H, thr, Fs.ctypes.data if want_F else None, counts, errsums,
                             C.byref(bi), bF, mask.ctypes.data if want_mask else None, threads)
    assert rc == 0
    return dict(F=Fs, counts=counts, errsums=errsums, best_index=int(bi.value), best_F=bF,
                best_mask=None if mask is None else mask.astype(bool),
                best_count=int(counts[bi.value]), best_errsum=float(errsums[bi.value]))


def ransac_timed(p0, p1, idx, thr=0.005, threads=1, reps=5):
    """Best and mean seconds per pass of `ransac` over `reps` passes after one warm-up pass, timed inside the C
    library on the monotonic clock; also the winner's index (for a check against the GPU)."""
    p0, p1, idx = _c(p0), _c(p1), _c(idx, np.int32)
    H, N = len(idx), len(p0)
    counts = np.empty(H, dtype=np.int32)
    errsums = np.empty(H)
    bi = C.c_int32(-1)
    best, mean = C.c_double(0), C.c_double(0)
    fn = lib().oracle_ransac_timed
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p,
                   C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rc = fn(p0.ctypes.data, p1.ctypes.data, N, idx.ctypes.data, H, thr, threads, reps, counts.ctypes.data,
            errsums.ctypes.data, C.byref(bi), C.byref(best), C.byref(mean))
    assert rc == 0
    return dict(best_seconds=best.value, mean_seconds=mean.value, best_index=int(bi.value))


def normalize_points(p):
    p = _c(p)
    out = np.empty_like(p)
    T = np.empty((3, 3))
    lib().oracle_normalize_points(p, len(p), out, T)
    return out, T


def triangulate(P0, P1, p0, p1, threads=1):
    P0, P1 = _c(np.asarray(P0)[:3]), _c(np.asarray(P1)[:3])
    p0, p1 = _c(p0), _c(p1)
    X = np.empty((len(p0), 3))
    lib().oracle_triangulate(P0, P1, p0, p1, len(p0), X, threads)
    return X


def pose_candidates(E):
    Rs = np.empty((4, 3, 3))
    ts = np.empty((4, 3))
    lib().oracle_pose_candidates(_c(E), Rs, ts)
    return Rs, ts


def pose_from_E(E, K, p0, p1, threads=1):
    p0, p1 = _c(p0), _c(p1)
    pose = np.empty((4, 4))
    X = np.empty((len(p0), 3))
    counts = np.empty(4, dtype=np.int32)
    w = C.c_int32(-1)
    lib().oracle_pose_from_E(_c(E), _c(K), p0, p1, len(p0), pose, X, counts, C.byref(w), threads)
    return pose, X, counts, int(w.value)


def pose_from_points(image_points, known_3d_points, want_extra=False):
    uv, X = _c(image_points), _c(known_3d_points)
    pose = np.empty((4, 4))
    extra = np.empty(24)
    assert lib().oracle_pose_from_points(uv, X, len(uv), pose, extra.ctypes.data) == 0
    if want_extra:
        return pose, dict(K=extra[:9].reshape(3, 3), R=extra[9:18].reshape(3, 3), t=extra[18:21], centre=extra[21:24])
    return pose


def tracking_ransac(previous_pose, prev_known, next_known, known_3d_points, K, idx):
    a, b, X3, idx = _c(prev_known), _c(next_known), _c(known_3d_points), _c(idx, np.int32)
    H = len(idx)
    poses = np.empty((H, 4, 4))
    counts = np.empty(H, dtype=np.int32)
    errs = np.empty(H)
    bi = C.c_int32(-1)
    rc = lib().oracle_tracking_ransac(_c(previous_pose), a, b, X3, len(a), _c(K), idx, H, poses, counts, errs,
                                      C.byref(bi))
    assert rc == 0
    return dict(poses=poses, counts=counts, errs=errs, best_index=int(bi.value), pose=poses[bi.value])


def match_knn(des0, des1, k=2):
    des0, des1 = _c(des0, np.uint8), _c(des1, np.uint8)
    assert des0.shape[1] == 32 and des1.shape[1] == 32
    index = np.empty((len(des0), k), dtype=np.int32)
    distance = np.empty((len(des0), k), dtype=np.int32)
    assert lib().oracle_match_knn(des0, des1, len(des0), len(des1), k, index, distance) == 0
    return index, distance


def select_matches(pts0, pts1, index, distance):
    pts0, pts1 = _c(pts0), _c(pts1)
    index, distance = _c(index, np.int32), _c(distance, np.int32)
    m0 = np.empty((len(pts0), 2))
    m1 = np.empty((len(pts0), 2))
    n = lib().oracle_select_matches(pts0, pts1, index, distance, len(pts0), m0, m1)
    return m0[:n].copy(), m1[:n].copy()
