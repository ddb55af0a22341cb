"""Eight-point fundamental-matrix estimation, RANSAC and pose recovery on the GPU.

Same names, arguments, return values, printed lines and failure behaviour as the reference's
src/epipolar.py; the arithmetic runs in libslam_amd.so (HIP, gfx950).  Functions ending in ``_detailed``
return what the reference computes but does not expose (winner index, inlier mask, per-hypothesis scores).
"""
from __future__ import annotations

import random

import numpy as np

from . import _lib, logger
from ._lib import REC_COUNT, REC_ERRSUM, REC_F, REC_INDEX, REC_LEN, f64, ptr


def two_d_to_three_d(pt):
    """(x, y) -> (x, y, 1)   (reference: src/epipolar.py:13-15)"""
    return np.asarray([pt[0], pt[1], 1])


def _pair(img0_points, img1_points):
    p0 = np.asarray(img0_points)
    p1 = np.asarray(img1_points)
    assert p0.shape == p1.shape                      # reference: src/epipolar.py:22,59
    assert len(p0) >= 8                              # reference: src/epipolar.py:23,60
    assert p0.ndim == 2 and p0.shape[1] == 2
    return f64(p0), f64(p1)


def calculate_fundamental_matrix(img0_points, img1_points):
    """F (3x3, rank 2) with x1^T F x0 = 0 from M >= 8 matches   (reference: src/epipolar.py:18-51)."""
    p0, p1 = _pair(img0_points, img1_points)
    ctx = _lib.default_context()
    F = np.empty((3, 3))
    ctx.check(ctx.lib.slam_eightpoint(ctx.handle, ptr(p0), ptr(p1), len(p0), ptr(F)))
    return F


def ransac_detailed(img0_points, img1_points, iterations=1000, idx=None, sampler=_lib.SAMPLER_CPYTHON,
                    seed=_lib.SEED_BASE, it0=0, threshold=_lib.DEFAULT_THRESHOLD, per_hypothesis=False, ctx=None,
                    scoring=_lib.SCORE_ALGEBRAIC):
    """Run the RANSAC loop of src/epipolar.py:54-84 and return everything it computes.

    idx: optional (H,8) pre-drawn index sets (then ``iterations`` is ignored); otherwise H =
    min(iterations, C(N,8)) samples are drawn on the GPU with ``sampler``.
    scoring: _lib.SCORE_ALGEBRAIC is the reference's residual; _lib.SCORE_SAMPSON (an extension) scores with the
    Sampson distance, to which ``threshold`` then applies.
    """
    p0, p1 = _pair(img0_points, img1_points)
    ctx = ctx or _lib.default_context()
    N = len(p0)
    if idx is not None:
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        assert idx.ndim == 2 and idx.shape[1] == 8
        assert idx.size == 0 or (idx.min() >= 0 and idx.max() < N)
        H = len(idx)
    else:
        H = int(ctx.lib.slam_num_hypotheses(N, int(iterations)))
    record = np.empty(REC_LEN)
    mask = np.empty(N, dtype=np.uint8)
    counts = np.empty(H, dtype=np.int32) if per_hypothesis else None
    errsums = np.empty(H) if per_hypothesis else None
    Fs = np.empty((H, 3, 3)) if per_hypothesis else None
    ctx.check(ctx.lib.slam_set_scoring(ctx.handle, int(scoring)))
    try:
        ctx.check(ctx.lib.slam_ransac_f(ctx.handle, ptr(p0), ptr(p1), N, ptr(idx), int(sampler), int(seed), int(it0), H,
                                        float(threshold), ptr(record), ptr(mask), ptr(counts), ptr(errsums), ptr(Fs)))
    finally:
        ctx.lib.slam_set_scoring(ctx.handle, _lib.SCORE_ALGEBRAIC)
    return dict(F=record[REC_F:REC_F + 9].reshape(3, 3).copy(), best_index=int(record[REC_INDEX]),
                best_count=int(record[REC_COUNT]), best_errsum=float(record[REC_ERRSUM]), best_mask=mask.astype(bool),
                counts=counts, errsums=errsums, F_all=Fs, H=H, record=record)


def ransac_adaptive(img0_points, img1_points, max_iterations=1000, confidence=0.99, chunk=2048,
                    sampler=_lib.SAMPLER_CPYTHON, seed=_lib.SEED_BASE, threshold=_lib.DEFAULT_THRESHOLD, ctx=None,
                    scoring=_lib.SCORE_ALGEBRAIC):
    """RANSAC with the standard stopping rule (an extension: the reference always runs every iteration).  Scores
    iterations 0, 1, ... in chunks and stops once ceil(log(1-confidence)/log(1-w^8)) iterations have been scored, w
    being the best inlier ratio so far.  The result equals ransac_detailed(iterations=r["iterations_run"])."""
    p0, p1 = _pair(img0_points, img1_points)
    ctx = ctx or _lib.default_context()
    N = len(p0)
    record = np.empty(REC_LEN)
    mask = np.empty(N, dtype=np.uint8)
    ran = np.zeros(1, dtype=np.int64)
    ctx.check(ctx.lib.slam_set_scoring(ctx.handle, int(scoring)))
    try:
        ctx.check(ctx.lib.slam_ransac_f_adaptive(ctx.handle, ptr(p0), ptr(p1), N, int(sampler), int(seed), int(max_iterations),
                                                 int(chunk), float(confidence), float(threshold), ptr(record), ptr(mask),
                                                 ptr(ran)))
    finally:
        ctx.lib.slam_set_scoring(ctx.handle, _lib.SCORE_ALGEBRAIC)
    return dict(F=record[REC_F:REC_F + 9].reshape(3, 3).copy(), best_index=int(record[REC_INDEX]),
                best_count=int(record[REC_COUNT]), best_errsum=float(record[REC_ERRSUM]), best_mask=mask.astype(bool),
                iterations_run=int(ran[0]), record=record)


def calculate_fundamental_matrix_with_ransac(img0_points, img1_points, iterations=1000):
    """Best-of-``iterations`` eight-point F by inlier count   (reference: src/epipolar.py:54-84)."""
    r = ransac_detailed(img0_points, img1_points, iterations)
    # the reference leaves the process-wide generator seeded by its last iteration and advanced by one draw
    random.seed(_lib.SEED_BASE + r["H"] - 1)
    random.sample(range(len(img0_points)), 8)
    logger.info("fundamental matrix ransac - num inliers: ", r["best_count"], "/", len(img0_points),
                " error: ", np.float64(r["best_errsum"]))
    return r["F"]


def fundamental_to_essential_matrix(fundamental_matrix, intrinsic_camera_matrix):
    """E = K^T F K / ||K^T F K||_F   (reference: src/epipolar.py:87-92)."""
    F = f64(fundamental_matrix)
    K = f64(intrinsic_camera_matrix)
    assert F.shape == (3, 3) and K.shape == (3, 3)
    ctx = _lib.default_context()
    E = np.empty((3, 3))
    ctx.check(ctx.lib.slam_fundamental_to_essential(ctx.handle, ptr(F), ptr(K), ptr(E)))
    return E


def pose_detailed(essential_matrix, img0_points, img1_points, intrinsic_camera_matrix, ctx=None):
    E = f64(essential_matrix)
    K = f64(intrinsic_camera_matrix)
    p0 = f64(np.asarray(img0_points))
    p1 = f64(np.asarray(img1_points))
    assert E.shape == (3, 3) and K.shape == (3, 3)
    assert p0.shape == p1.shape and p0.ndim == 2 and p0.shape[1] == 2 and len(p0) >= 1
    ctx = ctx or _lib.default_context()
    N = len(p0)
    pose = np.empty((4, 4))
    X = np.empty((N, 3))
    counts4 = np.empty(4, dtype=np.int32)
    winner = np.empty(1, dtype=np.int32)
    ctx.check(ctx.lib.slam_pose_from_E(ctx.handle, ptr(E), ptr(K), ptr(p0), ptr(p1), N, ptr(pose), ptr(X), ptr(counts4),
                                       ptr(winner)))
    return dict(pose=pose, X=X, counts=counts4, winner=int(winner[0]))


def calculate_pose_from_essential_matrix(essential_matrix, img0_points, img1_points, intrinsic_camera_matrix):
    """Pose of camera 2 and the triangulated matches for the (R, t) decomposition of E that puts most points in
    front of both cameras   (reference: src/epipolar.py:95-136)."""
    r = pose_detailed(essential_matrix, img0_points, img1_points, intrinsic_camera_matrix)
    logger.debug("essential matrix - winning num points: %d/%d" % (r["counts"][r["winner"]], len(img0_points)))
    return r["pose"], r["X"]
