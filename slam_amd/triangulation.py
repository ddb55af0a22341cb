"""Linear DLT triangulation on the GPU, with the reference's interface (src/triangulation.py)."""
from __future__ import annotations

import numpy as np

from . import _lib, logger
from ._lib import f64, ptr


def triangulate_points_from_pose(img0_camera_matrix, img1_camera_matrix, img0_points, img1_points, ctx=None):
    """3D points (N,3) from two camera matrices and N matches   (reference: src/triangulation.py:8-75).

    A camera matrix may be 3x4 or 4x4; only its first three rows are used (the reference's tracking step passes
    4x4 poses: src/triangulation.py:179-180).  The result reproduces the reference's scaling sequence
    (row maxima applied to columns before the SVD and multiplied back afterwards), not a textbook DLT.
    """
    P0 = f64(np.asarray(img0_camera_matrix)[:3])
    P1 = f64(np.asarray(img1_camera_matrix)[:3])
    assert P0.shape == (3, 4) and P1.shape == (3, 4)
    p0 = f64(np.asarray(img0_points)).reshape(-1, 2)
    p1 = f64(np.asarray(img1_points)).reshape(-1, 2)
    assert len(p0) == len(p1)                        # reference: src/triangulation.py:74
    X = np.empty((len(p0), 3))
    if len(p0) == 0:
        return X
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_triangulate(ctx.handle, ptr(P0), ptr(P1), ptr(p0), ptr(p1), len(p0), ptr(X)))
    return X


def pose_from_points_detailed(image_points, known_3d_points, ctx=None):
    """Pose plus the intermediate quantities the reference prints (calculated K, rotation, translation, centre)."""
    uv = f64(np.asarray(image_points)).reshape(-1, 2)
    X3 = f64(np.asarray(known_3d_points)).reshape(-1, 3)
    assert len(X3) == len(uv)                        # reference: src/triangulation.py:81
    assert len(X3) >= 6                              # reference: src/triangulation.py:82
    pose = np.empty((4, 4))
    extra = np.empty(24)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_pose_from_points(ctx.handle, ptr(uv), ptr(X3), len(uv), ptr(pose), ptr(extra)))
    return dict(pose=pose, K=extra[:9].reshape(3, 3), R=extra[9:18].reshape(3, 3), t=extra[18:21].copy(),
                centre=extra[21:24].copy())


def triangulate_pose_from_points(image_points, known_3d_points, intrinsic_camera_matrix, ctx=None):
    """Camera pose (4x4) from >= 6 matches between image points and known 3D points by DLT resection
    (reference: src/triangulation.py:78-132).  Like the reference, the intrinsic matrix argument is accepted and
    not used, and the calculated intrinsic matrix, rotation and translation are printed."""
    r = pose_from_points_detailed(image_points, known_3d_points, ctx)
    print("calculated intrinsic camera matrix")      # reference: src/triangulation.py:126-131
    print(r["K"])
    print("rotation")
    print(r["R"])
    print("translation")
    print(r["t"])
    return r["pose"]


def track_pose_detailed(previous_camera_pose, matching_previous_image_points_with_3d_position,
                        matching_next_image_points_with_3d_position, matching_3d_points,
                        matching_previous_image_points, matching_next_image_points, intrinsic_camera_matrix,
                        iterations=100, ctx=None, want_all=True):
    """The tracking RANSAC with its per-hypothesis results (counts, errors, poses) and the winner's position."""
    prev_pose = f64(np.asarray(previous_camera_pose))
    assert prev_pose.shape == (4, 4)
    a = f64(np.asarray(matching_previous_image_points_with_3d_position)).reshape(-1, 2)
    b = f64(np.asarray(matching_next_image_points_with_3d_position)).reshape(-1, 2)
    X3 = f64(np.asarray(matching_3d_points)).reshape(-1, 3)
    assert len(a) == len(b) == len(X3)               # reference: src/triangulation.py:143
    assert len(X3) >= 6                              # reference: src/triangulation.py:144
    pa = f64(np.asarray(matching_previous_image_points)).reshape(-1, 2)
    pb = f64(np.asarray(matching_next_image_points)).reshape(-1, 2)
    assert len(pa) == len(pb)                        # reference: src/triangulation.py:74 (the final triangulation)
    K = f64(np.asarray(intrinsic_camera_matrix))
    assert K.shape == (3, 3)
    ctx = ctx or _lib.default_context()
    M, N = len(X3), len(pa)
    H = int(ctx.lib.slam_num_tracking_hypotheses(M, int(iterations)))
    assert H >= 1, "iterations < 1"
    pose = np.empty((4, 4))
    X = np.empty((N, 3))
    best = np.empty(3)
    counts = np.empty(H, dtype=np.int32) if want_all else None
    errs = np.empty(H) if want_all else None
    poses = np.empty((H, 4, 4)) if want_all else None
    ctx.check(ctx.lib.slam_track_pose(ctx.handle, ptr(prev_pose), ptr(a), ptr(b), ptr(X3), M, ptr(pa), ptr(pb), N, ptr(K),
                                      int(iterations), ptr(pose), ptr(X) if N else None, ptr(best), ptr(counts),
                                      ptr(errs), ptr(poses)))
    return dict(pose=pose, X=X, best_index=int(best[0]), best_count=int(best[1]), best_err=float(best[2]),
                counts=counts, errs=errs, poses=poses)


def triangulate_pose_from_points_with_ransac(previous_camera_pose,
                                             matching_previous_image_points_with_3d_position,
                                             matching_next_image_points_with_3d_position,
                                             matching_3d_points,
                                             matching_previous_image_points,
                                             matching_next_image_points,
                                             intrinsic_camera_matrix,
                                             iterations=100, ctx=None):
    """Next camera pose and the triangulation of every match   (reference: src/triangulation.py:135-182).

    Same samples (random.seed(0x1337BEEF + iteration); random.sample(range(M), 6)), same scoring (points in front
    of both cameras, then the summed distance to the known 3D points) and the same first-best selection as the
    reference.  Its per-iteration debugging prints (the solver's matrices and every triangulated array) are not
    reproduced; the INFO line is.
    """
    r = track_pose_detailed(previous_camera_pose, matching_previous_image_points_with_3d_position,
                            matching_next_image_points_with_3d_position, matching_3d_points,
                            matching_previous_image_points, matching_next_image_points, intrinsic_camera_matrix,
                            iterations, ctx, want_all=False)
    logger.info("Pose from points, num good points: ", r["best_count"], "/", len(matching_3d_points))
    assert r["pose"].shape == (4, 4)                 # reference: src/triangulation.py:178
    return r["pose"], r["X"]
