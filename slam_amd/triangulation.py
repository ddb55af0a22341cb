"""Linear DLT triangulation on the GPU, with the reference's interface (src/triangulation.py)."""
from __future__ import annotations

import numpy as np

from . import _lib
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
