"""Pose algebra helpers with the reference's names and conventions (src/util.py:6-37).

A pose is the 4x4 matrix [[R^T, -R^T t], [0, 0, 0, 1]] for a camera with rotation R and centre t in world
coordinates; it maps world points into the camera frame.
"""
import numpy as np


def to_homogenous_coordinates(point):
    return np.concatenate([np.asarray(point), np.ones((len(point), 1))], axis=1)


def vector_to_cross_product_matrix(vector):
    assert vector.shape == (3,)
    x, y, z = vector
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


def pose_to_rotation(pose):
    assert pose.shape == (4, 4)
    return pose[:3, :3].T


def pose_to_translation(pose):
    assert pose.shape == (4, 4)
    return -1 * pose[:3, :3].T @ pose[:3, 3]


def rotation_translation_to_pose(rotation, translation):
    assert rotation.shape == (3, 3)
    assert translation.shape == (3,)
    pose = np.zeros((4, 4))
    pose[:3, :3] = rotation.T
    pose[:3, 3] = -1 * rotation.T @ translation
    pose[3, 3] = 1
    return pose


def point_distance(pt1, pt2):
    return np.linalg.norm(pt1 - pt2)


def rq_decomposition(m):
    """Returns (r, q) exactly as the reference's helper does (src/util.py:60-66): with J the row-reversal
    matrix and J m = q' r' (QR), r = J r'^T J (upper triangular) and q = J q'^T (orthogonal).
    Note that r @ q equals J m^T J, not m; the tracking code that consumes it relies on this form."""
    flip = np.eye(m.shape[0])[::-1]
    q_, r_ = np.linalg.qr(flip @ m)
    return flip @ r_.T @ flip, flip @ q_.T


def collect_symbolic_equation_with_respect_to_vars(eq, vars):
    """Coefficients of a sympy expression as a polynomial in `vars`: {monomial: coefficient}
    (reference: src/util.py:45-57; only the reference's symbolic research scripts use it)."""
    assert isinstance(vars, list)
    eq = eq.expand()
    if not vars:
        return {1: eq}
    out = {}
    for power, rest in eq.collect(vars[0], evaluate=False).items():
        for sub_power, coeff in collect_symbolic_equation_with_respect_to_vars(rest, vars[1:]).items():
            out[power * sub_power] = coeff
    return out
