"""Size-independent property of the DLT triangulation (src/triangulation.py:50-73) on inputs the goldens do not
cover: arbitrary 3x4 cameras (not K [R|t]), image points anywhere, matches that are no matches at all.  The point the
library returns is X = Y[:3] / Y[3] with Y = diag(m) y, so y is recovered from X up to scale, and y must be the least
right singular vector of the column-scaled matrix D': |D' y| equals the smallest singular value numpy finds, to a
relative 1e-12 - whatever the gap to the next singular value, which is what decides how far the VECTORS of two
solvers may differ and makes a comparison of points a test of the data rather than of the solver."""
import numpy as np
import pytest

from slam_amd import triangulation

pytestmark = pytest.mark.gpu


def _scaled_matrices(P0, P1, p0, p1):
    D = np.empty((len(p0), 4, 4))
    D[:, 0, :] = p0[:, 1:2] * P0[2] - P0[1]
    D[:, 1, :] = -p0[:, 0:1] * P0[2] + P0[0]
    D[:, 2, :] = p1[:, 1:2] * P1[2] - P1[1]
    D[:, 3, :] = -p1[:, 0:1] * P1[2] + P1[0]
    m = np.abs(D).max(axis=2)          # m_i = max_j |D_ij|  (:58)
    return D / m[:, None, :], m        # column j divided by m_j  (:60-62)


@pytest.mark.parametrize("kind", ["gaussian cameras", "poses as cameras", "near-parallel rays"])
def test_returned_point_is_the_least_singular_direction(kind):
    rng = np.random.default_rng({"gaussian cameras": 1, "poses as cameras": 2, "near-parallel rays": 3}[kind])
    n = 20000
    if kind == "gaussian cameras":
        P0, P1 = rng.normal(size=(3, 4)), rng.normal(size=(3, 4))
        p0, p1 = rng.normal(size=(n, 2)), rng.normal(size=(n, 2))
    elif kind == "poses as cameras":   # what the reference's tracking step triangulates with (src/triangulation.py:179)
        def pose():
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            return np.hstack([q * np.sign(np.linalg.det(q)), rng.normal(size=(3, 1))])
        P0, P1 = pose(), pose()
        p0, p1 = rng.uniform(0, 1200, (n, 2)), rng.uniform(0, 400, (n, 2))
    else:
        K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.22], [0, 0, 1.0]])
        P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
        P1 = K @ np.hstack([np.eye(3), np.array([[1e-4], [0.0], [0.0]])])   # a baseline of a tenth of a millimetre
        p0 = rng.uniform(0, 1200, (n, 2))
        p1 = p0 + rng.normal(0, 0.3, (n, 2))
    X = triangulation.triangulate_points_from_pose(P0, P1, p0, p1)
    Dp, m = _scaled_matrices(P0, P1, p0, p1)
    sigma = np.linalg.svd(Dp, compute_uv=False)
    ok = np.isfinite(X).all(axis=1)
    assert ok.mean() > 0.99             # (a point at infinity has Y[3] = 0: the reference divides by it too)
    y = np.hstack([X, np.ones((n, 1))]) / m
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    res = np.linalg.norm(np.einsum("nij,nj->ni", Dp, y), axis=1)
    excess = (res[ok] - sigma[ok, 3]) / sigma[ok, 0]
    print(f"{kind}: largest excess of |D' y| over the smallest singular value, relative to the largest: {excess.max():.2e}")
    assert excess.max() < 1e-12


def test_pose_recovery_does_not_depend_on_the_launch_shape():
    """The pose recovery kernel takes 64, 256 or 1024 matches per workgroup depending on how many there are (its
    count atomics queue on four words).  A match's point must not depend on that: the first n matches of a list give
    the same points, bit for bit, whether the list ends there or goes on across a shape boundary, and the cheirality
    counts of the longer list are those of the shorter one plus the extra matches' own."""
    from slam_amd import epipolar, synth
    base = synth.make_pair(1 << 16, 0.1, seed=31)
    E = synth.essential_from_motion(base["R"], base["t"])
    K = base["K"]
    reps = 17
    p0 = np.tile(base["pts0"], (reps, 1))
    p1 = np.tile(base["pts1"], (reps, 1)) + np.random.default_rng(32).uniform(0, 0.25, (reps << 16, 2))
    sizes = [4096, (1 << 16), (1 << 16) + 1, (1 << 20) - 37, (1 << 20), (1 << 20) + 1, reps << 16]
    ref_pose, ref_X = epipolar.calculate_pose_from_essential_matrix(E, p0[:sizes[-1]], p1[:sizes[-1]], K)
    for n in sizes[:-1]:
        pose, X = epipolar.calculate_pose_from_essential_matrix(E, p0[:n], p1[:n], K)
        assert np.array_equal(pose, ref_pose), n      # (the winner is the same for every prefix of this list)
        assert np.array_equal(X, ref_X[:n], equal_nan=True), n
