"""The example driver loop (examples/slam.py: Slam) against the reference's own Slam object (custom_slam.py:118-254) run on a
synthetic three-frame sequence: tests/golden/driver.npz, made by tests/golden/make_golden.py (gen_driver).

CPU: the bookkeeping on the CPU restatement.  GPU: the same object on the library."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import slam  # noqa: E402  (examples/slam.py)
from slam_amd import logger  # noqa: E402
from _oracle_backend import OracleBackend

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver.npz")


def run(backend):
    g = np.load(GOLD)
    frames = [slam.Frame(g[f"kp{k}"], g[f"des{k}"]) for k in range(3)]
    logger.set_log_level("WARN")
    s = slam.Slam(g["K"], backend=backend)
    s.find_initial_pose(frames[0], frames[1])
    first = dict(n_map=len(s.triangulated_point_lookup[2]), latest=np.array(s.latest_image_2d_points),
                 points=np.array(s.all_triangulated_points))
    s.find_next_pose(frames[1], frames[2])
    return g, s, first


def check(g, s, first, pose_tol):
    assert len(s.camera_poses) == int(g["n_poses"]) == 3
    assert np.array_equal(s.camera_poses[0], np.identity(4))
    assert first["n_map"] == int(g["n_map_frame2"])
    assert np.array_equal(first["latest"], g["latest_image_2d_points"])          # same matches, same order
    assert first["points"].shape == g["all_triangulated_points"].shape            # same z > 0 flags
    # tolerances: 3D points 1e-5 relative, pose entries to pose_tol of the matrix scale (entries are O(1))
    np.testing.assert_allclose(first["points"], g["all_triangulated_points"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(s.camera_poses[1], g["pose1"], rtol=0, atol=pose_tol)
    np.testing.assert_allclose(s.camera_poses[2], g["pose2"], rtol=0, atol=pose_tol)


def test_driver_bookkeeping_on_cpu_restatement():
    g, s, first = run(OracleBackend())
    check(g, s, first, 1e-7)


def test_lookup_is_by_frame_and_exact_point():
    s = slam.Slam(np.identity(3), backend=OracleBackend())
    s.set_triangulated_point(2, np.array([1.5, 2.5]), np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(s.get_triangulated_point(2, (1.5, 2.5)), [1.0, 2.0, 3.0])
    assert s.get_triangulated_point(1, (1.5, 2.5)) is None
    assert s.get_triangulated_point(2, (1.5, 2.5000001)) is None
    with pytest.raises(AssertionError):
        s.get_triangulated_point(2, (1.0, 2.0, 3.0))
    assert s.current_frame_num() == 1
    s.reset()
    assert s.get_triangulated_point(2, (1.5, 2.5)) is None


@pytest.mark.gpu
def test_driver_on_gpu_matches_reference_driver():
    g, s, first = run(slam.GpuBackend())
    check(g, s, first, 1e-6)
