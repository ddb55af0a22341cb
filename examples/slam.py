"""EXAMPLE, not part of the library: the reference driver's `Slam` class (custom_slam.py:118-254) on top of the GPU
path.  It shows how an application would call the matcher, the pair pipeline and the tracking step in sequence.

What differs from the reference, and only that: a frame is given as its keypoints and descriptors (``Frame``)
instead of an image, because ORB detection is OpenCV's and stays with the caller; there is no visualisation thread.
The bookkeeping is the reference's: camera poses start with the identity, triangulated points are looked up by
(frame number, exact 2D point), `find_initial_pose` registers every triangulated point for both frames (good or not,
as the reference does with its `if is_good` commented out), `find_next_pose` collects the matches whose previous
point has a 3D position and hands them to the tracking RANSAC.

The three heavy steps are looked up on a backend object so that tests can run the same bookkeeping on a CPU
restatement: `matches(f0, f1)`, `initialise(p0, p1, K)`, `track(pose, prev_known, next_known, X_known, prev_all,
next_all, K)`.
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

from slam_amd import logger, util

Frame = namedtuple("Frame", ["keypoints", "descriptors"])  # (n,2) float64 pixel positions; (n,32) uint8 or (n,256) bits


class GpuBackend:
    """The library's entry points for the three steps (one context; see slam_amd._lib.default_context)."""

    def __init__(self, ctx=None, iterations=1000, tracking_iterations=100):
        self.ctx = ctx
        self.iterations = iterations
        self.tracking_iterations = tracking_iterations

    def matches(self, f0, f1):
        from slam_amd import matching
        return matching.matches_between_keypoints(f0.keypoints, f0.descriptors, f1.keypoints, f1.descriptors, ctx=self.ctx)

    def initialise(self, img0_points, img1_points, K):
        from slam_amd import pipeline
        r = pipeline.initialise_pair(img0_points, img1_points, K, self.iterations, ctx=self.ctx)
        return r["pose"], r["X"], r["mask"]

    def pair_from_frames(self, f0, f1, K):
        from slam_amd import pipeline
        r = pipeline.frame_pair(f0.keypoints, f0.descriptors, f1.keypoints, f1.descriptors, K, self.iterations, ctx=self.ctx)
        return r["pose"], [r["matched0"], r["matched1"]], r["X"], r["mask"]

    def track(self, pose, prev_known, next_known, X_known, prev_all, next_all, K):
        from slam_amd import triangulation
        return triangulation.triangulate_pose_from_points_with_ransac(pose, prev_known, next_known, X_known, prev_all,
                                                                      next_all, K, self.tracking_iterations, ctx=self.ctx)


def find_pose_and_triangulated_points(frame0, frame1, intrinsic_camera_matrix, backend):
    """reference: custom_slam.py:105-115 (matching, F by RANSAC, E, pose and points, z > 0 flags)"""
    if hasattr(backend, "pair_from_frames"):  # both stages in one library call, the matches never leave the device
        return backend.pair_from_frames(frame0, frame1, intrinsic_camera_matrix)
    img0_points, img1_points = backend.matches(frame0, frame1)
    pose, X, good = backend.initialise(img0_points, img1_points, intrinsic_camera_matrix)
    return pose, [img0_points, img1_points], X, good


class Slam:
    def __init__(self, intrinsic_camera_matrix, backend=None):
        self.intrinsic_camera_matrix = np.asarray(intrinsic_camera_matrix, dtype=np.float64)
        self.backend = backend if backend is not None else GpuBackend()
        self.latest_image_2d_points = None
        self.camera_poses = None
        self.triangulated_point_lookup = None
        self.all_triangulated_points = None
        self.reset()

    def reset(self):                                             # custom_slam.py:140-144
        self.camera_poses = [np.identity(4)]
        self.triangulated_point_lookup = {}
        self.all_triangulated_points = None

    def get_triangulated_point(self, frame_num, point):          # custom_slam.py:157-159
        assert len(point) == 2
        return self.triangulated_point_lookup.get(frame_num, {}).get(tuple(point))

    def set_triangulated_point(self, frame_num, point, triangulated_point):   # custom_slam.py:161-164
        assert len(point) == 2
        self.triangulated_point_lookup.setdefault(frame_num, {})[tuple(point)] = triangulated_point

    def current_frame_num(self):                                 # custom_slam.py:166-167
        return len(self.camera_poses)

    def add_camera_pose(self, pose):
        self.camera_poses.append(pose)

    def add_triangulated_points(self, points):                   # custom_slam.py:173-178 without the viewer's lock
        if self.all_triangulated_points is None:
            self.all_triangulated_points = np.asarray(points)
        else:
            self.all_triangulated_points = np.append(self.all_triangulated_points, points, axis=0)

    def get_latest_camera_pose(self):
        return self.camera_poses[-1]

    def log_camera_pose(self, camera_pose):                      # custom_slam.py:200-201
        logger.info("Translation: ", util.pose_to_translation(camera_pose), ", camera vector: ",
                    util.pose_to_rotation(camera_pose) @ [0, 0, 1])

    def find_initial_pose(self, prev_frame, new_frame):
        """Two frames -> relative camera pose (up to scale) and triangulated points   (custom_slam.py:203-227)."""
        logger.info("-----------------------")
        logger.info("Finding initial pose...")
        self.reset()
        current_frame_num = self.current_frame_num()
        camera_pose, points2d, points3d, good_points_mask = find_pose_and_triangulated_points(
            prev_frame, new_frame, self.intrinsic_camera_matrix, self.backend)
        self.log_camera_pose(camera_pose)
        # every point, good or not (custom_slam.py:219-224), registered for both frames; built in bulk: the keys are
        # the same (x, y) tuples set_triangulated_point makes, a later duplicate of a 2D point wins as it does there
        for frame_num, pts in ((current_frame_num, points2d[0]), (current_frame_num + 1, points2d[1])):
            self.triangulated_point_lookup.setdefault(frame_num, {}).update(
                zip(map(tuple, np.asarray(pts).tolist()), points3d[:len(good_points_mask)]))
        self.add_triangulated_points(points3d[good_points_mask])
        self.add_camera_pose(camera_pose)
        self.latest_image_2d_points = points2d[1]
        return camera_pose

    def find_next_pose(self, prev_frame, new_frame):
        """The next camera pose from the matches whose previous point has a 3D position   (custom_slam.py:229-254)."""
        logger.info("-----------------------")
        logger.info("Finding next pose...")
        current_frame_num = self.current_frame_num()
        latest_camera_pose = self.get_latest_camera_pose()
        prev_image_points, new_image_points = self.backend.matches(prev_frame, new_frame)
        # the matches whose previous point has a 3D position (custom_slam.py:239-245), looked up in bulk
        known_here = self.triangulated_point_lookup.get(current_frame_num, {})
        found = [known_here.get(key) for key in map(tuple, np.asarray(prev_image_points).tolist())]
        keep = [i for i, point_3d in enumerate(found) if point_3d is not None]
        pose, points = self.backend.track(latest_camera_pose, np.asarray(prev_image_points)[keep],
                                          np.asarray(new_image_points)[keep], np.array([found[i] for i in keep]),
                                          prev_image_points, new_image_points, self.intrinsic_camera_matrix)
        self.log_camera_pose(pose)
        self.add_camera_pose(pose)
        return pose
