"""The three steps of the driver loop on the CPU restatement (test infrastructure: the checker for examples/slam.py)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, slam_oracle  # noqa: E402


class OracleBackend:
    def __init__(self, iterations=1000, tracking_iterations=100):
        self.iterations = iterations
        self.tracking_iterations = tracking_iterations

    def matches(self, f0, f1):
        d0, d1 = slam_oracle.pack_descriptors(f0.descriptors), slam_oracle.pack_descriptors(f1.descriptors)
        index, distance = c_oracle.match_knn(d0, d1, 2)
        return c_oracle.select_matches(f0.keypoints, f1.keypoints, index, distance)

    def initialise(self, p0, p1, K):
        n0, T0 = c_oracle.normalize_points(p0)
        n1, T1 = c_oracle.normalize_points(p1)
        H = slam_oracle.num_hypotheses(len(p0), self.iterations)
        r = c_oracle.ransac(n0, n1, c_oracle.sample(len(p0), H), want_F=False, want_mask=False)
        F = slam_oracle.denormalize(r["best_F"], T0, T1)
        E = slam_oracle.fundamental_to_essential(F, K)
        pose, X, _, _ = c_oracle.pose_from_E(E, K, p0, p1)
        return pose, X, X[:, 2] > 0

    def track(self, pose, prev_known, next_known, X_known, prev_all, next_all, K):
        M = len(X_known)
        assert M >= 6
        H = min(self.tracking_iterations, math.comb(M, 6))
        r = c_oracle.tracking_ransac(pose, prev_known, next_known, X_known, K, c_oracle.sample(M, H, k=6))
        return r["pose"], c_oracle.triangulate(pose, r["pose"], prev_all, next_all)
