#!/usr/bin/env python3
"""Time of the tracking step through its host entry point (numpy in, numpy out: the drop-in function's cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import logger, synth, triangulation
logger.set_log_level("WARN")
d = synth.make_track(300, 1000, 0.3, seed=7)
args = (d["previous_pose"], d["prev_known"], d["next_known"], d["X_known"], d["prev_all"], d["next_all"], d["K"])
for _ in range(20):
    triangulation.triangulate_pose_from_points_with_ransac(*args)
t0 = time.perf_counter()
for _ in range(300):
    triangulation.triangulate_pose_from_points_with_ransac(*args)
print(f"tracking step, host buffers: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
