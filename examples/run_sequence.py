#!/usr/bin/env python3
"""The reference driver's two calls (custom_slam.py:257-272: find_initial_pose on frames 0 and 1, find_next_pose on
frames 1 and 2) on a synthetic sequence, on the GPU.  Frames are keypoints + 256-bit descriptors: feature detection
is the caller's (the reference uses OpenCV's ORB for it)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import slam  # noqa: E402  (examples/slam.py)
from slam_amd import synth  # noqa: E402

seq = synth.make_sequence(3, n_landmarks=1500, n_clutter=300, seed=7)
frames = [slam.Frame(*f) for f in seq["frames"]]
driver = slam.Slam(seq["K"])
driver.find_initial_pose(frames[0], frames[1])          # warm-up: first call allocates the context's buffers
t0 = time.perf_counter()
driver.find_initial_pose(frames[0], frames[1])
t1 = time.perf_counter()
driver.find_next_pose(frames[1], frames[2])
t2 = time.perf_counter()
print(f"find_initial_pose {1e3 * (t1 - t0):.2f} ms, find_next_pose {1e3 * (t2 - t1):.2f} ms "
      f"({len(frames[0].keypoints)} / {len(frames[1].keypoints)} / {len(frames[2].keypoints)} keypoints)")
for k, (est, true) in enumerate(zip(driver.camera_poses, seq["poses"])):
    print(f"frame {k}: camera centre {np.round(np.linalg.inv(est)[:3, 3], 3)}  (generator: {np.round(np.linalg.inv(true)[:3, 3], 3)})")
