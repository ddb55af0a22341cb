#!/usr/bin/env python3
"""Time of the descriptor matcher through its host entry point (numpy in, numpy out), 2000 keypoints per image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import matching, synth
d = synth.make_feature_pair(1500, n_clutter=500, seed=3)
args = (d["kp0"], d["des0"], d["kp1"], d["des1"])
for _ in range(20):
    m0, m1 = matching.matches_between_keypoints(*args)
t0 = time.perf_counter()
for _ in range(300):
    matching.matches_between_keypoints(*args)
print(f"matcher, host buffers, 2000 x 2000 keypoints: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call, {len(m0)} matches")
