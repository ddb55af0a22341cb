#!/usr/bin/env python3
"""One tracking case of test_gpu_fuzz.py in detail (not collected by pytest): python tests/fuzz_track_case.py SEED CASE M ITER WRONG EXTRA"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co
from slam_amd import _lib, logger, synth, triangulation
seed, case, M, iters = (int(v) for v in sys.argv[1:5])
frac, extra = float(sys.argv[5]), int(sys.argv[6])
logger.set_log_level("WARN")
ctx = _lib.default_context()
d = synth.make_track(M, M + extra, frac, 5000 * seed + case)
r = triangulation.track_pose_detailed(d["previous_pose"], d["prev_known"], d["next_known"], d["X_known"], d["prev_all"],
                                      d["next_all"], d["K"], iters, ctx)
H = len(r["counts"])
want = co.tracking_ransac(d["previous_pose"], d["prev_known"], d["next_known"], d["X_known"], d["K"], co.sample(M, H, k=6))
np.set_printoptions(linewidth=200, precision=6)
print("H", H, "counts gpu", r["counts"]); print("counts cpu", want["counts"])
print("errs rel", np.abs(r["errs"] / want["errs"] - 1))
print("pose rel", [float(np.max(np.abs(r["poses"][h] - want["poses"][h])) / np.max(np.abs(want["poses"][h]))) for h in range(H)])
print("best", r["best_index"], want["best_index"])
