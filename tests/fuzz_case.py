#!/usr/bin/env python3
"""One case of test_gpu_fuzz.py in detail (not collected by pytest): python tests/fuzz_case.py CASE N H OUTLIERS NOISE THR"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co, slam_oracle as so
from slam_amd import _lib, epipolar, logger, pipeline, synth
case, N, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frac, noise, thr = float(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
logger.set_log_level("WARN")
ctx = _lib.default_context()
d = synth.make_pair(N, frac, seed=1000 + case, noise_px=noise)
n0, _ = pipeline.normalize_points(d["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(d["pts1"], ctx=ctx)
g = epipolar.ransac_detailed(n0, n1, H, threshold=thr, per_hypothesis=True, ctx=ctx)
idx = co.sample(N, g["H"])
o = co.ransac(n0, n1, idx, thr=thr)
dev = np.abs(g["errsums"] - o["errsums"])
h = int(np.argmax(dev - 1e-9 * np.abs(o["errsums"])))
print("hypothesis", h, "count", g["counts"][h], o["counts"][h], "errsum", g["errsums"][h], o["errsums"][h])
Fg, Fo = g["F_all"][h], o["F"][h]
s = np.sign(np.sum(Fg * Fo))
print("F rel diff GPU vs C restatement", np.linalg.norm(Fg * s - Fo) / np.linalg.norm(Fo))
Fn = so.eight_point(n0[idx[h]], n1[idx[h]])
s2 = np.sign(np.sum(Fn * Fo))
print("F rel diff LAPACK vs C restatement", np.linalg.norm(Fn * s2 - Fo) / np.linalg.norm(Fo), " LAPACK vs GPU", np.linalg.norm(Fn * s2 - Fg * s) / np.linalg.norm(Fo))
A = so.design_matrix(n0[idx[h]], n1[idx[h]])
sv = np.linalg.svd(A, compute_uv=False)
print("singular values of the 8x9 system", sv)
print("singular values of the unprojected f:", np.linalg.svd(np.linalg.svd(A)[2][-1].reshape(3, 3), compute_uv=False))
print("inlier flags of the sample", d["inlier"][idx[h]])
