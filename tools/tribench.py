"""Bulk triangulation (four candidates per point) on device-resident points: kernel time per launch."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=1 << 22)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--outliers", type=float, default=0.0, help="fraction of wrong matches in the tiled set")
a = ap.parse_args()
ctx = _lib.default_context()
lib = ctx.lib
dev = torch.device("cuda", ctx.device)
big = synth.make_pair(1 << 16, a.outliers, seed=5)
reps = max(1, a.n // (1 << 16))
p0 = torch.from_numpy(big["pts0"]).to(dev).repeat(reps, 1)[:a.n].contiguous()
p1 = torch.from_numpy(big["pts1"]).to(dev).repeat(reps, 1)[:a.n].contiguous()
torch.manual_seed(5)
p1 += torch.rand_like(p1) * 0.25
E = np.ascontiguousarray(synth.essential_from_motion(big["R"], big["t"]))
K = np.ascontiguousarray(big["K"])
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib.slam_profile_enable(ctx.handle, 1)
for i in range(a.steps + 1):
    if i == 1:
        lib.slam_profile_reset(ctx.handle)
    ctx.check(lib.slam_pose_candidates_dev(ctx.handle, _lib.ptr(E), _lib.ptr(K), p0.data_ptr(), p1.data_ptr(), a.n, cnt.data_ptr()))
ms, n = C.c_double(), C.c_int64()
lib.slam_profile_read(ctx.handle, _lib.K_TRIANGULATE, C.byref(ms), C.byref(n))
t = ms.value / n.value
print(f"n={a.n}: k_triangulate4 {t:.3f} ms, {a.n/t/1e6:.2f} G points/s, {4*a.n/t/1e6:.2f} G candidate triangulations/s, counts {cnt.cpu().tolist()}")
# plain triangulation with one camera pair (slam_triangulate_dev), the true motion's cameras
from slam_amd import util
P0 = np.ascontiguousarray(K @ np.hstack([np.eye(3), np.zeros((3, 1))]))
P1 = np.ascontiguousarray(K @ util.rotation_translation_to_pose(big["R"], big["t"])[:3])
X = torch.empty((a.n, 3), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for i in range(a.steps + 1):
    if i == 1:
        lib.slam_profile_reset(ctx.handle)
    ctx.check(lib.slam_triangulate_dev(ctx.handle, _lib.ptr(P0), _lib.ptr(P1), p0.data_ptr(), p1.data_ptr(), a.n, X.data_ptr()))
lib.slam_profile_read(ctx.handle, _lib.K_TRIANGULATE, C.byref(ms), C.byref(n))
t = ms.value / n.value
print(f"n={a.n}: k_triangulate (one camera pair) {t:.3f} ms, {a.n/t/1e6:.2f} G points/s")
# the same with the cameras of the four candidates of E (src/epipolar.py:97-107): how much the data decide the time
U, _, Vt = np.linalg.svd(E)
W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
for name, R in (("U W V^T", U @ W @ Vt), ("U W^T V^T", U @ W.T @ Vt)):
    R = R * np.sign(np.linalg.det(R))
    for sgn in (1.0, -1.0):
        P1c = np.ascontiguousarray(K @ util.rotation_translation_to_pose(R, sgn * U[:, 2])[:3])
        for i in range(a.steps + 1):
            if i == 1:
                lib.slam_profile_reset(ctx.handle)
            ctx.check(lib.slam_triangulate_dev(ctx.handle, _lib.ptr(P0), _lib.ptr(P1c), p0.data_ptr(), p1.data_ptr(), a.n, X.data_ptr()))
        lib.slam_profile_read(ctx.handle, _lib.K_TRIANGULATE, C.byref(ms), C.byref(n))
        t = ms.value / n.value
        print(f"  candidate R = {name}, t = {sgn:+.0f} u3: {t:.3f} ms")
