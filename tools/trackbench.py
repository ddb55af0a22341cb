"""Timing of the tracking step (device-resident inputs): whole call and its kernels."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from slam_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("-M", type=int, default=300)
ap.add_argument("-N", type=int, default=1000)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()

ctx = _lib.default_context()
lib = ctx.lib
dev = torch.device("cuda", ctx.device)
d = synth.make_track(a.M, a.N, 0.3, 5)
t = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in ("prev_known", "next_known", "X_known", "prev_all", "next_all")}
pose = torch.empty(16, dtype=torch.float64, device=dev)
X = torch.empty((a.N, 3), dtype=torch.float64, device=dev)
best = torch.empty(3, dtype=torch.float64, device=dev)
prev, K = _lib.f64(d["previous_pose"]), _lib.f64(d["K"])
torch.cuda.synchronize()


def step():
    ctx.check(lib.slam_track_pose_dev(ctx.handle, _lib.ptr(prev), t["prev_known"].data_ptr(), t["next_known"].data_ptr(),
                                      t["X_known"].data_ptr(), a.M, t["prev_all"].data_ptr(), t["next_all"].data_ptr(),
                                      a.N, _lib.ptr(K), a.iters, pose.data_ptr(), X.data_ptr(), best.data_ptr()))


for _ in range(5):
    step()
lib.slam_synchronize(ctx.handle)
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
lib.slam_synchronize(ctx.handle)
dt = (time.perf_counter() - t0) / a.steps
lib.slam_profile_enable(ctx.handle, 1)
lib.slam_profile_reset(ctx.handle)
for _ in range(a.steps):
    step()
import ctypes as C
out = {}
for name, k in (("resect", _lib.K_RESECT), ("score", _lib.K_TRACK_SCORE), ("triangulate", _lib.K_TRIANGULATE)):
    ms, n = C.c_double(), C.c_int64()
    lib.slam_profile_read(ctx.handle, k, C.byref(ms), C.byref(n))
    out[name] = 1e3 * ms.value / max(n.value, 1)
lib.slam_profile_enable(ctx.handle, 0)
H = int(lib.slam_num_tracking_hypotheses(a.M, a.iters))
print(f"M={a.M} N={a.N} H={H}: call {dt*1e6:.1f} us | " + " ".join(f"{k} {v:.1f} us" for k, v in out.items()), "| best", best.cpu().numpy())
