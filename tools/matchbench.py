"""Kernel time of the descriptor matcher (device-resident inputs)."""
import argparse
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from slam_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("-q", type=int, default=4096)
ap.add_argument("-t", type=int, default=4096)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
ctx = _lib.default_context()
lib = ctx.lib
dev = torch.device("cuda", ctx.device)
rng = np.random.default_rng(11)
t = [torch.from_numpy(x).to(dev) for x in (rng.integers(0, 256, (a.q, 32), dtype=np.uint8), rng.uniform(0, 1000, (a.q, 2)),
                                           rng.integers(0, 256, (a.t, 32), dtype=np.uint8), rng.uniform(0, 1000, (a.t, 2)))]
m0 = torch.empty((a.q, 2), dtype=torch.float64, device=dev)
m1 = torch.empty((a.q, 2), dtype=torch.float64, device=dev)
n = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def step():
    ctx.check(lib.slam_match_points_dev(ctx.handle, t[0].data_ptr(), t[1].data_ptr(), a.q, t[2].data_ptr(), t[3].data_ptr(), a.t,
                                        m0.data_ptr(), m1.data_ptr(), n.data_ptr(), None, None))


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
ms, k = C.c_double(), C.c_int64()
lib.slam_profile_read(ctx.handle, _lib.K_MATCH, C.byref(ms), C.byref(k))
kus = 1e3 * ms.value / max(k.value, 1)
print(f"q={a.q} t={a.t}: call {dt*1e6:.1f} us, k_match_knn2 {kus:.1f} us, {a.q*a.t/kus/1e6:.3f} T pairs/s")
