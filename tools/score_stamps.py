#!/usr/bin/env python3
"""Per-wave timeline of the large-H scoring kernel from the diagnostic (stamped) build (make -C slam_amd/csrc stamps).
usage: score_stamps.py [N] [H]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libslam_amd_stamps.so")
import torch  # noqa: E402
from slam_amd import pipeline, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
pair = synth.make_pair(N, 0.3, seed=2)
n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
d_rec = torch.zeros(12, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(20):
    ctx.check(lib.slam_ransac_f_dev(ctx.handle, d_p0.data_ptr(), d_p1.data_ptr(), N, None, 1, 3, 0, H, 0.005,
                                    d_rec.data_ptr(), None, None, None, None))
full = np.zeros(4096 * 8, dtype=np.int64)
lib.slam_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.check(lib.slam_debug_stamps(ctx.handle, full.ctypes.data, 4096 * 8))
st = full.reshape(4096, 8).astype(np.float64)
st = st[st[:, 0] > 0]
print(f"N={N} H={H}: {len(st)} waves stamped")
t0 = st[:, 0].min()
q = lambda x: np.quantile(x, [0.0, 0.5, 0.9, 1.0]).round(0).tolist()
print("wave start after first start (cycles) min/med/90%/max:", q(st[:, 0] - t0))
print("staging (entry -> chunk in LDS):", q(st[:, 1] - st[:, 0]))
print("group 1:", q(st[:, 2] - st[:, 1]), " group 2:", q(st[:, 3] - st[:, 2]), " group 3:", q(st[:, 4] - st[:, 3]))
print("whole wave:", q(st[:, 5] - st[:, 0]))
print("first start to last end, shader cycles:", st[:, 5].max() - t0)
wall = (st[:, 7].max() - st[:, 6].min()) * 10.0   # 100 MHz constant clock -> ns
print(f"first start to last end, constant clock: {wall / 1e3:.1f} us")
print("wave starts after the first, us (constant clock) min/med/90%/max:", [round(x, 2) for x in np.quantile((st[:, 6] - st[:, 6].min()) * 0.01, [0, 0.5, 0.9, 1.0])])
print("wave ends before the last, us min/med/90%/max:", [round(x, 2) for x in np.quantile((st[:, 7].max() - st[:, 7]) * 0.01, [0, 0.5, 0.9, 1.0])])
live = st[:, 5] - st[:, 0] > 50000      # shader clock while the kernel ran: shader cycles of a wave / its time on the constant clock
mhz = (st[live, 5] - st[live, 0]) / ((st[live, 7] - st[live, 6]) * 10.0) * 1e3
print("shader clock per wave, MHz min/med/max:", q(mhz))
short = st[~live]
print(f"{len(short)} waves lived under 50k cycles; their rows:", np.flatnonzero(~live)[:20].tolist())
