#!/usr/bin/env python3
"""How the workgroups of back-to-back RANSAC calls on several streams share the GPU: lifetimes of every workgroup of
the fused kernel over a few hundred calls (log-only diagnostic build: make -C slam_amd/csrc wglog; constant 100 MHz
clock), grouped by launch.
usage: wg_timeline.py [LANES]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libslam_amd_wglog.so")
import torch  # noqa: E402
from slam_amd import pipeline, synth  # noqa: E402

LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, H = 4096, 2048
ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
pair = synth.make_pair(N, 0.3, seed=2)
n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
recs = [torch.zeros(12, dtype=torch.float64, device=dev) for _ in range(LANES)]
masks = [torch.zeros(N, dtype=torch.uint8, device=dev) for _ in range(LANES)]
ctx.check(lib.slam_set_concurrency(ctx.handle, LANES))
jobs = []
for k in range(LANES):
    j = C.c_void_p()
    ctx.check(lib.slam_ransac_job_create(ctx.handle, d_p0.data_ptr(), d_p1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE,
                                         0, H, _lib.DEFAULT_THRESHOLD, recs[k].data_ptr(), masks[k].data_ptr(), None, None, None,
                                         C.byref(j)))
    jobs.append(j)
lib.slam_debug_wg_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
for i in range(30000):
    lib.slam_job_run(jobs[i % LANES])
ctx.check(lib.slam_synchronize(ctx.handle))
ctx.check(lib.slam_debug_wg_log(ctx.handle, 1, None, 0, None))
STEPS = 400
for i in range(STEPS):
    lib.slam_job_run(jobs[i % LANES])
ctx.check(lib.slam_synchronize(ctx.handle))
out = np.zeros((1 << 17, 4), dtype=np.int64)
cnt = C.c_int(0)
ctx.check(lib.slam_debug_wg_log(ctx.handle, 0, out.ctypes.data, len(out), C.byref(cnt)))
L = out[:cnt.value]
tag, blk = L[:, 0], L[:, 1] >> 32
t1 = L[:, 3].astype(np.float64) / 100.0                    # us, constant clock
t0 = t1 - L[:, 2].astype(np.float64) / 2350.0              # lifetime in shader cycles at about 2.35 GHz
xcc, cu = (L[:, 1] >> 16) & 0xf, L[:, 1] & 0xffff
t_base = t0.min()
t0 -= t_base
t1 -= t_base
print(f"{cnt.value} workgroups logged over {t1.max():.0f} us = {t1.max() / STEPS:.2f} us per step with {LANES} stream(s)")
print(f"workgroup lifetime: median {np.median(t1 - t0):.2f} us, 90% {np.quantile(t1 - t0, 0.9):.2f}, max {(t1 - t0).max():.2f}")
# launches: per stream tag, split the workgroups by time into groups of 256
launches = []
for tg in np.unique(tag):
    m = np.flatnonzero(tag == tg)
    order = m[np.argsort(t0[m])]
    for g in range(0, len(order) - 255, 256):
        sel = order[g:g + 256]
        launches.append((t0[sel].min(), t0[sel].max(), t1[sel].min(), t1[sel].max(), tg))
launches.sort()
A = np.array([l[:4] for l in launches])
print(f"{len(A)} launches; first-to-last workgroup START within a launch: median {np.median(A[:, 1] - A[:, 0]):.2f} us; "
      f"first start to last end: median {np.median(A[:, 3] - A[:, 0]):.2f} us; interval between launches {np.median(np.diff(A[:, 0])):.2f} us")
# how many launches are alive (first start .. last end) at a launch's first start
alive = [int(np.sum((A[:i, 0] <= A[i, 0]) & (A[:i, 3] > A[i, 0]))) for i in range(len(A))]
print("launches still running when a launch's first workgroup starts:", np.bincount(alive[20:]))
# workgroups resident at the same time on the chip, sampled at workgroup starts
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
res = np.cumsum(ev[:, 1])
mid = (ev[:, 0] > t1.max() * 0.1) & (ev[:, 0] < t1.max() * 0.9)
dt = np.diff(ev[:, 0])
w = dt[mid[:-1]]
r = res[:-1][mid[:-1]]
print(f"workgroups resident on the chip, time-weighted: mean {np.sum(r * w) / np.sum(w):.0f}, of 512 slots; "
      f"share of time with more than 256 resident: {np.sum(w[r > 256]) / np.sum(w):.2f}, with fewer than 128: {np.sum(w[r < 128]) / np.sum(w):.2f}")
# per CU: does a second workgroup start on a CU while another is alive there?
key = xcc * 65536 + cu
print("distinct (xcc, hw id) keys:", len(np.unique(key)))
