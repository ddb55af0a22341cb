#!/usr/bin/env python3
"""Where the time of the host-buffer RANSAC call (slam_ransac_f: upload, kernels, download, one wait) goes: the call
timed for variations that leave parts out (no mask, few hypotheses, few matches), against the device-resident call
with a wait.     usage: hostpath_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import _lib, pipeline, synth

ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)


def timed(fn, reps=400):
    for _ in range(50):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for N, H in ((4096, 2048), (4096, 8), (64, 8), (2048, 1000)):
    pair = synth.make_pair(N, 0.3, seed=2)
    n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
    n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
    rec = np.empty(_lib.REC_LEN)
    mask = np.empty(N, dtype=np.uint8)
    d0, d1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    d_rec = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    d_mask = torch.zeros(N, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def host(with_mask):
        ctx.check(lib.slam_ransac_f(ctx.handle, _lib.ptr(n0), _lib.ptr(n1), N, None, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE, 0, H,
                                    _lib.DEFAULT_THRESHOLD, _lib.ptr(rec), _lib.ptr(mask) if with_mask else None, None, None, None))

    def resident(with_mask):
        ctx.check(lib.slam_ransac_f_dev(ctx.handle, d0.data_ptr(), d1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE, 0, H,
                                        _lib.DEFAULT_THRESHOLD, d_rec.data_ptr(), d_mask.data_ptr() if with_mask else None, None, None, None))
        ctx.check(lib.slam_synchronize(ctx.handle))

    print(f"N={N} H={H}: host buffers with mask {timed(lambda: host(True)):.1f} us, without {timed(lambda: host(False)):.1f} us; "
          f"resident + wait with mask {timed(lambda: resident(True)):.1f} us, without {timed(lambda: resident(False)):.1f} us")
