#!/usr/bin/env python3
"""Host side of the headline step: time to ENQUEUE a burst of prepared calls (slam_job_run in a Python loop) against
the time until the burst has drained, per step.  If the two agree the host, not the GPU, sets the step rate.
usage: enqueue_probe.py [lanes] [steps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import _lib, pipeline, synth

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
pair = synth.make_pair(4096, 0.3, seed=2)
n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
ctx.check(lib.slam_set_concurrency(ctx.handle, lanes))
recs = [torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev) for _ in range(lanes)]
masks = [torch.zeros(4096, dtype=torch.uint8, device=dev) for _ in range(lanes)]
jobs = []
for k in range(lanes):
    j = C.c_void_p()
    ctx.check(lib.slam_ransac_job_create(ctx.handle, p0.data_ptr(), p1.data_ptr(), 4096, None, _lib.SAMPLER_CPYTHON,
                                         _lib.SEED_BASE, 0, 2048, _lib.DEFAULT_THRESHOLD, recs[k].data_ptr(),
                                         masks[k].data_ptr(), None, None, None, C.byref(j)))
    jobs.append(j)
run = lib.slam_job_run
seq = [jobs[i % lanes] for i in range(K)]
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.3:
    for j in seq[:200]:
        run(j)
    lib.slam_synchronize(ctx.handle)
for rep in range(6):
    lib.slam_synchronize(ctx.handle)
    t0 = time.perf_counter()
    for j in seq:
        run(j)
    t1 = time.perf_counter()
    lib.slam_synchronize(ctx.handle)
    t2 = time.perf_counter()
    print(f"lanes {lanes}: enqueue {1e6 * (t1 - t0) / K:.2f} us/step, drained after {1e6 * (t2 - t0) / K:.2f} us/step")
