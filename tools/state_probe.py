#!/usr/bin/env python3
"""Which of the two step rates a context lands in: contexts created one after the other in ONE process, each timed
over bursts of 2000 headline steps (four calls in flight).  Tells whether the rate is a property of the process or
of the context's streams.     usage: state_probe.py [contexts] [keep]   (keep=1: earlier contexts stay alive)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import _lib, pipeline, synth

import subprocess
import tempfile

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
keep = len(sys.argv) > 2 and sys.argv[2] == "1"
# clocks, power and temperature are sampled by a process of its own (no share in this interpreter's lock)
sample_file = tempfile.NamedTemporaryFile(suffix=".txt", delete=False).name
sampler = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "clock_sampler.py"), sample_file])
marks = []
dev = torch.device("cuda", 0)
pair = synth.make_pair(4096, 0.3, seed=2)
K, lanes = 2000, 4
kept = []
for c in range(n_ctx):
    ctx = _lib.Context(0)
    lib = ctx.lib
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
    while time.perf_counter() - t_pre < 0.25:
        for j in seq[:200]:
            run(j)
        lib.slam_synchronize(ctx.handle)
    ts = []
    for rep in range(8):
        lib.slam_synchronize(ctx.handle)
        m0 = time.monotonic_ns()
        t0 = time.perf_counter()
        for j in seq:
            run(j)
        lib.slam_synchronize(ctx.handle)
        ts.append((time.perf_counter() - t0) / K * 1e6)
        marks.append((c, rep, m0, time.monotonic_ns(), ts[-1]))
    print(f"context {c}: us/step " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    if keep:
        kept.append((ctx, p0, p1, recs, masks, jobs))
    else:
        for j in jobs:
            lib.slam_job_destroy(j)
        ctx.close()

sampler.terminate()
sampler.wait()
samples = np.loadtxt(sample_file, ndmin=2)
for c, rep, m0, m1, us in marks:
    sel = samples[(samples[:, 0] >= m0) & (samples[:, 0] <= m1)]
    if len(sel):
        print(f"context {c} burst {rep}: {us:.2f} us/step, {len(sel)} samples: sclk {sel[:, 1].mean():.0f} MHz "
              f"(min {sel[:, 1].min():.0f}, max {sel[:, 1].max():.0f}), power {sel[:, 2].mean() / 1e6:.0f} W, temp {sel[:, 3].mean() / 1e3:.0f} C")
os.unlink(sample_file)
