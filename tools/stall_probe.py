#!/usr/bin/env python3
"""Per-call host time of the headline step's enqueue loop (four calls in flight): where the host waits.
For each burst: total per step, the enqueue loop per step, and the calls that took longer than 20 us (index: us).
usage: stall_probe.py [bursts]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import _lib, pipeline, synth

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
pair = synth.make_pair(4096, 0.3, seed=2)
K, lanes = 2000, 4
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
now = time.perf_counter_ns
plan = os.environ.get("PLAN", "").split()   # e.g. "2000 2000 sleep 2000 200 200 2000": burst lengths, or a 100 ms pause
for b in range(bursts if not plan else len(plan)):
    if plan:
        if plan[b] == "sleep":
            time.sleep(0.1)
            print("pause of 100 ms")
            continue
        K = int(plan[b])
        seq = [jobs[i % lanes] for i in range(K)]
    lib.slam_synchronize(ctx.handle)
    stamps = [0] * (K + 1)
    stamps[0] = now()
    i = 1
    for j in seq:
        run(j)
        stamps[i] = now()
        i += 1
    lib.slam_synchronize(ctx.handle)
    t_end = now()
    d = np.diff(np.array(stamps)) / 1e3
    slow = np.nonzero(d > 20.0)[0]
    print(f"burst {b}: {(t_end - stamps[0]) / K / 1e3:.2f} us/step in all, enqueue {(stamps[-1] - stamps[0]) / K / 1e3:.2f} us/step, "
          f"call median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} p99 {np.percentile(d, 99):.2f} max {d.max():.1f} us; "
          f"{len(slow)} calls over 20 us taking {d[slow].sum():.0f} us: " + " ".join(f"{i}:{d[i]:.0f}" for i in slow[:12]), flush=True)
