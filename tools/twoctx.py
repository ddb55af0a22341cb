#!/usr/bin/env python3
"""Development probe: throughput of back-to-back RANSAC calls when they alternate between K contexts (streams)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import _lib, pipeline, synth
N, H = 4096, 2048
dev = torch.device("cuda", 0)
for K in (1, 2, 3, 4):
    ctxs = [_lib.Context(0) for _ in range(K)]
    lib = ctxs[0].lib
    pair = synth.make_pair(N, 0.3, seed=2)
    n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctxs[0])
    n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctxs[0])
    d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    recs = [torch.zeros(12, dtype=torch.float64, device=dev) for _ in range(K)]
    masks = [torch.zeros(N, dtype=torch.uint8, device=dev) for _ in range(K)]
    torch.cuda.synchronize()
    def step(i):
        c = ctxs[i % K]
        c.check(lib.slam_ransac_f_dev(c.handle, d_p0.data_ptr(), d_p1.data_ptr(), N, None, 0, _lib.SEED_BASE, 0, H, 0.005,
                                      recs[i % K].data_ptr(), masks[i % K].data_ptr(), None, None, None))
    for i in range(200):
        step(i)
    for c in ctxs:
        c.check(lib.slam_synchronize(c.handle))
    t0 = time.perf_counter()
    S = 4000
    for i in range(S):
        step(i)
    for c in ctxs:
        c.check(lib.slam_synchronize(c.handle))
    dt = (time.perf_counter() - t0) / S
    print(f"{K} stream(s): {dt*1e6:.2f} us per step, {H/dt/1e6:.1f} M hypotheses/s; winner {int(recs[0][2].item())}")
    for c in ctxs:
        c.close()
