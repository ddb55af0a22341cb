#!/usr/bin/env python3
"""Where the time of a short burst goes: K prepared RANSAC calls after an idle device, timed as
enqueue (the K library calls return) / drain (device synchronise returns), median over many bursts.
    burstprobe.py [K [LANES [spin]]]     spin: the device is put into spin-wait scheduling before anything else uses it"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    spin = len(sys.argv) > 3 and sys.argv[3] == "spin"
    import torch
    if spin:
        hip = C.CDLL("libamdhip64.so")
        rc = hip.hipSetDeviceFlags(1)      # hipDeviceScheduleSpin
        print("hipSetDeviceFlags(spin) ->", rc)
    from slam_amd import _lib, pipeline, synth
    ctx = _lib.Context(0)
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    N, H = 4096, 2048
    pair = synth.make_pair(N, 0.3, seed=2)
    n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
    n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
    d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    ctx.check(lib.slam_set_concurrency(ctx.handle, lanes))
    recs = [torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev) for _ in range(lanes)]
    masks = [torch.zeros(N, dtype=torch.uint8, device=dev) for _ in range(lanes)]
    jobs = []
    for k in range(lanes):
        j = C.c_void_p()
        ctx.check(lib.slam_ransac_job_create(ctx.handle, d_p0.data_ptr(), d_p1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON,
                                             _lib.SEED_BASE, 0, H, _lib.DEFAULT_THRESHOLD, recs[k].data_ptr(),
                                             masks[k].data_ptr(), None, None, None, C.byref(j)))
        jobs.append(j)
    run = lib.slam_job_run
    seq = [jobs[i % lanes] for i in range(K)]
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for j in seq:
            run(j)
        lib.slam_synchronize(ctx.handle)
    for how in ("torch.cuda.synchronize", "slam_synchronize"):
        enq, drain = [], []
        for _ in range(200):
            lib.slam_synchronize(ctx.handle)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in seq:
                run(j)
            t1 = time.perf_counter()
            if how == "torch.cuda.synchronize":
                torch.cuda.synchronize()
            else:
                lib.slam_synchronize(ctx.handle)
            t2 = time.perf_counter()
            enq.append(t1 - t0)
            drain.append(t2 - t1)
        e, d = np.median(enq) * 1e6, np.median(drain) * 1e6
        print(f"K={K} lanes={lanes} {how}: enqueue {e:.1f} us ({e / K:.2f}/step), drain {d:.1f} us, total {(e + d) / K:.2f} us/step")
    # an empty synchronise, for scale
    t = []
    for _ in range(200):
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    print(f"synchronise of an idle device: {np.median(t) * 1e6:.1f} us")


if __name__ == "__main__":
    main()
