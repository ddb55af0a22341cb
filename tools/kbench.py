#!/usr/bin/env python3
"""Per-kernel timings of one RANSAC configuration (development aid; prints event-based averages in microseconds)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-N", type=int, default=4096)
    ap.add_argument("-H", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--idx", action="store_true", help="hand in pre-drawn indices")
    ap.add_argument("--sampler", type=int, default=0)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--f64", action="store_true", help="keep long calls on the float64 scoring kernel")
    args = ap.parse_args()
    if args.f64:
        os.environ["SLAM_AMD_F32_COUNTS"] = "0"   # read when the context is created
    import torch
    from slam_amd import _lib, pipeline, synth
    ctx = _lib.Context(0)
    lib = ctx.lib
    dev = torch.device("cuda", 0)
    pair = synth.make_pair(args.N, args.outliers, seed=2)
    n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
    n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
    d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    d_rec = torch.zeros(12, dtype=torch.float64, device=dev)
    d_mask = torch.zeros(args.N, dtype=torch.uint8, device=dev)
    d_idx = None
    if args.idx:
        idx = np.empty((args.H, 8), dtype=np.int32)
        ctx.check(lib.slam_sample_indices(ctx.handle, args.sampler, _lib.SEED_BASE, 0, args.H, args.N, 8, _lib.ptr(idx)))
        d_idx = torch.from_numpy(idx).to(dev)
    torch.cuda.synchronize()

    def step():
        ctx.check(lib.slam_ransac_f_dev(ctx.handle, d_p0.data_ptr(), d_p1.data_ptr(), args.N,
                                        d_idx.data_ptr() if d_idx is not None else None, args.sampler, _lib.SEED_BASE, 0,
                                        args.H, 0.005, d_rec.data_ptr(), d_mask.data_ptr(), None, None, None))
    for _ in range(50):
        step()
    ctx.check(lib.slam_synchronize(ctx.handle))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.check(lib.slam_synchronize(ctx.handle))
    dt = (time.perf_counter() - t0) / args.steps
    ctx.check(lib.slam_profile_enable(ctx.handle, 1))
    ctx.check(lib.slam_profile_reset(ctx.handle))
    for _ in range(args.steps):
        step()
    res = {}
    for name, kid in (("solve", 0), ("score", 1), ("select", 2)):
        ms, n = C.c_double(0), C.c_int64(0)
        ctx.check(lib.slam_profile_read(ctx.handle, kid, C.byref(ms), C.byref(n)))
        res[name] = ms.value / max(n.value, 1) * 1e3
    tf = 18.0 * args.N * args.H / (res["score"] * 1e-6) / 1e12
    print(f"N={args.N} H={args.H} idx={args.idx} sampler={args.sampler}: step {dt*1e6:.2f} us | "
          + " ".join(f"{k} {v:.2f}" for k, v in res.items()) + f" | score {tf:.1f} TFLOP/s ({tf/78.6*100:.0f}%)")


if __name__ == "__main__":
    main()
