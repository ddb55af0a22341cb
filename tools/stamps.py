#!/usr/bin/env python3
"""Phase breakdown of the fused RANSAC kernel from the diagnostic (stamped) build: SLAM_AMD_LIB points at it."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("SLAM_STAMPS_LIB", "libslam_amd_stamps.so"))
import torch  # noqa: E402
from slam_amd import pipeline, synth  # noqa: E402

# usage: stamps.py [N [H [json-file tag [wide]]]]   (with a file name: the phase table is also appended to it as one JSON
# record; "wide": the context is told that several calls may be in flight, which selects sixteen hypotheses per workgroup
# at N > 2048 - the calls are still issued one at a time, so that one launch's stamps are read)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
JSON_OUT = sys.argv[3] if len(sys.argv) > 3 else None
TAG = sys.argv[4] if len(sys.argv) > 4 else "untagged"
WIDE = len(sys.argv) > 5 and sys.argv[5] == "wide"
ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
pair = synth.make_pair(N, 0.3, seed=2)
n0, _ = pipeline.normalize_points(pair["pts0"], ctx=ctx)
n1, _ = pipeline.normalize_points(pair["pts1"], ctx=ctx)
d_p0, d_p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
d_rec = torch.zeros(12, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
if WIDE:
    ctx.check(lib.slam_set_concurrency(ctx.handle, 4))
for _ in range(20):
    ctx.check(lib.slam_ransac_f_dev(ctx.handle, d_p0.data_ptr(), d_p1.data_ptr(), N, None, 0, _lib.SEED_BASE, 0, H, 0.005,
                                    d_rec.data_ptr(), None, None, None, None))
    if WIDE:
        ctx.check(lib.slam_synchronize(ctx.handle))
per_wg = 16 if WIDE and N > 2048 and (H + 15) // 16 >= 128 else 8
nb = min(4095, (H + per_wg - 1) // per_wg)
full = np.zeros(4096 * 8, dtype=np.int64)
lib.slam_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.check(lib.slam_debug_stamps(ctx.handle, full.ctypes.data, 4096 * 8))
out = full[:nb * 8]
raw = out.reshape(nb, 8).astype(np.float64)
# slots in program order: 0 start, 1 drawn, 7 null vector found, 2 solved, 3 past the barrier, 4 scored,
# 6 wave totals written, 5 past the last barrier
order = [0, 1, 7, 2, 3, 4, 6, 5]
st = raw[:, order]
t0 = st[:, 0].min()
names = ["sample", "null vec", "rank-2", "barrier", "score", "wave sum", "barrier"]
d = np.diff(st, axis=1)
print(f"N={N} H={H} blocks={nb}; cycles (shader clock) median over workgroups")
for i, nme in enumerate(names):
    q = np.quantile(d[:, i], [0.5, 0.9, 0.99])
    print(f"  {nme:8s} median {q[0]:9.0f}  90% {q[1]:9.0f}  99% {q[2]:9.0f}  max {d[:, i].max():9.0f}")
tot = st[:, -1] - st[:, 0]
if JSON_OUT:
    import json
    rec = {"tag": TAG, "library": os.path.basename(_lib.LIB_PATH), "build_hash": lib.slam_build_hash().decode(), "N": N, "H": H,
           "hypotheses_per_workgroup": per_wg, "workgroups": int(nb), "unit": "shader cycles, per workgroup (thread 0 stamps)",
           "phases": {nme: {"median": float(np.median(d[:, i])), "p90": float(np.quantile(d[:, i], 0.9)), "max": float(d[:, i].max())}
                      for i, nme in enumerate(["draw", "null vector", "projection", "barrier before scoring", "score", "wave sum", "barrier and fold"])},
           "workgroup_total": {"median": float(np.median(tot)), "max": float(tot.max())},
           "first_start_to_last_end": float(st[:, -1].max() - t0)}
    try:
        allrecs = json.load(open(JSON_OUT))
    except Exception:
        allrecs = []
    allrecs.append(rec)
    json.dump(allrecs, open(JSON_OUT, "w"), indent=1)
print(f"  workgroup total median {np.median(tot):.0f}  max {tot.max():.0f}; first start to last end {st[:, -1].max() - t0:.0f} cycles")
starts = np.sort(st[:, 0] - t0)
ends = np.sort(st[:, -1] - t0)
print("  workgroup starts after the first one, cycles: " + " ".join(f"{q:.0f}" for q in np.quantile(starts, [0.1, 0.5, 0.9, 1.0]))
      + "; ends: " + " ".join(f"{q:.0f}" for q in np.quantile(ends, [0.0, 0.1, 0.5, 0.9, 1.0])))
sel = full[4095 * 8:4095 * 8 + 6].astype(np.float64)
print("selection kernel (one workgroup), cycles: " + ", ".join(f"{n} {d:.0f}" for n, d in zip(
    ["loads+scan", "wave reduction", "barrier", "fold of 16", "F load + mask"], np.diff(sel))) + f"; total {sel[5] - sel[0]:.0f}")

# per-wave stamps of the score phase (rows 1024 + workgroup * 8 + wave): 0 loop entry, 1 loop end, 2 past the barrier, 3 folded
w = full[1024 * 8:(1024 + 8 * min(nb, 256)) * 8].reshape(-1, 8, 8).astype(np.float64)     # [workgroup][wave][slot]
if w[:, :, 0].min() > 0:
    ref = w[:, :, 0].min(axis=1, keepdims=True)
    for name, col in (("score loop entry after the first wave's", 0), ("score loop end", 1), ("past the barrier", 2), ("folded", 3)):
        med = np.median(w[:, :, col] - ref, axis=0)
        print(f"  per wave, {name}: " + " ".join(f"{x:6.0f}" for x in med))

# constant-clock stamps (build with STAMP_FLAGS=-DSLAM_WALL_STAMPS): when the workgroups of the last launch started and ended
wl = full[3072 * 8:(3072 + min(nb, 256)) * 8].reshape(-1, 8)[:, :2].astype(np.float64)
if wl.min() > 0:
    t0w = wl[:, 0].min()
    s_ns, e_ns = np.sort(wl[:, 0] - t0w) * 10.0, np.sort(wl[:, 1] - t0w) * 10.0
    print("constant clock, ns after the first workgroup's start: starts " + " ".join(f"{q:.0f}" for q in np.quantile(s_ns, [0.1, 0.5, 0.9, 1.0]))
          + "; ends " + " ".join(f"{q:.0f}" for q in np.quantile(e_ns, [0.0, 0.1, 0.5, 0.9, 1.0]))
          + f"; lifetime median {np.median(wl[:, 1] - wl[:, 0]) * 10:.0f} ns = {np.median(tot):.0f} shader cycles")
