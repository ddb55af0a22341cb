"""slam_init_pairs on 100 pairs: the library call alone, with pageable and with pinned host buffers."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from slam_amd import _lib, synth
from slam_amd._lib import ptr

ctx = _lib.default_context()
lib = ctx.lib
P, N, IT = 100, 2048, 1000
data = [synth.make_pair(N, 0.3, seed=100 + i) for i in range(P)]
K = _lib.f64(data[0]["K"])
offsets = np.arange(P + 1, dtype=np.int64) * N
total = P * N


def run(pin):
    mk = (lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()) if pin else (lambda shape, dt: torch.empty(shape, dtype=dt).numpy())
    p0 = mk((total, 2), torch.float64); p1 = mk((total, 2), torch.float64)
    p0[:] = np.concatenate([d["pts0"] for d in data]); p1[:] = np.concatenate([d["pts1"] for d in data])
    X = mk((total, 3), torch.float64); good = mk((total,), torch.uint8)
    F = np.empty((P, 9)); E = np.empty((P, 9)); pose = np.empty((P, 16)); rec = np.empty((P, 12))
    c4 = np.empty((P, 4), dtype=np.int32); win = np.empty(P, dtype=np.int32)
    def call():
        ctx.check(lib.slam_init_pairs(ctx.handle, ptr(p0), ptr(p1), ptr(offsets), P, ptr(K), IT, ptr(F), ptr(E), ptr(pose), ptr(X),
                                      ptr(good), ptr(rec), ptr(c4), ptr(win)))
    call()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    dt = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.check(lib.slam_init_pairs(ctx.handle, ptr(p0), ptr(p1), ptr(offsets), P, ptr(K), IT, ptr(F), ptr(E), ptr(pose), None,
                                      None, ptr(rec), ptr(c4), ptr(win)))
    dt2 = (time.perf_counter() - t0) / 10
    print(f"{'pinned' if pin else 'pageable'}: {dt*1e3:.3f} ms per batch with points and flags downloaded, {dt2*1e3:.3f} ms without")


run(False)
run(True)
