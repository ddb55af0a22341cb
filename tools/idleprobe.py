"""Does a short idle gap before a burst of launches slow the burst down?"""
import sys
import time

sys.path.insert(0, ".")
import torch

from oracle import slam_oracle as so
from slam_amd import _lib, synth

ctx = _lib.Context(0)
lib = ctx.lib
dev = torch.device("cuda", 0)
d = synth.make_pair(4096, 0.3, seed=2)
n0, _ = so.normalize_points(d["pts0"])
n1, _ = so.normalize_points(d["pts1"])
p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
recs = [torch.zeros(12, dtype=torch.float64, device=dev) for _ in range(3)]
torch.cuda.synchronize()
lib.slam_set_concurrency(ctx.handle, 3)
fn, h = lib.slam_ransac_f_dev, ctx.handle
a0, a1 = p0.data_ptr(), p1.data_ptr()
rp = [r.data_ptr() for r in recs]


def burst(k):
    t0 = time.perf_counter()
    for i in range(k):
        fn(h, a0, a1, 4096, None, 0, 0x1337BEEF, 0, 2048, 0.005, rp[i % 3], None, None, None, None)
    lib.slam_synchronize(h)
    return (time.perf_counter() - t0) / k * 1e6


burst(30000)
for gap_ms in (0, 0.2, 1, 2, 5, 20, 100):
    out = []
    for rep in range(3):
        burst(2000)
        time.sleep(gap_ms / 1000)
        out.append(burst(300))
    print(f"idle {gap_ms} ms -> 300-step burst {min(out):.2f} .. {max(out):.2f} us/step", flush=True)
