"""Host-side cost of one slam_ransac_f_dev call (three streams in flight), with and without the selection launch."""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from oracle import slam_oracle as so
from slam_amd import _lib, synth

ctx = _lib.default_context()
lib = ctx.lib
dev = torch.device("cuda", ctx.device)
d = synth.make_pair(4096, 0.3, seed=2)
n0, _ = so.normalize_points(d["pts0"])
n1, _ = so.normalize_points(d["pts1"])
p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
recs = [torch.zeros(12, dtype=torch.float64, device=dev) for _ in range(4)]
torch.cuda.synchronize()
lib.slam_set_concurrency(ctx.handle, int(os.environ.get("LANES", "3")))
fn, h = lib.slam_ransac_f_dev, ctx.handle
a0, a1 = p0.data_ptr(), p1.data_ptr()
rp = [r.data_ptr() for r in recs]
K = 4000
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        fn(h, a0, a1, 4096, None, 0, 0x1337BEEF, 0, 2048, 0.005, rp[i % 4], None, None, None, None)
    t1 = time.perf_counter()
    lib.slam_synchronize(h)
    t2 = time.perf_counter()
    print(f"enqueue {1e6*(t1-t0)/K:.2f} us/call, drained after {1e6*(t2-t0)/K:.2f} us/call")
