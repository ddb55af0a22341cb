"""Does an initialised RCCL process group change the cost of the library's launches?  (single rank)"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import torch.distributed as dist

mode = sys.argv[1]          # none | before | after
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
from oracle import slam_oracle as so
from slam_amd import _lib, synth

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if mode == "before":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
ctx = _lib.Context(0)
lib = ctx.lib
if mode == "after":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
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
K = 3000
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        fn(h, a0, a1, 4096, None, 0, 0x1337BEEF, 0, 2048, 0.005, rp[i % 3], None, None, None, None)
    t1 = time.perf_counter()
    lib.slam_synchronize(h)
    t2 = time.perf_counter()
    print(f"{mode}: enqueue {1e6*(t1-t0)/K:.2f} us/call, drained after {1e6*(t2-t0)/K:.2f} us/call", flush=True)
if mode != "none":
    dist.destroy_process_group()
