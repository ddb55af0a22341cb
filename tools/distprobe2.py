"""Bursts of library launches in a process with an RCCL process group: after a barrier collective, and without."""
import os
import sys
import time

sys.path.insert(0, ".")
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
from oracle import slam_oracle as so
from slam_amd import _lib, synth

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
if os.environ.get("EARLY_BARRIER"):
    dist.barrier()
    torch.cuda.synchronize()
ctx = _lib.Context(0)
lib = ctx.lib
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
print("no collective yet:", [round(burst(300), 2) for _ in range(3)], flush=True)
t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("first barrier ms", (time.perf_counter() - t0) * 1e3)
print("after one barrier:", [round(burst(300), 2) for _ in range(3)], flush=True)
for _ in range(3):
    t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
    print(f"barrier {tb:.3f} ms then burst {burst(300):.2f} us/step, next burst {burst(300):.2f}", flush=True)
t = torch.ones(1, device=dev)
dist.all_reduce(t); torch.cuda.synchronize()
print("after all_reduce:", [round(burst(300), 2) for _ in range(3)], flush=True)
dist.destroy_process_group()
