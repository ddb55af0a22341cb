"""Which HIP runtime copies end up in the process, and in which initialisation order do both see the GPU?"""
import sys

sys.path.insert(0, ".")
order = sys.argv[1]


def maps():
    out = set()
    for ln in open("/proc/self/maps"):
        if "amdhip64" in ln or "hsa-runtime" in ln:
            out.add(ln.split()[-1])
    return sorted(out)


def mine():
    from slam_amd import _lib
    c = _lib.Context(0)
    print("libslam_amd context ok", flush=True)
    return c


def torch_side():
    import torch
    print("torch sees", torch.cuda.device_count(), torch.cuda.is_available(), flush=True)
    if torch.cuda.is_available():
        torch.zeros(4, device="cuda").sum().item()
        print("torch kernel ok", flush=True)


if order == "lib_first":
    c = mine(); torch_side()
else:
    torch_side(); c = mine()
print("\n".join(maps()))
