#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/prof_TAG/ into profiles/ (TAG_*) and rewrites
profiles/traffic.json, the file bench.py reads roofline.traffic from.     usage: collect_profiles.py TAG [--traffic-only]
(--traffic-only: what profile_round.sh runs on the GPU box between its counter passes and its bench lines, so that
the lines quote the counters of the build they ran)"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
traffic_only = "--traffic-only" in sys.argv[2:]
if not traffic_only:
    for name in ("bench.json", "bench_20steps.json", "bench_2000steps.json", "kernel_stats.csv", "overlap.json", "overlap.txt"):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))


def load(name):
    return json.load(open(os.path.join(src, name + ".json")))


def pick(d, sub):
    for k, v in d.items():
        if sub in k:
            return v
    raise KeyError(sub)


counters = {"note": "rocprofv3 --pmc passes of tools/profile_round.sh, averages per dispatch; SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, "
                    "SQ_WAIT_* are in units of 4 shader cycles; FETCH_SIZE / WRITE_SIZE in KB as the tool reports them "
                    "(FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads on gfx950)"}
for label, sub, sq, fetch, write in (("k_ransac (N=4096, H=2048)", "k_ransac", "pmc_sq_head", "pmc_fetch_head", "pmc_write_head"),
                                     ("k_score_lds (N=4096, H=65536)", "k_score_lds", "pmc_sq_big", "pmc_fetch_big", "pmc_write_big"),
                                     ("k_count_f32 (N=4096, H=65536)", "k_count_f32", "pmc_sq_big32", "pmc_fetch_big32", "pmc_write_big32"),
                                     ("k_solve (H=65536)", "k_solve", "pmc_sq_big", "pmc_fetch_big", "pmc_write_big"),
                                     ("k_triangulate4 (16 M points)", "k_triangulate4", "pmc_sq_tri", "pmc_fetch_tri", "pmc_write_tri")):
    e = dict(pick(load(sq), sub))
    e["FETCH_SIZE_KB"] = pick(load(fetch), sub)["FETCH_SIZE"]
    e["WRITE_SIZE_KB"] = pick(load(write), sub)["WRITE_SIZE"]
    e["memory_side_bytes_per_launch"] = int(2 * 1024 * e["FETCH_SIZE_KB"] + 1024 * e["WRITE_SIZE_KB"])
    counters[label] = e
json.dump(counters, open(os.path.join(dst, f"{tag}_counters.json"), "w"), indent=1)

kr, ks, kt = counters["k_ransac (N=4096, H=2048)"], counters["k_score_lds (N=4096, H=65536)"], counters["k_triangulate4 (16 M points)"]
kc = counters["k_count_f32 (N=4096, H=65536)"]
traffic = {
    "build_hash": open(os.path.join(src, "build_hash.txt")).read().strip(),   # slam_build_hash() of the library the passes ran
    "source": f"profiles/{tag}_counters.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), "
              "mean per dispatch; read side doubled (gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane stream)",
    "k_ransac_bytes_per_launch": kr["memory_side_bytes_per_launch"],
    "k_ransac_algorithmic_bytes": 4096 * 32 + 2048 * (256 + 72 + 12),
    "k_score_lds_bytes_per_launch": ks["memory_side_bytes_per_launch"],
    "k_score_lds_algorithmic_bytes": 4096 * 32 + 65536 * (72 + 12),
    "k_count_f32_bytes_per_launch": kc["memory_side_bytes_per_launch"],
    "k_count_f32_algorithmic_bytes": 4096 * 32 + 65536 * (36 + 4),
    "k_triangulate4_bytes_per_launch": kt["memory_side_bytes_per_launch"],
    "k_triangulate4_algorithmic_bytes": (1 << 24) * (32 + 48),
    "note": "k_ransac: the 131 KB of matches are fetched once by each of the 8 compute dies' private L2 (1.0 MB, served by the "
            "Infinity Cache) and every hypothesis reads 256 B of its row of sampler words (0.5 MB); k_score_lds and "
            "k_triangulate4 move what the algorithm needs (matches once per workgroup, F once, results once; two candidate planes)",
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
