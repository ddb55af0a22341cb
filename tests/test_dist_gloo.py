"""Hypothesis-range sharding across ranks, rehearsed with gloo on the CPU (world size 2).

Each rank scores its own iteration range (here with the CPU restatement standing in for the GPU kernels, which is
what a test may do), builds its 96-byte winner record, and the product's exchange code
(slam_amd.pipeline.allgather_winner / merge_records / shard_range) must make every rank agree on the winner the
unsharded loop picks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle as co
    from oracle import slam_oracle as so
    from slam_amd import _lib, pipeline, synth
    d = synth.make_pair(512, 0.4, seed=77)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    lo, hi = pipeline.shard_range(H, rank, world)
    idx = co.sample(512, hi - lo, it0=lo)
    r = co.ransac(n0, n1, idx, want_mask=False)
    rec = np.zeros(_lib.REC_LEN)
    rec[_lib.REC_COUNT] = r["best_count"]
    rec[_lib.REC_ERRSUM] = r["best_errsum"]
    rec[_lib.REC_INDEX] = lo + r["best_index"]
    rec[_lib.REC_F:] = r["best_F"].ravel()
    who, win = pipeline.allgather_winner(rec)
    np.save(os.path.join(out_dir, f"win_{rank}.npy"), np.concatenate([[who], win]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", (64, 301))
def test_hypothesis_sharding_world2(tmp_path, H):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, str(tmp_path)), nprocs=world, join=True)
    wins = [np.load(tmp_path / f"win_{r}.npy") for r in range(world)]
    assert np.array_equal(wins[0], wins[1])                    # every rank holds the same winner
    # and it is the winner of the unsharded loop
    from oracle import c_oracle as co
    from oracle import slam_oracle as so
    from slam_amd import _lib, synth
    d = synth.make_pair(512, 0.4, seed=77)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    full = co.ransac(n0, n1, co.sample(512, H), want_mask=False)
    assert int(wins[0][1 + _lib.REC_INDEX]) == full["best_index"]
    assert int(wins[0][1 + _lib.REC_COUNT]) == full["best_count"]
    assert np.array_equal(wins[0][1 + _lib.REC_F:].reshape(3, 3), full["best_F"])
