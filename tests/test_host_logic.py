"""Host-side logic that needs no GPU: record merging, range sharding, the synthetic generator, drop-in modules."""
import importlib

import numpy as np
import pytest

from slam_amd import _lib, pipeline, synth, util
from oracle import slam_oracle as so


def rec(count, errsum, index):
    r = np.zeros(_lib.REC_LEN)
    r[_lib.REC_COUNT], r[_lib.REC_ERRSUM], r[_lib.REC_INDEX] = count, errsum, index
    r[_lib.REC_F:] = index
    return r


def test_merge_rule_matches_sequential_reference_rule():
    rng = np.random.default_rng(0)
    for _ in range(200):
        H = int(rng.integers(1, 40))
        counts = rng.integers(0, 4, H)
        errs = rng.choice([0.1, 0.2, 0.3], H)
        best_c, best_e, best_i = -1, None, -1
        for h in range(H):                      # src/epipolar.py:78-81
            if so.better(counts[h], errs[h], best_c, best_e):
                best_c, best_e, best_i = counts[h], errs[h], h
        world = int(rng.integers(1, 9))
        locals_ = []
        for g in range(world):
            lo, hi = pipeline.shard_range(H, g, world)
            if lo == hi:
                locals_.append(rec(-1, 0.0, 2 ** 31 - 1))
                continue
            _, r = pipeline.merge_records([rec(counts[h], errs[h], h) for h in range(lo, hi)])
            locals_.append(r)
        _, win = pipeline.merge_records(locals_)
        assert int(win[_lib.REC_INDEX]) == best_i
        # the C entry point applies the same rule
        lib = _lib.load()
        arr = np.ascontiguousarray(np.array(locals_))
        out = np.empty(_lib.REC_LEN)
        pos = lib.slam_merge_records(_lib.ptr(arr), len(arr), _lib.ptr(out))
        assert pos >= 0 and int(out[_lib.REC_INDEX]) == best_i


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 1000, 2 ** 20):
        for world in (1, 2, 3, 8):
            spans = [pipeline.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_synth_pair_is_deterministic_and_consistent():
    a = synth.make_pair(500, 0.3, seed=7)
    b = synth.make_pair(500, 0.3, seed=7)
    assert np.array_equal(a["pts0"], b["pts0"]) and np.array_equal(a["pts1"], b["pts1"])
    assert a["inlier"].sum() == 350
    E = synth.essential_from_motion(a["R"], a["t"])
    Kinv = np.linalg.inv(a["K"])
    x0 = np.column_stack([a["pts0"], np.ones(500)]) @ Kinv.T
    x1 = np.column_stack([a["pts1"], np.ones(500)]) @ Kinv.T
    r = np.abs(np.einsum("ni,ij,nj->n", x1, E, x0))
    assert np.median(r[a["inlier"]]) < 2e-3 < np.median(r[~a["inlier"]])


def test_pose_helpers_round_trip():
    d = synth.make_pair(8, 0, seed=1)
    pose = util.rotation_translation_to_pose(d["R"], d["t"])
    assert np.allclose(util.pose_to_rotation(pose), d["R"])
    assert np.allclose(util.pose_to_translation(pose), d["t"])
    assert np.allclose(pose, so.rotation_translation_to_pose(d["R"], d["t"]))


def test_drop_in_package_layout():
    for name in ("epipolar", "triangulation", "util", "logger", "plot", "display"):
        importlib.import_module("src." + name)
    from src import epipolar, triangulation
    for fn in ("calculate_fundamental_matrix", "calculate_fundamental_matrix_with_ransac",
               "fundamental_to_essential_matrix", "calculate_pose_from_essential_matrix", "two_d_to_three_d"):
        assert callable(getattr(epipolar, fn))
    assert callable(triangulation.triangulate_points_from_pose)


def test_shape_preconditions_raise_like_the_reference():
    from slam_amd import epipolar
    with pytest.raises(AssertionError):
        epipolar.calculate_fundamental_matrix(np.zeros((7, 2)), np.zeros((7, 2)))
    with pytest.raises(AssertionError):
        epipolar.calculate_fundamental_matrix_with_ransac(np.zeros((9, 2)), np.zeros((8, 2)))
