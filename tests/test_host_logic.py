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


def test_hypothesis_counts_follow_the_reference_formulas():
    """min(iterations, C(n, 8)) for the fundamental-matrix loop (src/epipolar.py:69) and min(iterations, C(m, 6)) for
    the tracking loop (src/triangulation.py:145-146); both entry points are host-only."""
    import math
    lib = _lib.load()
    for n in (7, 8, 9, 10, 12, 20, 100, 4096, 1 << 20):
        for it in (1, 44, 45, 1000, 10 ** 9):
            want = 0 if n < 8 else min(it, math.comb(n, 8))
            assert lib.slam_num_hypotheses(n, it) == want, (n, it)
    for m in (5, 6, 7, 8, 9, 12, 300):
        for it in (1, 7, 28, 100, 10 ** 9):
            want = 0 if m < 6 else min(it, math.comb(m, 6))
            assert lib.slam_num_tracking_hypotheses(m, it) == want, (m, it)


def test_synthetic_tracking_and_feature_scenes_are_consistent():
    t = synth.make_track(50, 120, 0.2, seed=3)
    K = t["K"]
    # the known points project onto the next-image points of the true matches with the true next pose
    Xh = np.hstack([t["X_known"], np.ones((50, 1))])
    proj = (K @ t["next_pose"][:3] @ Xh.T).T
    uv = proj[:, :2] / proj[:, 2:3]
    err = np.linalg.norm(uv - t["next_known"], axis=1)
    assert np.median(err[t["inlier"]]) < 5.0 and err[~t["inlier"]].min() > 5.0
    f = synth.make_feature_pair(40, n_clutter=30, seed=4)
    has = f["truth"] >= 0
    assert has.sum() == 40
    d = np.unpackbits(f["des0"][has] ^ f["des1"][f["truth"][has]], axis=1).sum(axis=1)
    assert np.all(d == 20)                        # partners differ in exactly the flipped bits
