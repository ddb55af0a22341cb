"""Parity of the descriptor matcher on the GPU, through the C ABI, with golden vectors captured from the reference's
brute-force matcher (custom_slam.py:14-26) and its ratio-test loop (:62-73), and with the CPU restatements on larger
inputs.  Everything here is integer or gathered data: compared exactly."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle as co
from oracle import slam_oracle as so
from slam_amd import _lib, logger, matching

pytestmark = pytest.mark.gpu
CASES = ("rand", "ties", "wide", "two")


@pytest.fixture(scope="module")
def ctx():
    logger.set_log_level("WARN")
    return _lib.default_context()


def random_descriptors(n0, n1, n_true, flips, seed):
    rng = np.random.default_rng(seed)
    d0 = rng.integers(0, 256, (n0, 32), dtype=np.uint8)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    where = rng.choice(n1, n_true, replace=False)
    bits = np.unpackbits(d0[:n_true], axis=1)
    for q in range(n_true):
        bits[q, rng.choice(256, flips, replace=False)] ^= 1
    d1[where] = np.packbits(bits, axis=1)
    return d0, d1


def test_knn_three_and_more_nearest(ctx):
    """match_knn_brute_force slices its sorted list at any num_matches (custom_slam.py:24): the reference's own output
    for num_matches = 3, then 3..8 nearest on random descriptors with many equal distances against the oracle, and a
    train set smaller than num_matches (the reference's rows are then shorter)."""
    g = load_golden("matching")
    index, distance = matching.knn(g["k3_des0"], g["k3_des1"], 3, ctx)
    assert np.array_equal(index, g["k3_knn_index"]) and np.array_equal(distance, g["k3_knn_distance"])
    rng = np.random.default_rng(8)
    for k in range(3, matching.MAX_K + 1):
        n0, n1 = int(rng.integers(1, 700)), int(rng.integers(k, 900))
        few = rng.integers(0, 256, (12, 32), dtype=np.uint8)
        d0 = few[rng.integers(0, 12, n0)] ^ (rng.integers(0, 256, (n0, 32), dtype=np.uint8) & 1)
        d1 = few[rng.integers(0, 12, n1)] ^ (rng.integers(0, 256, (n1, 32), dtype=np.uint8) & 1)
        wi, wd = so.match_knn_brute_force(d0, d1, k)
        gi, gd = matching.knn(d0, d1, k, ctx)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd), k
    out = matching.match_knn_brute_force(g["k3_des0"], g["k3_des1"][:2], 5, ctx)
    assert all(len(row) == 2 for row in out)
    with pytest.raises(AssertionError):
        matching.knn(g["k3_des0"], g["k3_des1"], 9, ctx)


@pytest.mark.parametrize("name", CASES)
def test_knn_matches_golden(ctx, name):
    g = load_golden("matching")
    index, distance = matching.knn(g[name + "_des0"], g[name + "_des1"], 2, ctx)
    assert np.array_equal(index, g[name + "_knn_index"])
    assert np.array_equal(distance, g[name + "_knn_distance"])
    i1, d1 = matching.knn(g[name + "_des0"], g[name + "_des1"], 1, ctx)
    assert np.array_equal(i1[:, 0], g[name + "_knn_index"][:, 0]) and np.array_equal(d1[:, 0], g[name + "_knn_distance"][:, 0])


@pytest.mark.parametrize("name", CASES)
def test_matched_points_match_golden(ctx, name):
    g = load_golden("matching")
    m0, m1 = matching.matches_between_keypoints(g[name + "_pts0"], g[name + "_des0"], g[name + "_pts1"], g[name + "_des1"], ctx)
    assert np.array_equal(m0, g[name + "_matched0"])
    assert np.array_equal(m1, g[name + "_matched1"])


def test_reference_structure_and_unpacked_input(ctx):
    g = load_golden("matching")
    bits0, bits1 = np.unpackbits(g["rand_des0"], axis=1), np.unpackbits(g["rand_des1"], axis=1)
    out = matching.match_knn_brute_force(bits0, bits1, 2, ctx)
    assert len(out) == len(bits0) and all(len(row) == 2 for row in out)
    assert [row[0]["index"] for row in out] == g["rand_knn_index"][:, 0].tolist()
    assert [row[1]["distance"] for row in out] == g["rand_knn_distance"][:, 1].tolist()


def test_single_train_descriptor(ctx):
    g = load_golden("matching")
    out = matching.match_knn_brute_force(g["rand_des0"], g["rand_des1"][:1], 2, ctx)
    assert all(len(row) == 1 and row[0]["index"] == 0 for row in out)
    m0, m1 = matching.matches_between_keypoints(g["rand_pts0"], g["rand_des0"], g["rand_pts1"][:1], g["rand_des1"][:1], ctx)
    assert len(m0) == 0 and len(m1) == 0


@pytest.mark.parametrize("n0,n1", [(4096, 4096), (1000, 3001), (70, 9000), (5000, 257)])
def test_larger_sets_against_c_restatement(ctx, n0, n1):
    d0, d1 = random_descriptors(n0, n1, min(n0, n1) // 2, 30, seed=n0 + n1)
    rng = np.random.default_rng(1)
    p0, p1 = rng.uniform(0, 1000, (n0, 2)), rng.uniform(0, 1000, (n1, 2))
    m0, m1, index, distance = matching.matches_between_keypoints(p0, d0, p1, d1, ctx, return_knn=True)
    wi, wd = co.match_knn(d0, d1, 2)
    assert np.array_equal(index, wi) and np.array_equal(distance, wd)
    w0, w1 = co.select_matches(p0, p1, wi, wd)
    assert np.array_equal(m0, w0) and np.array_equal(m1, w1)
    assert len(m0) >= min(n0, n1) // 2 - 5            # the planted counterparts pass the ratio test


def test_many_ties_keep_ascending_index(ctx):
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    d0 = base[rng.integers(0, 4, 300)]
    d1 = base[rng.integers(0, 4, 2000)]
    index, distance = matching.knn(d0, d1, 2, ctx)
    wi, wd = so.match_knn_brute_force(d0, d1, 2)
    assert np.array_equal(index, wi) and np.array_equal(distance, wd)


def test_device_entry_point(ctx):
    import torch
    dev = torch.device("cuda", ctx.device)
    d0, d1 = random_descriptors(3000, 2500, 1200, 25, seed=9)
    rng = np.random.default_rng(2)
    p0, p1 = rng.uniform(0, 1000, (3000, 2)), rng.uniform(0, 1000, (2500, 2))
    t = [torch.from_numpy(x).to(dev) for x in (d0, p0, d1, p1)]
    m0 = torch.empty((3000, 2), dtype=torch.float64, device=dev)
    m1 = torch.empty((3000, 2), dtype=torch.float64, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.slam_match_points_dev(ctx.handle, t[0].data_ptr(), t[1].data_ptr(), 3000, t[2].data_ptr(), t[3].data_ptr(),
                                            2500, m0.data_ptr(), m1.data_ptr(), n.data_ptr(), None, None))
    ctx.check(ctx.lib.slam_synchronize(ctx.handle))
    h0, h1 = matching.matches_between_keypoints(p0, d0, p1, d1, ctx)
    k = int(n.item())
    assert k == len(h0)
    assert np.array_equal(m0[:k].cpu().numpy(), h0) and np.array_equal(m1[:k].cpu().numpy(), h1)


def test_frame_pair_equals_its_two_stages(ctx):
    """slam_frame_pair (descriptors in, pose and points out, matches kept on the device) returns bit for bit what
    matching followed by the whole-pair entry point returns, and what the CPU restatements give within tolerance."""
    from conftest import fix_sign_scale, rel_err
    from slam_amd import pipeline, synth
    from test_gpu_parity import point_err
    d = synth.make_feature_pair(1500, n_clutter=700, seed=21)
    r = pipeline.frame_pair(d["kp0"], d["des0"], d["kp1"], d["des1"], d["K"], iterations=500, ctx=ctx)
    m0, m1 = matching.matches_between_keypoints(d["kp0"], d["des0"], d["kp1"], d["des1"], ctx)
    assert np.array_equal(r["matched0"], m0) and np.array_equal(r["matched1"], m1)
    assert 1450 <= len(m0) <= 1500                         # the scene points match, the clutter fails the ratio test
    two = pipeline.initialise_pair(m0, m1, d["K"], iterations=500, ctx=ctx)
    for key in ("F", "E", "pose", "X", "mask", "record", "counts"):
        assert np.array_equal(np.asarray(r[key]), np.asarray(two[key])), key
    # against the restatements
    wi, wd = so.match_knn_brute_force(d["des0"], d["des1"], 2)
    w0, w1, _ = so.select_matches(d["kp0"], d["kp1"], wi, wd)
    assert np.array_equal(w0, m0) and np.array_equal(w1, m1)
    ref = so.initialise_pair(w0, w1, d["K"], iterations=500)
    assert int(r["record"][_lib.REC_INDEX]) == ref["ransac"]["best_index"]
    assert rel_err(fix_sign_scale(r["F"]), fix_sign_scale(ref["F"])) < 1e-6
    assert np.allclose(r["pose"], ref["pose"], atol=1e-7)
    assert point_err(r["X"], ref["X"]) < 1e-5
    # and the motion is roughly the scene's (the 8-point estimate without refinement is only good to a degree or two)
    R = r["pose"][:3, :3].T
    cos = (np.trace(R.T @ d["R"]) - 1) / 2
    assert np.degrees(np.arccos(np.clip(cos, -1, 1))) < 5.0


def test_frame_pair_with_too_few_matches(ctx):
    from slam_amd import pipeline, synth
    d = synth.make_feature_pair(5, n_clutter=200, seed=22)
    with pytest.raises(AssertionError):
        pipeline.frame_pair(d["kp0"], d["des0"], d["kp1"], d["des1"], d["K"], ctx=ctx)
