"""The numpy restatement in oracle/ must reproduce what the reference itself produced
(golden vectors captured by tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import fix_sign_scale, load_golden, rel_err
from oracle import slam_oracle as so

SAMPLER_NS = (8, 9, 10, 12, 21, 22, 64, 85, 86, 100, 255, 256, 257, 1000, 2048, 4096, 4097, 65535, 65536, 1 << 20)


@pytest.mark.parametrize("N", SAMPLER_NS)
def test_sampler_matches_cpython(N):
    g = load_golden("sampler")
    its = g[f"its_{N}"]
    got = so.sample_indices(N, its)
    assert np.array_equal(got, g[f"idx_{N}"])


@pytest.mark.parametrize("N", (6, 20, 21, 22, 300))
def test_sampler_k6(N):
    g = load_golden("sampler")
    got = so.sample_indices(N, g[f"k6_its_{N}"], k=6)
    assert np.array_equal(got, g[f"k6_idx_{N}"])


def test_sampler_survey_pins():
    assert so.sample_indices(4096, [0])[0].tolist() == [2464, 1951, 1732, 2252, 382, 2443, 3679, 3301]
    assert so.sample_indices(85, [999])[0].tolist() == [22, 0, 21, 36, 32, 82, 49, 16]
    assert so.sample_indices(86, [999])[0].tolist() == [22, 0, 21, 36, 32, 49, 16, 6]


def test_normalize_points():
    g = load_golden("normalize")
    for i in range(3):
        for j in range(2):
            q, T = so.normalize_points(g[f"in_{i}_{j}"])
            assert rel_err(q, g[f"pts_{i}_{j}"]) < 1e-14
            assert rel_err(T, g[f"T_{i}_{j}"]) < 1e-14


def test_solve_known_answer():
    g = load_golden("solve")
    F = so.eight_point(g["kat_a"], g["kat_b"])
    assert np.allclose(F, g["kat_F"], rtol=0, atol=1e-15)
    assert abs(np.linalg.norm(F) - 0.9945512502256663) < 1e-14


def test_solve_minimal_batch():
    g = load_golden("solve")
    F = so.eight_point_batch(g["min_p0"], g["min_p1"], g["min_idx"])
    for h in range(len(F)):
        assert rel_err(fix_sign_scale(F[h]), fix_sign_scale(g["min_F"][h])) < 1e-9
        one = so.eight_point(g["min_p0"][g["min_idx"][h]], g["min_p1"][g["min_idx"][h]])
        assert np.array_equal(one, g["min_F"][h])


@pytest.mark.parametrize("M", (9, 12, 40, 200))
def test_solve_overdetermined(M):
    g = load_golden("solve")
    F = so.eight_point(g[f"over_p0_{M}"], g[f"over_p1_{M}"])
    assert np.array_equal(F, g[f"over_F_{M}"])


@pytest.mark.parametrize("name", ("n8", "n10", "n12", "n100", "n256", "n4096", "n5000"))
def test_ransac_trace(name):
    g = load_golden("ransac_" + name)
    r = so.ransac(g["p0"], g["p1"], int(g["iterations"]))
    assert np.array_equal(r["idx"], g["idx"])
    assert np.array_equal(r["counts"], g["counts"])
    assert np.allclose(r["errsums"], g["errsums"], rtol=1e-9, atol=1e-15)
    if name != "n10":
        assert r["best_index"] == int(g["best_index"])
        assert np.array_equal(r["best_mask"], g["best_mask"])
        assert rel_err(fix_sign_scale(r["best_F"]), fix_sign_scale(g["F_full"])) < 1e-9
    assert r["best_count"] == int(g["best_count"])


def test_triangulate():
    g = load_golden("triangulate")
    X = so.triangulate(g["kat_P0"], g["kat_P1"], g["kat_a"], g["kat_b"])
    assert rel_err(X, g["kat_X"]) < 1e-12
    X = so.triangulate(g["P0"], g["P1"], g["p0"], g["p1"])
    assert np.max(np.abs(X - g["X"]) / np.maximum(np.abs(g["X"]), 1e-3)) < 1e-7
    X = so.triangulate(g["Q0"], g["Q1"], g["q0"], g["q1"])
    assert np.max(np.abs(X - g["XQ"]) / np.maximum(np.abs(g["XQ"]), 1e-3)) < 1e-7


@pytest.mark.parametrize("name", ("clean", "mixed", "heavy"))
def test_pose_from_essential(name):
    g = load_golden("pose_" + name)
    E = so.fundamental_to_essential(g["F"], g["K"])
    assert rel_err(E, g["E"]) < 1e-13
    cands = so.pose_candidates(g["E"])
    for k, (R, t) in enumerate(cands):
        assert np.allclose(R, g["cand_R"][k], atol=1e-13)
        assert np.allclose(t, g["cand_t"][k], atol=1e-13)
    pose, X, counts = so.pose_from_essential(g["E"], g["pts0"], g["pts1"], g["K"])
    assert np.array_equal(counts, g["cand_counts"])
    assert np.allclose(pose, g["pose"], atol=1e-13)
    assert np.max(np.abs(X - g["X"]) / np.maximum(np.abs(g["X"]), 1e-3)) < 1e-7


@pytest.mark.parametrize("name", ("clean", "mixed", "heavy"))
def test_full_pair(name):
    g = load_golden("pose_" + name)
    r = so.initialise_pair(g["pts0"], g["pts1"], g["K"])
    assert rel_err(fix_sign_scale(r["F"]), fix_sign_scale(g["F"])) < 1e-8
    assert np.array_equal(r["cheirality"], g["cand_counts"])


# ---------------------------------------------------------------------------------------------- tracking step
TRACK_CASES = ("m6", "m7", "m40", "m300")


def test_rq_decomposition_matches_reference():
    g = load_golden("tracking_solve")
    r, q = so.rq_decomposition(g["rq_in"])
    assert np.allclose(r, g["rq_r"], rtol=0, atol=1e-14) and np.allclose(q, g["rq_q"], rtol=0, atol=1e-14)
    # the reference's function does not factor its argument: r @ q is the argument transposed about the anti-diagonal
    J = np.rot90(np.eye(3))
    assert np.allclose(r @ q, J @ g["rq_in"].T @ J)


def test_pose_from_points_self_check_fixture():
    g = load_golden("tracking_solve")
    assert rel_err(so.pose_from_points(g["self_uv"], g["self_X"]), g["self_pose"]) < 1e-9


@pytest.mark.parametrize("M", (6, 7, 20, 300))
def test_pose_from_points_matches_golden(M):
    g = load_golden("tracking_solve")
    assert rel_err(so.pose_from_points(g[f"solve_uv_{M}"], g[f"solve_X_{M}"]), g[f"solve_pose_{M}"]) < 1e-9


@pytest.mark.parametrize("name", TRACK_CASES)
def test_tracking_ransac_matches_golden(name):
    g = load_golden("tracking_" + name)
    r = so.pose_from_points_ransac(g["previous_pose"], g["prev_known"], g["next_known"], g["X_known"], g["prev_all"],
                                   g["next_all"], g["K"], int(g["iterations"]))
    assert np.array_equal(r["idx"], g["idx"])
    assert np.array_equal(r["counts"], g["counts"])
    assert np.max(np.abs(r["errs"] / g["errs"] - 1)) < 1e-9
    assert r["best_index"] == int(g["best_index"]) and r["best_count"] == int(g["best_count"])
    assert rel_err(r["pose"], g["pose"]) < 1e-9
    assert rel_err(r["X"], g["X"]) < 1e-9


def test_tracking_asserts():
    g = load_golden("tracking_m6")
    with pytest.raises(AssertionError):
        so.pose_from_points(g["next_known"][:5], g["X_known"][:5])
    with pytest.raises(AssertionError):
        so.pose_from_points_ransac(g["previous_pose"], g["prev_known"][:5], g["next_known"][:5], g["X_known"][:5],
                                   g["prev_all"], g["next_all"], g["K"])


# ---------------------------------------------------------------------------------------------- descriptor matching
@pytest.mark.parametrize("name", ("rand", "ties", "wide", "two"))
def test_matching_restatements_match_golden(name):
    from oracle import c_oracle as co
    g = load_golden("matching")
    for impl in (so.match_knn_brute_force, co.match_knn):
        index, distance = impl(g[name + "_des0"], g[name + "_des1"], 2)
        assert np.array_equal(index, g[name + "_knn_index"]) and np.array_equal(distance, g[name + "_knn_distance"])
    m0, m1, _ = so.select_matches(g[name + "_pts0"], g[name + "_pts1"], g[name + "_knn_index"], g[name + "_knn_distance"])
    assert np.array_equal(m0, g[name + "_matched0"]) and np.array_equal(m1, g[name + "_matched1"])
    c0, c1 = co.select_matches(g[name + "_pts0"], g[name + "_pts1"], g[name + "_knn_index"], g[name + "_knn_distance"])
    assert np.array_equal(c0, m0) and np.array_equal(c1, m1)


def test_matching_three_nearest():
    from oracle import c_oracle as co
    g = load_golden("matching")
    for impl in (so.match_knn_brute_force, co.match_knn):
        index, distance = impl(g["k3_des0"], g["k3_des1"], 3)
        assert np.array_equal(index, g["k3_knn_index"]) and np.array_equal(distance, g["k3_knn_distance"])


# ---------------------------------------------------------------------------------------------- sampler vs the interpreter
def test_samplers_replay_the_interpreters_generator():
    """The restated samplers against CPython's own random module (the generator the reference calls,
    src/epipolar.py:70-71 and src/triangulation.py:154-155), on sizes around every branch of random.sample."""
    import random
    from oracle import c_oracle as co
    rng = np.random.default_rng(17)
    sizes = [8, 9, 16, 20, 21, 22, 63, 64, 65, 84, 85, 86, 87, 127, 128, 129, 1023, 1024, 1025] + \
        [int(x) for x in rng.integers(8, 1 << 21, 40)]
    for N in sizes:
        for k in (8, 6):
            its = [0, 1, 2, 3, 999, int(rng.integers(0, 1 << 20))]
            want = []
            for it in its:
                random.seed(so.SEED_BASE + it)
                want.append(random.sample(range(N), k))
            want = np.array(want, dtype=np.int32)
            got = np.array([so.sample_indices(N, it + 1, k=k)[it] for it in its[:4]], dtype=np.int32)
            assert np.array_equal(got, want[:4]), (N, k)
            for j, it in enumerate(its):
                assert np.array_equal(co.sample(N, 1, k=k, it0=it)[0], want[j]), (N, k, it)


# ---------------------------------------------------------------------------------------------- configuration 1, whole chain
def test_config1_chain_restatement_matches_reference():
    """Keypoints + descriptors -> matches -> F -> E -> pose and points at the reference driver's intrinsics: the numpy
    restatement against what the reference's own functions returned for the same input."""
    g = load_golden("config1")
    index, distance = so.match_knn_brute_force(g["des0"], g["des1"], 2)
    m0, m1, _ = so.select_matches(g["kp0"], g["kp1"], index, distance)
    assert np.array_equal(m0, g["matched0"]) and np.array_equal(m1, g["matched1"])
    r = so.initialise_pair(m0, m1, g["K"], iterations=1000)
    assert rel_err(fix_sign_scale(r["F"]), fix_sign_scale(g["F"])) < 1e-9
    assert rel_err(fix_sign_scale(r["E"]), fix_sign_scale(g["E"])) < 1e-9
    assert np.allclose(r["pose"], g["pose"], atol=1e-9)
    assert rel_err(r["X"], g["X"]) < 1e-7
    assert np.array_equal(r["mask"], g["good"])


def test_config4_pair_restatement_matches_reference():
    """One pair of the batched configuration (N=2048, 1000 iterations) through the reference's caller-side functions."""
    g = load_golden("config4_pair")
    r = so.initialise_pair(g["pts0"], g["pts1"], g["K"])
    assert np.array_equal(r["ransac"]["counts"], g["counts"])
    assert r["ransac"]["best_index"] == int(g["best_index"])
    assert np.array_equal(r["ransac"]["best_mask"], np.unpackbits(g["best_mask"])[:2048].astype(bool))
    assert rel_err(fix_sign_scale(r["F"]), fix_sign_scale(g["F"])) < 1e-9
    assert rel_err(fix_sign_scale(r["E"]), fix_sign_scale(g["E"])) < 1e-9
    assert np.array_equal(r["cheirality"], g["cand_counts"])
    assert np.allclose(r["pose"], g["pose"], atol=1e-10)
    assert np.array_equal(r["mask"], g["good"])


def test_pose_ties_first_maximum_wins():
    """With LAPACK's SVD (the reference's) the candidate order is the reference's, and the strict `>` of
    src/epipolar.py:129 keeps the first of the tied candidates."""
    g = load_golden("pose_tie")
    assert int(g["cases"]) >= 4
    for k in range(int(g["cases"])):
        pose, X, counts = so.pose_from_essential(g[f"E_{k}"], g[f"pts0_{k}"], g[f"pts1_{k}"], g[f"K_{k}"])
        ref = g[f"cand_counts_{k}"]
        assert np.array_equal(counts, ref) and np.count_nonzero(ref == ref.max()) >= 2
        assert np.allclose(pose, g[f"pose_{k}"], atol=1e-12)
        first = int(np.argmax(ref))
        from slam_amd import util
        assert np.allclose(pose, util.rotation_translation_to_pose(g[f"cand_R_{k}"][first], g[f"cand_t_{k}"][first]), atol=1e-12)
