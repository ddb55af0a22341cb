#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference itself.

Run in the build container only (the reference tree does not exist on the GPU
box):   cd /tmp && MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

The reference modules are imported from /root/reference unchanged.  Its import
chain names three viewer libraries the hot path never calls (cv2, pangolin,
OpenGL); empty placeholder modules satisfy those imports, and the driver
script's request for an interactive matplotlib backend is ignored.  Every array
stored here is either an input we generated or an output the reference's own
functions returned for it; no reference source text is stored.
"""
import contextlib
import io
import math
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from slam_amd import synth  # noqa: E402  (build-owned input generator)

for _m in ("cv2", "pangolin", "OpenGL", "OpenGL.GL"):
    sys.modules[_m] = types.ModuleType(_m)
import matplotlib  # noqa: E402

matplotlib.use = lambda *a, **k: None
sys.path.insert(0, "/root/reference")
# the repo ships its own drop-in package called `src`; make sure the reference's wins here
for _k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
    del sys.modules[_k]
from src import epipolar, logger, triangulation, util  # noqa: E402
import custom_slam  # noqa: E402

assert epipolar.__file__.startswith("/root/reference"), epipolar.__file__
logger.set_log_level("WARN")

SEED_BASE = 0x1337BEEF
THRESH = 0.005


def draw(N, it, k=8):
    random.seed(SEED_BASE + it)
    return random.sample(range(N), k)


def gen_sampler():
    out = {}
    for N in (8, 9, 10, 12, 21, 22, 64, 85, 86, 100, 255, 256, 257, 1000, 2048, 4096, 4097, 65535, 65536, 1 << 20):
        its = list(range(64)) + [999, 1000, 2047, 65535, 1 << 20, (1 << 32) - SEED_BASE - 1, (1 << 32) - SEED_BASE, (1 << 32) - SEED_BASE + 5]
        out[f"its_{N}"] = np.array(its, dtype=np.int64)
        out[f"idx_{N}"] = np.array([draw(N, it) for it in its], dtype=np.int32)
    # k = 6 is what the tracking RANSAC draws (src/triangulation.py:155)
    for N in (6, 20, 21, 22, 300):
        its = list(range(32))
        out[f"k6_its_{N}"] = np.array(its, dtype=np.int64)
        out[f"k6_idx_{N}"] = np.array([draw(N, it, 6) for it in its], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)


def normalized_pair(n, outlier_frac, seed, K=synth.KITTI_K, size=synth.KITTI_SIZE):
    d = synth.make_pair(n, outlier_frac, seed, K=K, size=size)
    n0, T0 = custom_slam.normalize_points(d["pts0"])
    n1, T1 = custom_slam.normalize_points(d["pts1"])
    return d, n0, T0, n1, T1


def gen_normalize():
    out = {}
    for i, (n, seed) in enumerate([(8, 11), (100, 12), (4096, 2)]):
        d = synth.make_pair(n, 0.3, seed)
        for j, p in enumerate((d["pts0"], d["pts1"])):
            q, T = custom_slam.normalize_points(p)
            out[f"in_{i}_{j}"] = p
            out[f"pts_{i}_{j}"] = q
            out[f"T_{i}_{j}"] = T
    np.savez_compressed(os.path.join(HERE, "normalize.npz"), **out)


def gen_solve():
    out = {}
    a = np.array([[-1, -.5], [.3, .9], [1.2, -.7], [-.4, .2], [.8, .6], [-1.1, 1], [.1, -1.3], [.6, .05]])
    b = np.array([[-.9, -.45], [.35, .8], [1.1, -.75], [-.5, .25], [.85, .5], [-1, 1.1], [.15, -1.2], [.5, .1]])
    out["kat_a"], out["kat_b"] = a, b
    out["kat_F"] = epipolar.calculate_fundamental_matrix(a, b)
    # minimal (8-point) solves on index sets drawn the way the RANSAC loop draws them
    d, n0, T0, n1, T1 = normalized_pair(256, 0.3, 21)
    H = 96
    idx = np.array([draw(256, it) for it in range(H)], dtype=np.int32)
    F = np.array([epipolar.calculate_fundamental_matrix(n0[list(i)], n1[list(i)]) for i in idx])
    out["min_p0"], out["min_p1"], out["min_idx"], out["min_F"] = n0, n1, idx, F
    # over-determined solves (the function accepts M >= 8: src/epipolar.py:23)
    for M in (9, 12, 40, 200):
        dd, m0, _, m1, _ = normalized_pair(M, 0.0, 30 + M)
        out[f"over_p0_{M}"], out[f"over_p1_{M}"] = m0, m1
        out[f"over_F_{M}"] = epipolar.calculate_fundamental_matrix(m0, m1)
    np.savez_compressed(os.path.join(HERE, "solve.npz"), **out)


def ransac_trace(n0, n1, iterations):
    """Per-hypothesis quantities evaluated with the reference's own expressions
    (src/epipolar.py:64-77), plus the winner derived by its rule (:78-81)."""
    N = len(n0)
    H = min(iterations, math.comb(N, 8))
    x0 = np.hstack((n0, np.ones((N, 1))))
    x1 = np.hstack((n1, np.ones((N, 1))))
    idx = np.empty((H, 8), dtype=np.int32)
    Fs = np.empty((H, 3, 3))
    counts = np.empty(H, dtype=np.int32)
    errsums = np.empty(H)
    best, best_err, best_i = -1, None, -1
    masks = None
    for it in range(H):
        ch = draw(N, it)
        idx[it] = ch
        F = epipolar.calculate_fundamental_matrix(n0[ch], n1[ch])
        errors = np.square([(p1.T @ F) @ p0 for p0, p1 in zip(x0, x1)])
        mask = errors < THRESH
        e = np.sum(errors[mask])
        c = np.count_nonzero(mask)
        Fs[it], counts[it], errsums[it] = F, c, e
        if c > best or (c == best and e < best_err):
            best, best_err, best_i, masks = c, e, it, mask.copy()
    return dict(idx=idx, F=Fs, counts=counts, errsums=errsums, best_index=np.int32(best_i),
                best_count=np.int32(best), best_errsum=np.float64(best_err), best_mask=masks)


def gen_ransac():
    cases = [  # name, N, outlier fraction, seed, iterations
        ("n8", 8, 0.0, 41, 1000),
        ("n10", 10, 0.2, 42, 1000),
        ("n12", 12, 0.25, 43, 1000),
        ("n100", 100, 0.3, 44, 1000),
        ("n256", 256, 0.3, 45, 300),
        ("n4096", 4096, 0.3, 2, 48),
        ("n5000", 5000, 0.7, 46, 24),
    ]
    for name, N, frac, seed, iters in cases:
        d, n0, T0, n1, T1 = normalized_pair(N, frac, seed)
        tr = ransac_trace(n0, n1, iters)
        buf = io.StringIO()
        logger.set_log_level("INFO")
        with contextlib.redirect_stdout(buf):
            F_full = epipolar.calculate_fundamental_matrix_with_ransac(n0, n1, iterations=iters)
        logger.set_log_level("WARN")
        # the full call must return exactly the per-hypothesis F at the derived winner
        assert np.array_equal(F_full, tr["F"][tr["best_index"]]), name
        np.savez_compressed(os.path.join(HERE, f"ransac_{name}.npz"), p0=n0, p1=n1,
                            iterations=np.int32(iters), F_full=F_full, log_line=np.array(buf.getvalue()), **tr)
        print(name, "H", len(tr["counts"]), "winner", int(tr["best_index"]), "count", int(tr["best_count"]),
              buf.getvalue().strip())


def gen_config2():
    """The headline configuration run by the reference itself, every one of its 2048 iterations: N=4096 matches, 30 %
    wrong, seed 2 (the pair of ransac_n4096.npz, which holds the points; they are not stored twice).  About a minute
    of reference time.  Stored: every hypothesis' inlier count and error sum, the winner, its mask, the F the full call
    returns, its log line, and every 64th hypothesis' F."""
    d, n0, T0, n1, T1 = normalized_pair(4096, 0.3, 2)
    held = np.load(os.path.join(HERE, "ransac_n4096.npz"))
    assert np.array_equal(held["p0"], n0) and np.array_equal(held["p1"], n1)
    tr = ransac_trace(n0, n1, 2048)
    buf = io.StringIO()
    logger.set_log_level("INFO")
    with contextlib.redirect_stdout(buf):
        F_full = epipolar.calculate_fundamental_matrix_with_ransac(n0, n1, iterations=2048)
    logger.set_log_level("WARN")
    assert np.array_equal(F_full, tr["F"][tr["best_index"]])
    np.savez_compressed(os.path.join(HERE, "ransac_config2.npz"), iterations=np.int32(2048), F_full=F_full,
                        log_line=np.array(buf.getvalue()), counts=tr["counts"], errsums=tr["errsums"],
                        best_index=tr["best_index"], best_count=tr["best_count"], best_errsum=tr["best_errsum"],
                        best_mask=np.packbits(tr["best_mask"]), idx=tr["idx"].astype(np.int16), F_every_64th=tr["F"][::64])
    print("config2: winner", int(tr["best_index"]), "count", int(tr["best_count"]), buf.getvalue().strip())


def gen_config4():
    """One pair of the batched-sequence configuration through the reference's own caller-side functions: seed 100,
    N=2048 matches, 30 % wrong, KITTI-like intrinsics, 1000 iterations (custom_slam.py:96-102, then
    src/epipolar.py:87-136)."""
    d = synth.make_pair(2048, 0.3, 100)
    K = d["K"]
    buf = io.StringIO()
    logger.set_log_level("DEBUG")
    with contextlib.redirect_stdout(buf):
        F = custom_slam.normalize_points_and_find_fundamental_matrix(d["pts0"], d["pts1"])
        E = epipolar.fundamental_to_essential_matrix(F, K)
        pose, X = epipolar.calculate_pose_from_essential_matrix(E, d["pts0"], d["pts1"], K)
    logger.set_log_level("WARN")
    Rs, ts, counts = pose_trace(E, d["pts0"], d["pts1"], K)
    n0, T0 = custom_slam.normalize_points(d["pts0"])
    n1, T1 = custom_slam.normalize_points(d["pts1"])
    tr = ransac_trace(n0, n1, 1000)
    assert np.array_equal(T1.T @ tr["F"][tr["best_index"]] @ T0, F)
    np.savez_compressed(os.path.join(HERE, "config4_pair.npz"), pts0=d["pts0"], pts1=d["pts1"], K=K, F=F, E=E, pose=pose, X=X,
                        good=X[:, 2] > 0, cand_counts=counts, log=np.array(buf.getvalue()), counts=tr["counts"],
                        errsums=tr["errsums"], best_index=tr["best_index"], best_count=tr["best_count"],
                        best_mask=np.packbits(tr["best_mask"]))
    print("config4 pair: winner", int(tr["best_index"]), "count", int(tr["best_count"]), "cheirality", counts,
          buf.getvalue().strip().replace("\n", " | "))


def gen_pose_tie():
    """Pose recovery where two of the four candidates tie on the in-front count, so that the strict `>` of
    src/epipolar.py:129 (the FIRST maximum in the order (R0,t0),(R0,t1),(R1,t0),(R1,t1) wins) decides.  Found by
    searching small scenes of mostly wrong matches; the candidate order is that of the reference's SVD."""
    found = 0
    out = {}
    for seed in range(2000, 4000):
        N = 8 + seed % 9
        d = synth.make_pair(N, 0.8, seed)
        K = d["K"]
        try:
            F = custom_slam.normalize_points_and_find_fundamental_matrix(d["pts0"], d["pts1"])
        except Exception:
            continue
        E = epipolar.fundamental_to_essential_matrix(F, K)
        Rs, ts, counts = pose_trace(E, d["pts0"], d["pts1"], K)
        top = counts.max()
        if np.count_nonzero(counts == top) < 2 or top < 2:
            continue
        buf = io.StringIO()
        logger.set_log_level("DEBUG")
        with contextlib.redirect_stdout(buf):
            pose, X = epipolar.calculate_pose_from_essential_matrix(E, d["pts0"], d["pts1"], K)
        logger.set_log_level("WARN")
        k = found
        out[f"pts0_{k}"], out[f"pts1_{k}"], out[f"K_{k}"], out[f"E_{k}"] = d["pts0"], d["pts1"], K, E
        out[f"cand_R_{k}"], out[f"cand_t_{k}"], out[f"cand_counts_{k}"] = Rs, ts, counts
        out[f"pose_{k}"], out[f"X_{k}"] = pose, X
        print("pose tie", k, "seed", seed, "N", N, "counts", counts, buf.getvalue().strip())
        found += 1
        if found == 6:
            break
    out["cases"] = np.int32(found)
    np.savez_compressed(os.path.join(HERE, "pose_tie.npz"), **out)


def gen_triangulate():
    out = {}
    a = np.array([[-1, -.5], [.3, .9]])
    b = np.array([[-.9, -.45], [.35, .8]])
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = np.hstack([np.eye(3), np.array([[-1.0], [0.1], [0.05]])])
    out["kat_P0"], out["kat_P1"], out["kat_a"], out["kat_b"] = P0, P1, a, b
    out["kat_X"] = triangulation.triangulate_points_from_pose(P0, P1, a, b)
    d = synth.make_pair(512, 0.3, 51)
    K = d["K"]
    P0 = K @ util.rotation_translation_to_pose(np.identity(3), np.zeros(3))[:3]
    P1 = K @ util.rotation_translation_to_pose(d["R"], d["t"])[:3]
    out["P0"], out["P1"], out["p0"], out["p1"] = P0, P1, d["pts0"], d["pts1"]
    out["X"] = triangulation.triangulate_points_from_pose(P0, P1, d["pts0"], d["pts1"])
    # 4x4 "camera matrices" (only rows 0..2 are read): src/triangulation.py:179-180
    Q0 = util.rotation_translation_to_pose(np.identity(3), np.zeros(3))
    Q1 = util.rotation_translation_to_pose(d["R"], d["t"])
    n0 = (np.hstack([d["pts0"][:64], np.ones((64, 1))]) @ np.linalg.inv(K).T)[:, :2]
    n1 = (np.hstack([d["pts1"][:64], np.ones((64, 1))]) @ np.linalg.inv(K).T)[:, :2]
    out["Q0"], out["Q1"], out["q0"], out["q1"] = Q0, Q1, n0, n1
    out["XQ"] = triangulation.triangulate_points_from_pose(Q0, Q1, n0, n1)
    np.savez_compressed(os.path.join(HERE, "triangulate.npz"), **out)


def pose_trace(E, p0, p1, K):
    """Candidate enumeration exactly as src/epipolar.py:97-133, recording each candidate."""
    u, s, v = np.linalg.svd(E)
    w = np.asarray([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    rots = [u @ w @ v, u @ w.T @ v]
    mult = 1 if np.linalg.det(rots[0]) > 0 else -1
    rots = [mult * r for r in rots]
    trans = [u[:, -1], -1 * u[:, -1]]
    Rs, ts, counts = [], [], []
    for R in rots:
        for t in trans:
            P0 = K @ util.rotation_translation_to_pose(np.identity(3), np.zeros(3))[:3]
            P1 = K @ util.rotation_translation_to_pose(R, t)[:3]
            X = triangulation.triangulate_points_from_pose(P0, P1, p0, p1)
            c2 = R @ np.asarray([0, 0, 1])
            f1 = X[:, 2] > 0
            f2 = np.dot((X - t), c2) > 0
            Rs.append(R); ts.append(t); counts.append(np.count_nonzero(np.multiply(f1, f2)))
    return np.array(Rs), np.array(ts), np.array(counts, dtype=np.int32)


def gen_pose():
    cases = [("clean", 300, 0.0, 61), ("mixed", 400, 0.3, 62), ("heavy", 300, 0.7, 63)]
    for name, N, frac, seed in cases:
        d = synth.make_pair(N, frac, seed)
        K = d["K"]
        # F from the full front half of the path (normalise -> RANSAC -> denormalise), as custom_slam.py:96-102
        F = custom_slam.normalize_points_and_find_fundamental_matrix(d["pts0"], d["pts1"])
        E = epipolar.fundamental_to_essential_matrix(F, K)
        Rs, ts, counts = pose_trace(E, d["pts0"], d["pts1"], K)
        buf = io.StringIO()
        logger.set_log_level("DEBUG")
        with contextlib.redirect_stdout(buf):
            pose, X = epipolar.calculate_pose_from_essential_matrix(E, d["pts0"], d["pts1"], K)
        logger.set_log_level("WARN")
        np.savez_compressed(os.path.join(HERE, f"pose_{name}.npz"), pts0=d["pts0"], pts1=d["pts1"], K=K,
                            F=F, E=E, cand_R=Rs, cand_t=ts, cand_counts=counts, pose=pose, X=X,
                            true_R=d["R"], true_t=d["t"], log_line=np.array(buf.getvalue()))
        print(name, "cheirality counts", counts, buf.getvalue().strip())


def quiet(fn, *a, **k):
    """Call a reference function with its debugging prints swallowed."""
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def tracking_trace(d, iterations):
    """Per-iteration quantities of the tracking RANSAC evaluated with the reference's own functions and
    expressions (src/triangulation.py:153-176), plus the winner by its rule."""
    a, b, X3, K, prev_pose = d["prev_known"], d["next_known"], d["X_known"], d["K"], d["previous_pose"]
    M = len(X3)
    H = min(iterations, math.comb(M, 6))
    idx = np.empty((H, 6), dtype=np.int32)
    poses = np.empty((H, 4, 4))
    counts = np.empty(H, dtype=np.int32)
    errs = np.empty(H)
    best, best_err, best_i = -1, None, -1
    for it in range(H):
        ch = draw(M, it, 6)
        idx[it] = ch
        pose = quiet(triangulation.triangulate_pose_from_points, b[ch], X3[ch], K)
        X = triangulation.triangulate_points_from_pose(K @ prev_pose[:3], K @ pose[:3], a, b)
        R = util.pose_to_rotation(pose)
        t = util.pose_to_translation(pose)
        c2 = R @ np.asarray([0, 0, 1])
        f1 = X[:, 2] > 0
        f2 = np.dot((X - t), c2) > 0
        c = np.count_nonzero(np.multiply(f1, f2))
        e = np.sum(np.linalg.norm(X3 - X, axis=1))
        poses[it], counts[it], errs[it] = pose, c, e
        if c > best or (c == best and best_err > e):
            best, best_err, best_i = c, e, it
    return dict(idx=idx, poses=poses, counts=counts, errs=errs, best_index=np.int32(best_i),
                best_count=np.int32(best), best_err=np.float64(best_err))


def gen_tracking():
    out = {}
    # the 8 correspondences and intrinsics of the reference's own self-check (src/triangulation.py:188-206)
    X8 = np.asarray([[6.03787985, -4.89579876, 62.66137844], [-8.39317134, 3.03552779, 18.92342537],
                     [-15.53855632, -2.46611132, 92.2252502], [-2.58636809, 0.3066047, 20.0799362],
                     [-16.17911693, -0.85989838, 44.88294597], [-18.83353488, -10.9979917, 71.92970211],
                     [2.28097053, -0.6059205, 58.0935008], [3.27491554, -1.18588676, 85.08488668]])
    uv8 = np.asarray([[783.60003662, 152.40000916], [142.56001282, 427.68002319], [513.60003662, 204.00001526],
                      [504.5760498, 250.56002808], [299.52001953, 213.12001038], [413.2800293, 73.44000244],
                      [726.58959961, 221.46052551], [727.38592529, 218.57409668]])
    out["self_X"], out["self_uv"], out["self_K"] = X8, uv8, synth.REFERENCE_K
    out["self_pose"] = quiet(triangulation.triangulate_pose_from_points, uv8, X8, synth.REFERENCE_K)
    # single solves on 6, 7, 20 and 300 correspondences of a synthetic step
    d = synth.make_track(300, 300, 0.0, 71)
    for M in (6, 7, 20, 300):
        out[f"solve_uv_{M}"], out[f"solve_X_{M}"] = d["next_known"][:M], d["X_known"][:M]
        out[f"solve_pose_{M}"] = quiet(triangulation.triangulate_pose_from_points, d["next_known"][:M], d["X_known"][:M], d["K"])
    r = np.array([[2.0, -1.0, 0.5], [0.3, 1.5, -0.7], [-0.2, 0.4, 1.1]])
    out["rq_in"] = r
    out["rq_r"], out["rq_q"] = util.rq_decomposition(r)
    np.savez_compressed(os.path.join(HERE, "tracking_solve.npz"), **out)

    cases = [("m6", 6, 40, 0.0, 72, 100), ("m7", 7, 40, 0.0, 73, 100), ("m40", 40, 200, 0.2, 74, 100),
             ("m300", 300, 1000, 0.3, 75, 100)]
    for name, M, N, frac, seed, iters in cases:
        d = synth.make_track(M, N, frac, seed)
        tr = tracking_trace(d, iters)
        buf = io.StringIO()
        logger.set_log_level("INFO")
        with contextlib.redirect_stdout(buf):
            pose, X = triangulation.triangulate_pose_from_points_with_ransac(
                d["previous_pose"], d["prev_known"], d["next_known"], d["X_known"], d["prev_all"], d["next_all"],
                d["K"], iterations=iters)
        logger.set_log_level("WARN")
        assert np.array_equal(pose, tr["poses"][tr["best_index"]]), name
        log_line = [ln for ln in buf.getvalue().splitlines() if "Pose from points" in ln][-1]
        np.savez_compressed(os.path.join(HERE, f"tracking_{name}.npz"), previous_pose=d["previous_pose"],
                            prev_known=d["prev_known"], next_known=d["next_known"], X_known=d["X_known"],
                            prev_all=d["prev_all"], next_all=d["next_all"], K=d["K"], true_next_pose=d["next_pose"],
                            iterations=np.int32(iters), pose=pose, X=X, log_line=np.array(log_line), **tr)
        print(name, "H", len(tr["counts"]), "winner", int(tr["best_index"]), "count", int(tr["best_count"]),
              "err", float(tr["best_err"]), log_line)


def synth_descriptors(n0, n1, n_true, flip_bits, seed, distinct=None):
    """Unpacked 256-bit descriptors (0/1 uint8, as custom_slam.py:51 produces them): n_true of the n0 queries have a
    counterpart in the train set that differs in about flip_bits bits; `distinct` limits the number of different
    descriptors to force ties."""
    rng = np.random.default_rng(seed)
    if distinct:
        base = rng.integers(0, 2, (distinct, 256), dtype=np.uint8)
        d0 = base[rng.integers(0, distinct, n0)]
        d1 = base[rng.integers(0, distinct, n1)]
        return d0, d1
    d0 = rng.integers(0, 2, (n0, 256), dtype=np.uint8)
    d1 = rng.integers(0, 2, (n1, 256), dtype=np.uint8)
    where = rng.choice(n1, n_true, replace=False)
    for q, j in enumerate(where):
        d = d0[q].copy()
        d[rng.choice(256, flip_bits, replace=False)] ^= 1
        d1[j] = d
    return d0, d1


def gen_matching():
    out = {}
    cases = [("rand", 40, 50, 25, 20, 91, None), ("ties", 30, 45, 0, 0, 92, 6), ("wide", 70, 33, 20, 40, 93, None),
             ("two", 5, 2, 2, 10, 94, None)]
    for name, n0, n1, n_true, flips, seed, distinct in cases:
        d0, d1 = synth_descriptors(n0, n1, n_true, flips, seed, distinct)
        rng = np.random.default_rng(seed + 1000)
        pts0 = rng.uniform(0, 1000, (n0, 2))
        pts1 = rng.uniform(0, 1000, (n1, 2))
        knn = custom_slam.match_knn_brute_force(d0, d1, 2)
        # the selection loop of custom_slam.py:62-73 applied to the reference's own k-NN result
        m0, m1 = [], []
        for idx, (m, n) in enumerate(knn):
            if m["distance"] < 0.5 * n["distance"]:
                m0.append(pts0[idx])
                m1.append(pts1[m["index"]])
        out[f"{name}_des0"], out[f"{name}_des1"] = np.packbits(d0, axis=1), np.packbits(d1, axis=1)
        out[f"{name}_pts0"], out[f"{name}_pts1"] = pts0, pts1
        out[f"{name}_knn_index"] = np.array([[e["index"] for e in row] for row in knn], dtype=np.int32)
        out[f"{name}_knn_distance"] = np.array([[e["distance"] for e in row] for row in knn], dtype=np.int32)
        out[f"{name}_matched0"] = np.asarray(m0).reshape(-1, 2)
        out[f"{name}_matched1"] = np.asarray(m1).reshape(-1, 2)
        print("matching", name, "queries", n0, "train", n1, "kept", len(m0))
    # k = 3 on one case: the function takes any k
    d0, d1 = synth_descriptors(12, 20, 6, 25, 95)
    knn = custom_slam.match_knn_brute_force(d0, d1, 3)
    out["k3_des0"], out["k3_des1"] = np.packbits(d0, axis=1), np.packbits(d1, axis=1)
    out["k3_knn_index"] = np.array([[e["index"] for e in row] for row in knn], dtype=np.int32)
    out["k3_knn_distance"] = np.array([[e["distance"] for e in row] for row in knn], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "matching.npz"), **out)


def gen_config1():
    """The reference's own driver sequence (custom_slam.py:56-82 after the ORB calls, then :96-115) on a synthetic
    stand-in for its bundled image pair: 500 keypoints per image (ORB's default budget) with 256-bit descriptors at
    the intrinsics the driver hard-codes (custom_slam.py:260-262), 1000 RANSAC iterations (the default)."""
    d = synth.make_feature_pair(150, n_clutter=350, seed=111, K=synth.REFERENCE_K, size=synth.REFERENCE_SIZE)
    K = d["K"]
    bits0, bits1 = np.unpackbits(d["des0"], axis=1), np.unpackbits(d["des1"], axis=1)    # as custom_slam.py:51
    knn = custom_slam.match_knn_brute_force(bits0, bits1, 2)
    m0, m1 = [], []
    for idx, (m, n) in enumerate(knn):                                                  # custom_slam.py:65-70
        if m["distance"] < 0.5 * n["distance"]:
            m0.append(d["kp0"][idx])
            m1.append(d["kp1"][m["index"]])
    m0, m1 = np.asarray(m0), np.asarray(m1)
    buf = io.StringIO()
    logger.set_log_level("DEBUG")
    with contextlib.redirect_stdout(buf):
        F = custom_slam.normalize_points_and_find_fundamental_matrix(m0, m1)            # :107
        E = epipolar.fundamental_to_essential_matrix(F, K)                              # :108
        pose, X = epipolar.calculate_pose_from_essential_matrix(E, m0, m1, K)           # :109-112
    logger.set_log_level("WARN")
    np.savez_compressed(os.path.join(HERE, "config1.npz"), kp0=d["kp0"], des0=d["des0"], kp1=d["kp1"], des1=d["des1"], K=K,
                        matched0=m0, matched1=m1, F=F, E=E, pose=pose, X=X, good=X[:, 2] > 0, true_R=d["R"], true_t=d["t"],
                        log=np.array(buf.getvalue()))
    print("config1: matches", len(m0), buf.getvalue().strip().replace("\n", " | "))


def gen_driver():
    """The reference's own driver object (custom_slam.py:118-254: Slam.find_initial_pose, then Slam.find_next_pose, the
    two calls its main() makes) run on a synthetic three-frame sequence.  Only the two things OpenCV does for it are
    replaced, by assignment at run time: compute_orb returns a frame's synthetic keypoints and unpacked descriptor
    bits, and match_knn (an approximate OpenCV index) calls the reference's own exact match_knn_brute_force.  An
    "image" is a frame number.  Everything else - matching loop, ratio test, normalisation, both RANSACs, the map
    lookups by 2D point - is the reference's code, unchanged."""
    seq = synth.make_sequence(3, n_landmarks=260, n_clutter=60, seed=131, K=synth.REFERENCE_K, size=synth.REFERENCE_SIZE)
    frames = seq["frames"]
    sys.modules["cv2"].ORB_create = lambda *a, **k: None
    custom_slam.compute_orb = lambda img, orb: (np.asarray(frames[img][0]), np.unpackbits(frames[img][1], axis=1))
    custom_slam.match_knn = lambda d0, d1, num_matches=2: custom_slam.match_knn_brute_force(d0, d1, num_matches)
    slam = custom_slam.Slam(seq["K"], show_image_keypoints=False, show_3d_visualization=False)
    buf = io.StringIO()
    logger.set_log_level("INFO")
    with contextlib.redirect_stdout(buf):
        slam.find_initial_pose(0, 1)
        n_map = len(slam.triangulated_point_lookup[2])
        latest = np.array(slam.latest_image_2d_points)
        all_pts = np.array(slam.all_triangulated_points)
    with contextlib.redirect_stdout(io.StringIO()):       # the tracking loop prints every intermediate matrix
        logger.set_log_level("WARN")
        slam.find_next_pose(1, 2)
    out = {"K": seq["K"], "pose1": slam.camera_poses[1], "pose2": slam.camera_poses[2], "n_poses": len(slam.camera_poses),
           "n_map_frame2": n_map, "latest_image_2d_points": latest, "all_triangulated_points": all_pts,
           "true_pose1": seq["poses"][1], "true_pose2": seq["poses"][2], "log": np.array(buf.getvalue())}
    for k, (kp, des) in enumerate(frames):
        out[f"kp{k}"], out[f"des{k}"] = kp, des
    np.savez_compressed(os.path.join(HERE, "driver.npz"), **out)
    print("driver: map points for frame 2:", n_map, "| poses", len(slam.camera_poses), "|",
          buf.getvalue().strip().splitlines()[-1])


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_sampler()
    gen_normalize()
    gen_solve()
    gen_ransac()
    gen_triangulate()
    gen_pose()
    gen_tracking()
    gen_matching()
    gen_config1()
    gen_driver()
    gen_config2()
    gen_config4()
    gen_pose_tie()
    print("python", sys.version.split()[0], "numpy", np.__version__)
