"""Extensions beyond the reference (never part of parity with it): Sampson scoring and adaptive termination.
Checked against the numpy restatements in oracle/ and against the library's own fixed-length runs."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import slam_oracle as so
from slam_amd import _lib, epipolar, logger, pipeline, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    logger.set_log_level("WARN")
    return _lib.default_context()


def normalized(n, frac, seed):
    d = synth.make_pair(n, frac, seed)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    return n0, n1


@pytest.mark.parametrize("n,h", [(1000, 256), (300, 100), (4096, 64), (5000, 24)])
def test_sampson_scoring_matches_restatement(ctx, n, h):
    n0, n1 = normalized(n, 0.3, seed=500 + n)
    thr = 1e-4
    r = epipolar.ransac_detailed(n0, n1, h, per_hypothesis=True, ctx=ctx, scoring=_lib.SCORE_SAMPSON, threshold=thr)
    counts = np.empty(h, dtype=np.int32)
    sums = np.empty(h)
    for i in range(h):
        counts[i], sums[i], _ = so.score_sampson(r["F_all"][i], n0, n1, thr)
    assert np.array_equal(r["counts"], counts)
    assert np.max(np.abs(r["errsums"] - sums) / np.maximum(sums, 1e-300)) < 1e-10
    best = max(range(h), key=lambda i: (counts[i], -sums[i], -i))
    assert r["best_index"] == best and r["best_count"] == counts[best]
    assert np.array_equal(r["best_mask"], so.score_sampson(r["F_all"][best], n0, n1, thr)[2])
    # the solve does not depend on the scoring mode
    a = epipolar.ransac_detailed(n0, n1, h, per_hypothesis=True, ctx=ctx)
    assert np.array_equal(a["F_all"], r["F_all"])


def test_scoring_mode_does_not_leak(ctx):
    n0, n1 = normalized(512, 0.3, seed=7)
    before = epipolar.ransac_detailed(n0, n1, 200, per_hypothesis=True, ctx=ctx)
    epipolar.ransac_detailed(n0, n1, 200, ctx=ctx, scoring=_lib.SCORE_SAMPSON, threshold=1e-4)
    after = epipolar.ransac_detailed(n0, n1, 200, per_hypothesis=True, ctx=ctx)
    assert np.array_equal(before["counts"], after["counts"]) and np.array_equal(before["errsums"], after["errsums"])


def test_sampson_large_h_path(ctx):
    """More than 4096 hypotheses take the split solve/score kernels: same numbers as the fused kernel gives."""
    n0, n1 = normalized(2000, 0.4, seed=8)
    big = epipolar.ransac_detailed(n0, n1, 6000, per_hypothesis=True, ctx=ctx, scoring=_lib.SCORE_SAMPSON, threshold=1e-4,
                                   sampler=_lib.SAMPLER_COUNTER)
    small = epipolar.ransac_detailed(n0, n1, 3000, per_hypothesis=True, ctx=ctx, scoring=_lib.SCORE_SAMPSON, threshold=1e-4,
                                     sampler=_lib.SAMPLER_COUNTER)
    assert np.array_equal(big["counts"][:3000], small["counts"])
    assert np.array_equal(big["errsums"][:3000], small["errsums"])


@pytest.mark.parametrize("n,frac,chunk,conf", [(4096, 0.3, 256, 0.99), (2000, 0.5, 512, 0.999), (1000, 0.7, 1000, 0.99),
                                               (300, 0.1, 64, 0.9)])
def test_adaptive_termination(ctx, n, frac, chunk, conf):
    n0, n1 = normalized(n, frac, seed=600 + n)
    limit = 20000
    r = epipolar.ransac_adaptive(n0, n1, limit, confidence=conf, chunk=chunk, ctx=ctx)
    ran = r["iterations_run"]
    assert 1 <= ran <= limit and (ran % chunk == 0 or ran == limit)
    fixed = epipolar.ransac_detailed(n0, n1, ran, per_hypothesis=True, ctx=ctx)
    assert np.array_equal(r["record"], fixed["record"])
    assert np.array_equal(r["best_mask"], fixed["best_mask"])
    # the stopping point follows the rule, evaluated on the CPU restatement's counts
    o = co.ransac(n0, n1, co.sample(n, ran), threads=8, want_F=False, want_mask=False)
    assert so.adaptive_iterations(o["counts"], n, chunk, conf, limit) == ran
    if frac <= 0.5:
        assert ran < limit            # with at least half of the matches true the rule stops early


def test_adaptive_runs_to_the_limit_when_nothing_fits(ctx):
    rng = np.random.default_rng(5)
    p0, p1 = rng.normal(size=(500, 2)), rng.normal(size=(500, 2))
    r = epipolar.ransac_adaptive(p0, p1, 3000, confidence=0.99, chunk=1024, ctx=ctx)
    assert r["iterations_run"] == 3000
    fixed = epipolar.ransac_detailed(p0, p1, 3000, ctx=ctx)
    assert np.array_equal(r["record"], fixed["record"])


def test_stream_priority_setting_is_checked(monkeypatch):
    """SLAM_AMD_STREAM_PRIORITY selects the priority level of a context's streams: high (default), normal, low; anything
    else makes slam_create fail with a message, and every valid level gives a working context."""
    import ctypes as C
    lib = _lib.load()
    d = synth.make_pair(300, 0.3, seed=5)
    n0, _ = pipeline.normalize_points(d["pts0"])
    n1, _ = pipeline.normalize_points(d["pts1"])
    want = epipolar.ransac_detailed(n0, n1, 50)
    for level in ("normal", "low", "high"):
        monkeypatch.setenv("SLAM_AMD_STREAM_PRIORITY", level)
        ctx = _lib.Context(0)
        try:
            got = epipolar.ransac_detailed(n0, n1, 50, ctx=ctx)
            assert got["best_index"] == want["best_index"] and got["best_count"] == want["best_count"], level
        finally:
            ctx.close()
    monkeypatch.setenv("SLAM_AMD_STREAM_PRIORITY", "urgent")
    handle = C.c_void_p()
    assert lib.slam_create(0, C.byref(handle)) < 0
    assert b"SLAM_AMD_STREAM_PRIORITY" in lib.slam_last_error(None)
