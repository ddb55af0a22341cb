"""slam_ransac_f_dev_noalloc (the entry point for callers that cannot allocate) against slam_ransac_f_dev, through the
C ABI: bit-identical record and mask, two calls in a row, the second on a smaller input; the size limit is enforced;
what is captured into a HIP graph replays unchanged after later calls have rewritten the context's sampler table.

Collected last on purpose (the file name sorts after every other test file)."""
import numpy as np
import pytest

from oracle import slam_oracle as so
from slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _pair_on_device(torch, dev, N, seed):
    d = synth.make_pair(N, 0.3, seed=seed)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    return torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)


def _both(ctx, torch, dev, p0, p1, N, H, sampler, seed, it0=0, d_idx=None):
    lib = ctx.lib
    out = []
    for fn, extra in ((lib.slam_ransac_f_dev, (None, None, None)), (lib.slam_ransac_f_dev_noalloc, ())):
        rec = torch.full((_lib.REC_LEN,), -7.0, dtype=torch.float64, device=dev)
        mask = torch.full((N,), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.check(fn(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None if d_idx is None else d_idx.data_ptr(), sampler, seed,
                     it0, H, 0.005, rec.data_ptr(), mask.data_ptr(), *extra))
        ctx.check(lib.slam_synchronize(ctx.handle))
        out.append((rec.cpu().numpy(), mask.cpu().numpy()))
    return out


def test_noalloc_entry_point_equals_the_allocating_one():
    import torch
    ctx = _lib.default_context()
    dev = torch.device("cuda", ctx.device)
    # first call: the benchmark shape (fused kernel), CPython-compatible sampler; the allocating call comes first and
    # leaves the sampler's word table behind, which the second form then finds
    p0, p1 = _pair_on_device(torch, dev, 4096, 41)
    (r_a, m_a), (r_b, m_b) = _both(ctx, torch, dev, p0, p1, 4096, 2048, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE)
    assert np.array_equal(r_a, r_b) and np.array_equal(m_a, m_b)
    assert r_b[_lib.REC_COUNT] == m_b.sum() and set(np.unique(m_b)) <= {0, 1}
    # second call in a row, smaller input (another scoring shape), counter-based sampler: stale scratch must not leak
    q0, q1 = _pair_on_device(torch, dev, 700, 42)
    (r_a, m_a), (r_b, m_b) = _both(ctx, torch, dev, q0, q1, 700, 300, _lib.SAMPLER_COUNTER, 99, it0=1000)
    assert np.array_equal(r_a, r_b) and np.array_equal(m_a, m_b)
    assert r_b[_lib.REC_COUNT] == m_b.sum() and 1000 <= r_b[_lib.REC_INDEX] < 1300
    # the largest call it takes (split solve / score / select path), pre-drawn indices
    H = _lib.NOALLOC_MAX_H
    idx = np.empty((H, 8), dtype=np.int32)
    ctx.check(ctx.lib.slam_sample_indices(ctx.handle, _lib.SAMPLER_COUNTER, 5, 0, H, 4096, 8, _lib.ptr(idx)))
    d_idx = torch.from_numpy(idx).to(dev)
    (r_a, m_a), (r_b, m_b) = _both(ctx, torch, dev, p0, p1, 4096, H, _lib.SAMPLER_COUNTER, 5, d_idx=d_idx)
    assert np.array_equal(r_a, r_b) and np.array_equal(m_a, m_b)
    # and a small one again after the large one
    (r_a, m_a), (r_b, m_b) = _both(ctx, torch, dev, q0, q1, 700, 8, _lib.SAMPLER_COUNTER, 99)
    assert np.array_equal(r_a, r_b) and np.array_equal(m_a, m_b)


def test_noalloc_entry_point_with_several_chunks_of_matches_and_many_hypotheses():
    """More than 4096 matches AND more than 4096 hypotheses: the allocating call scores through per-chunk partial
    tables in a context buffer; the other form must get by without them (it runs the fused kernel over ranges of 4096
    hypotheses) and still return the same bits.  A fresh context, the non-allocating call first: the device's free
    memory does not change across it."""
    import torch
    ctx = _lib.Context(0)
    try:
        _several_chunks_and_many_hypotheses(torch, ctx)
    finally:
        ctx.close()


def _several_chunks_and_many_hypotheses(torch, ctx):
    dev = torch.device("cuda", ctx.device)
    lib = ctx.lib
    N, H = 300000, 4200   # 74 chunks: the partial tables of the allocating call take 3.7 MB
    p0, p1 = _pair_on_device(torch, dev, N, 44)
    rec = torch.full((_lib.REC_LEN,), -7.0, dtype=torch.float64, device=dev)
    mask = torch.full((N,), 9, dtype=torch.uint8, device=dev)
    want_rec, want_mask = torch.empty_like(rec), torch.empty_like(mask)
    # (a first, small call: the runtime loads the library's code object at the first launch of a process, which takes
    # device memory of its own)
    ctx.check(lib.slam_ransac_f_dev_noalloc(ctx.handle, p0.data_ptr(), p1.data_ptr(), 64, None, _lib.SAMPLER_COUNTER, 7, 0, 8,
                                            0.005, rec.data_ptr(), None))
    ctx.check(lib.slam_synchronize(ctx.handle))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    ctx.check(lib.slam_ransac_f_dev_noalloc(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None, _lib.SAMPLER_COUNTER, 7, 0, H,
                                            0.005, rec.data_ptr(), mask.data_ptr()))
    ctx.check(lib.slam_synchronize(ctx.handle))
    assert torch.cuda.mem_get_info(dev)[0] == free_before
    ctx.check(lib.slam_ransac_f_dev(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None, _lib.SAMPLER_COUNTER, 7, 0, H, 0.005,
                                    want_rec.data_ptr(), want_mask.data_ptr(), None, None, None))
    ctx.check(lib.slam_synchronize(ctx.handle))
    assert torch.equal(rec, want_rec) and torch.equal(mask, want_mask)
    assert rec[_lib.REC_COUNT].item() == mask.sum().item() and rec[_lib.REC_COUNT].item() >= 8
    # pre-drawn indices take the same route (the ranges read their own part of the index table)
    idx = np.empty((H, 8), dtype=np.int32)
    ctx.check(lib.slam_sample_indices(ctx.handle, _lib.SAMPLER_COUNTER, 7, 0, H, N, 8, _lib.ptr(idx)))
    d_idx = torch.from_numpy(idx).to(dev)
    rec.fill_(-7.0)
    ctx.check(lib.slam_ransac_f_dev_noalloc(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, d_idx.data_ptr(), _lib.SAMPLER_COUNTER, 7,
                                            0, H, 0.005, rec.data_ptr(), mask.data_ptr()))
    ctx.check(lib.slam_synchronize(ctx.handle))
    assert torch.equal(rec, want_rec) and torch.equal(mask, want_mask)


def test_noalloc_entry_point_rejects_what_it_cannot_do():
    import torch
    ctx = _lib.default_context()
    dev = torch.device("cuda", ctx.device)
    p0, p1 = _pair_on_device(torch, dev, 512, 43)
    rec = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    lib = ctx.lib
    args = (ctx.handle, p0.data_ptr(), p1.data_ptr(), 512, None)
    rc = lib.slam_ransac_f_dev_noalloc(*args, _lib.SAMPLER_COUNTER, 1, 0, _lib.NOALLOC_MAX_H + 1, 0.005, rec.data_ptr(), None)
    assert rc == -1 and b"at most" in lib.slam_last_error(ctx.handle)
    # CPython-compatible sampler with a seed whose word table nobody has made: refused, not generated
    rc = lib.slam_ransac_f_dev_noalloc(*args, _lib.SAMPLER_CPYTHON, 0xDEADBEEF, 0, 64, 0.005, rec.data_ptr(), None)
    assert rc == -1 and b"word table" in lib.slam_last_error(ctx.handle)
    # the context is still fine afterwards
    rc = lib.slam_ransac_f_dev_noalloc(*args, _lib.SAMPLER_COUNTER, 1, 0, 64, 0.005, rec.data_ptr(), None)
    assert rc == 0 and lib.slam_synchronize(ctx.handle) == 0
    assert rec.cpu().numpy()[_lib.REC_COUNT] >= 8


def test_captured_call_replays_after_other_calls_changed_the_sampler_table():
    import torch
    ctx = _lib.Context(0)
    try:
        lib = ctx.lib
        dev = torch.device("cuda", ctx.device)
        N, H = 4096, 2048
        d = synth.make_pair(N, 0.3, seed=51)
        n0, _ = so.normalize_points(d["pts0"])
        n1, _ = so.normalize_points(d["pts1"])
        p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
        want = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
        want_mask = torch.zeros(N, dtype=torch.uint8, device=dev)
        idx = np.empty((H, 8), dtype=np.int32)      # the reference's samples, drawn once, kept in the caller's buffer
        ctx.check(lib.slam_sample_indices(ctx.handle, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE, 0, H, N, 8, _lib.ptr(idx)))
        d_idx = torch.from_numpy(idx).to(dev)
        torch.cuda.synchronize()
        ctx.check(lib.slam_ransac_f_dev(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON, _lib.SEED_BASE,
                                        0, H, 0.005, want.data_ptr(), want_mask.data_ptr(), None, None, None))
        ctx.check(lib.slam_synchronize(ctx.handle))

        rec = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
        mask = torch.zeros(N, dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream(dev)
        torch.cuda.synchronize()
        ctx.check(lib.slam_set_stream(ctx.handle, s.cuda_stream))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            # refused while capturing, and nothing is queued by the refused call
            rc_table = lib.slam_ransac_f_dev_noalloc(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON,
                                                     _lib.SEED_BASE, 0, H, 0.005, rec.data_ptr(), mask.data_ptr())
            msg = lib.slam_last_error(ctx.handle)
            rc = lib.slam_ransac_f_dev_noalloc(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, d_idx.data_ptr(), _lib.SAMPLER_CPYTHON,
                                               _lib.SEED_BASE, 0, H, 0.005, rec.data_ptr(), mask.data_ptr())
        assert rc_table == -1 and b"capture" in msg
        assert rc == 0
        # a later call with another seed and a longer range regenerates (and moves) the context's word table
        other = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
        ctx.check(lib.slam_ransac_f_dev(ctx.handle, p0.data_ptr(), p1.data_ptr(), N, None, _lib.SAMPLER_CPYTHON, 12345, 0,
                                        3 * H, 0.005, other.data_ptr(), None, None, None, None))
        ctx.check(lib.slam_synchronize(ctx.handle))
        for _ in range(2):
            rec.fill_(-1.0)
            mask.fill_(7)
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                g.replay()
            s.synchronize()
            assert torch.equal(rec, want) and torch.equal(mask, want_mask)
        ctx.check(lib.slam_set_stream(ctx.handle, None))
    finally:
        ctx.close()
