"""The exchange step of sharded runs inside the library (RCCL collectives on the context's stream + merge kernel).
One GPU is what a test box has, so the communicator has a single rank: the collectives are then copies, and what is
checked is the plumbing (run-time binding of RCCL, communicator life cycle, stream order, record in = record out) and
the merge kernel on several records against the host rule (src/epipolar.py:78-81 applied across ranks)."""
import numpy as np
import pytest

from slam_amd import _lib, pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm_ctx():
    ctx = _lib.Context()
    pipeline.comm_init(ctx, rank=0, world=1)
    yield ctx
    ctx.close()


def _records(rng, n):
    recs = rng.normal(size=(n, _lib.REC_LEN))
    recs[:, _lib.REC_COUNT] = rng.integers(100, 104, n)
    recs[:, _lib.REC_ERRSUM] = rng.choice([0.25, 0.5, 0.75], n)
    recs[:, _lib.REC_INDEX] = rng.permutation(n) * 7
    return recs


def test_merge_kernel_equals_host_rule(comm_ctx):
    """slam_merge_records_dev against slam_merge_records / pipeline.merge_records on records with many ties in the
    inlier count and the error sum (so that all three fields of the key decide somewhere)."""
    import torch
    ctx = comm_ctx
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 8, 64, 257):
        recs = _records(rng, n)
        best, want = pipeline.merge_records(recs)
        out_h = np.empty(_lib.REC_LEN)
        assert ctx.lib.slam_merge_records(_lib.ptr(recs), n, _lib.ptr(out_h)) == best
        d = torch.from_numpy(recs).to(dev)
        o = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.slam_merge_records_dev(ctx.handle, d.data_ptr(), n, o.data_ptr()))
        ctx.check(ctx.lib.slam_synchronize(ctx.handle))
        assert np.array_equal(o.cpu().numpy(), want)
        assert np.array_equal(out_h, want)


def test_merge_in_place_into_a_rank_slot(comm_ctx):
    """The output may be one of the input slots: slot 0 receives the winner when the winner is another slot."""
    import torch
    ctx = comm_ctx
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(6)
    for n in (2, 5, 8):
        recs = _records(rng, n)
        recs[0, _lib.REC_COUNT] = 1                     # slot 0 loses
        best, want = pipeline.merge_records(recs)
        assert best != 0
        d = torch.from_numpy(recs).to(dev)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.slam_merge_records_dev(ctx.handle, d.data_ptr(), n, d.data_ptr()))
        ctx.check(ctx.lib.slam_synchronize(ctx.handle))
        got = d.cpu().numpy()
        assert np.array_equal(got[0], want) and np.array_equal(got[1:], recs[1:])


def test_comm_count_reports_the_communicator_size(comm_ctx):
    import ctypes as C
    n = C.c_int(-1)
    comm_ctx.check(comm_ctx.lib.slam_comm_count(comm_ctx.handle, None, C.byref(n)))
    assert n.value == 1


def test_exchange_with_one_rank_keeps_the_record(comm_ctx):
    """A RANSAC call followed by the exchange on the same stream, no synchronisation in between: with one rank the
    global winner is the local one, bit for bit; the counts all-reduce leaves the counts as they are."""
    import torch
    from slam_amd import synth
    from oracle import slam_oracle as so
    ctx = comm_ctx
    dev = torch.device("cuda", ctx.device)
    d = synth.make_pair(2048, 0.3, seed=21)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    rec = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    ref = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    args = (p0.data_ptr(), p1.data_ptr(), 2048, None, _lib.SAMPLER_COUNTER, 9, 0, 6000, _lib.DEFAULT_THRESHOLD)
    ctx.check(ctx.lib.slam_ransac_f_dev(ctx.handle, *args, ref.data_ptr(), None, None, None, None))
    ctx.check(ctx.lib.slam_synchronize(ctx.handle))
    for _ in range(3):
        ctx.check(ctx.lib.slam_ransac_f_dev(ctx.handle, *args, rec.data_ptr(), None, None, None, None))
        pipeline.exchange_winner_dev(ctx, rec.data_ptr())
    ctx.check(ctx.lib.slam_synchronize(ctx.handle))
    assert rec[_lib.REC_COUNT].item() > 100
    assert torch.equal(rec, ref)
    counts = torch.tensor([5, 1 << 40, 7, 0], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    pipeline.allreduce_counts4_dev(ctx, counts.data_ptr())
    ctx.check(ctx.lib.slam_synchronize(ctx.handle))
    assert counts.cpu().tolist() == [5, 1 << 40, 7, 0]


def test_merge_after_calls_on_the_internal_streams(comm_ctx):
    """With several calls in flight (slam_set_concurrency) the RANSAC calls run on the context's internal streams; a
    merge or an exchange queued right after them, without a synchronisation, still sees their records: records of four
    hypothesis ranges, merged on the device, equal the record of the whole range."""
    import torch
    from slam_amd import synth
    from oracle import slam_oracle as so
    ctx = comm_ctx
    dev = torch.device("cuda", ctx.device)
    d = synth.make_pair(4096, 0.3, seed=22)
    n0, _ = so.normalize_points(d["pts0"])
    n1, _ = so.normalize_points(d["pts1"])
    p0, p1 = torch.from_numpy(n0).to(dev), torch.from_numpy(n1).to(dev)
    H = 4096
    whole = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    parts = torch.zeros((4, _lib.REC_LEN), dtype=torch.float64, device=dev)
    out = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def call(it0, n, rec):
        ctx.check(ctx.lib.slam_ransac_f_dev(ctx.handle, p0.data_ptr(), p1.data_ptr(), 4096, None, _lib.SAMPLER_COUNTER, 11,
                                            it0, n, _lib.DEFAULT_THRESHOLD, rec, None, None, None, None))
    call(0, H, whole.data_ptr())
    ctx.check(ctx.lib.slam_synchronize(ctx.handle))
    ctx.check(ctx.lib.slam_set_concurrency(ctx.handle, 4))
    try:
        for rep in range(5):
            parts.zero_()
            torch.cuda.synchronize()
            for k in range(4):
                call(k * (H // 4), H // 4, parts[k].data_ptr())
            ctx.check(ctx.lib.slam_merge_records_dev(ctx.handle, parts.data_ptr(), 4, out.data_ptr()))
            pipeline.exchange_winner_dev(ctx, out.data_ptr())
            ctx.check(ctx.lib.slam_synchronize(ctx.handle))
            assert torch.equal(out, whole), rep
    finally:
        ctx.check(ctx.lib.slam_set_concurrency(ctx.handle, 1))


def test_exchange_without_communicator_is_an_error():
    import torch
    ctx = _lib.Context()
    try:
        rec = torch.zeros(_lib.REC_LEN, dtype=torch.float64, device=torch.device("cuda", ctx.device))
        rc = ctx.lib.slam_exchange_winner_dev(ctx.handle, None, rec.data_ptr())
        assert rc < 0 and b"communicator" in ctx.lib.slam_last_error(ctx.handle)
    finally:
        ctx.close()
