"""slam_ransac_f_dev_noalloc under HIP stream capture: what is captured can be replayed after later calls have changed
the context's sampler table; the one form that cannot (CPython-compatible sampler without pre-drawn indices) is refused
while the stream is capturing."""
import numpy as np
import pytest

from oracle import slam_oracle as so
from slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


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
