"""Caller-side glue of the reference driver (custom_slam.py:85-115) on the GPU, plus the multi-GPU exchange."""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import REC_LEN, f64, ptr


def normalize_points(points, ctx=None):
    """(points - mean) / s and the matrix T the reference pairs with it, s = mean distance / sqrt(2)
    (reference: custom_slam.py:85-93; T carries s where a similarity would carry 1/s, as there)."""
    p = f64(np.asarray(points))
    assert p.ndim == 2 and p.shape[1] == 2 and len(p) >= 1
    ctx = ctx or _lib.default_context()
    out = np.empty_like(p)
    T = np.empty((3, 3))
    ctx.check(ctx.lib.slam_normalize_points(ctx.handle, ptr(p), len(p), ptr(out), ptr(T)))
    return out, T


def normalize_points_and_find_fundamental_matrix(img0_points, img1_points, iterations=1000):
    """reference: custom_slam.py:96-102"""
    from . import epipolar
    n0, T0 = normalize_points(img0_points)
    n1, T1 = normalize_points(img1_points)
    F = epipolar.calculate_fundamental_matrix_with_ransac(n0, n1, iterations)
    return T1.T @ F @ T0


def initialise_pair(img0_points, img1_points, intrinsic_camera_matrix, iterations=1000, ctx=None):
    """Whole two-view initialisation of one pair with one upload and one download
    (normalise, RANSAC, denormalise, F->E, pose, z>0 flags: custom_slam.py:96-115 after the matching step)."""
    p0 = f64(np.asarray(img0_points))
    p1 = f64(np.asarray(img1_points))
    K = f64(intrinsic_camera_matrix)
    assert p0.shape == p1.shape and p0.ndim == 2 and p0.shape[1] == 2
    assert len(p0) >= 8
    ctx = ctx or _lib.default_context()
    N = len(p0)
    F, E, pose = np.empty((3, 3)), np.empty((3, 3)), np.empty((4, 4))
    X = np.empty((N, 3))
    good = np.empty(N, dtype=np.uint8)
    record = np.empty(REC_LEN)
    counts4 = np.empty(4, dtype=np.int32)
    winner = np.empty(1, dtype=np.int32)
    ctx.check(ctx.lib.slam_init_pair(ctx.handle, ptr(p0), ptr(p1), N, ptr(K), int(iterations), ptr(F), ptr(E), ptr(pose),
                                     ptr(X), ptr(good), ptr(record), ptr(counts4), ptr(winner)))
    return dict(F=F, E=E, pose=pose, X=X, mask=good.astype(bool), record=record, counts=counts4, winner=int(winner[0]))


def initialise_pairs(pairs, intrinsic_camera_matrix, iterations=1000, ctx=None):
    """initialise_pair for a batch of independent pairs in one library call (one upload, one download, one launch per
    pipeline stage for the whole batch; complete on return).  pairs: sequence of (img0_points, img1_points).
    Returns a list of the dicts initialise_pair returns, bit-identical to calling it pair by pair."""
    K = f64(intrinsic_camera_matrix)
    a = [f64(np.asarray(p0)) for p0, _ in pairs]
    b = [f64(np.asarray(p1)) for _, p1 in pairs]
    for x, y in zip(a, b):
        assert x.shape == y.shape and x.ndim == 2 and x.shape[1] == 2
        assert len(x) >= 8
    P = len(a)
    if P == 0:
        return []
    offsets = np.zeros(P + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(x) for x in a])
    total = int(offsets[-1])
    p0, p1 = np.concatenate(a), np.concatenate(b)
    F, E, pose = np.empty((P, 3, 3)), np.empty((P, 3, 3)), np.empty((P, 4, 4))
    X = np.empty((total, 3))
    good = np.empty(total, dtype=np.uint8)
    record = np.empty((P, REC_LEN))
    counts4 = np.empty((P, 4), dtype=np.int32)
    winner = np.empty(P, dtype=np.int32)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_init_pairs(ctx.handle, ptr(p0), ptr(p1), ptr(offsets), P, ptr(K), int(iterations), ptr(F), ptr(E),
                                      ptr(pose), ptr(X), ptr(good), ptr(record), ptr(counts4), ptr(winner)))
    out = []
    for i in range(P):
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        out.append(dict(F=F[i], E=E[i], pose=pose[i], X=X[lo:hi], mask=good[lo:hi].astype(bool), record=record[i],
                        counts=counts4[i], winner=int(winner[i])))
    return out


def initialise_pairs_packed(pts0, pts1, offsets, intrinsic_camera_matrix, iterations=1000, ctx=None, out=None):
    """initialise_pairs for a caller that already holds its batch packed: pts0, pts1 (total, 2) float64 with pair p in
    rows offsets[p]:offsets[p+1].  Returns arrays instead of a list of dicts: F, E (P,3,3), pose (P,4,4), X (total,3),
    mask (total,) bool, record (P,12), counts (P,4), winner (P,).  Same numbers as initialise_pairs.
    out: a dict returned by an earlier call of the same shape (or with X and mask from Context.pinned_empty), whose
    arrays are written again instead of allocating new ones."""
    p0, p1, K = f64(np.asarray(pts0)), f64(np.asarray(pts1)), f64(intrinsic_camera_matrix)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    P = len(offsets) - 1
    assert P >= 1 and p0.shape == p1.shape and p0.ndim == 2 and p0.shape[1] == 2
    assert offsets[0] == 0 and offsets[-1] == len(p0) and np.all(np.diff(offsets) >= 8)
    total = len(p0)
    if out is not None:
        F, E, pose, X, record, counts4, winner = (out[k] for k in ("F", "E", "pose", "X", "record", "counts", "winner"))
        good = out["mask"].view(np.uint8)
        assert X.shape == (total, 3) and good.shape == (total,) and F.shape == (P, 3, 3) and pose.shape == (P, 4, 4)
        assert all(a.flags.c_contiguous for a in (F, E, pose, X, good, record, counts4, winner))
    else:
        F, E, pose = np.empty((P, 3, 3)), np.empty((P, 3, 3)), np.empty((P, 4, 4))
        X = np.empty((total, 3))
        good = np.empty(total, dtype=np.uint8)
        record = np.empty((P, REC_LEN))
        counts4 = np.empty((P, 4), dtype=np.int32)
        winner = np.empty(P, dtype=np.int32)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_init_pairs(ctx.handle, ptr(p0), ptr(p1), ptr(offsets), P, ptr(K), int(iterations), ptr(F), ptr(E),
                                      ptr(pose), ptr(X), ptr(good), ptr(record), ptr(counts4), ptr(winner)))
    return dict(F=F, E=E, pose=pose, X=X, mask=good.view(np.bool_), record=record, counts=counts4, winner=winner)


def frame_pair(img0_pts, img0_des, img1_pts, img1_des, intrinsic_camera_matrix, iterations=1000, ctx=None):
    """Keypoints and descriptors of two images -> matches (2-NN, ratio test) -> whole two-view initialisation, in one
    library call; the matches stay on the device between the two stages   (reference: custom_slam.py:56-82 after the
    ORB calls, then :96-115).  Returns the dict of initialise_pair plus the matched point lists."""
    from .matching import pack_descriptors
    p0 = f64(np.asarray(img0_pts)).reshape(-1, 2)
    p1 = f64(np.asarray(img1_pts)).reshape(-1, 2)
    d0, d1 = pack_descriptors(img0_des), pack_descriptors(img1_des)
    assert len(p0) == len(d0) and len(p1) == len(d1)
    K = f64(intrinsic_camera_matrix)
    n0 = len(p0)
    m0, m1 = np.empty((n0, 2)), np.empty((n0, 2))
    n = np.zeros(1, dtype=np.int32)
    F, E, pose = np.empty((3, 3)), np.empty((3, 3)), np.empty((4, 4))
    X = np.empty((n0, 3))
    good = np.empty(n0, dtype=np.uint8)
    record = np.empty(REC_LEN)
    counts4 = np.empty(4, dtype=np.int32)
    winner = np.empty(1, dtype=np.int32)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_frame_pair(ctx.handle, ptr(d0), ptr(p0), n0, ptr(d1), ptr(p1), len(p1), ptr(K), int(iterations),
                                      ptr(m0), ptr(m1), ptr(n), ptr(F), ptr(E), ptr(pose), ptr(X), ptr(good), ptr(record),
                                      ptr(counts4), ptr(winner)))
    k = int(n[0])
    return dict(matched0=m0[:k].copy(), matched1=m1[:k].copy(), F=F, E=E, pose=pose, X=X[:k].copy(),
                mask=good[:k].astype(bool), record=record, counts=counts4, winner=int(winner[0]))


def find_pose_and_triangulated_points(img0_points, img1_points, intrinsic_camera_matrix, iterations=1000):
    """The reference's return tuple (custom_slam.py:105-115), starting from matches instead of images."""
    r = initialise_pair(img0_points, img1_points, intrinsic_camera_matrix, iterations)
    return r["pose"], [np.asarray(img0_points), np.asarray(img1_points)], r["X"], r["mask"]


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU: hypothesis ranges per rank, one all-gather of the 96-byte winner records, same rule on every rank
# ------------------------------------------------------------------------------------------------------------------
def better_record(x, y):
    """True if record x beats record y under the reference's rule (src/epipolar.py:78-81) extended across ranks:
    more inliers, else smaller inlier-error sum, else earlier iteration."""
    if x[_lib.REC_COUNT] != y[_lib.REC_COUNT]:
        return x[_lib.REC_COUNT] > y[_lib.REC_COUNT]
    if x[_lib.REC_ERRSUM] != y[_lib.REC_ERRSUM]:
        return x[_lib.REC_ERRSUM] < y[_lib.REC_ERRSUM]
    return x[_lib.REC_INDEX] < y[_lib.REC_INDEX]


def merge_records(records):
    """Winner among per-rank records ((R, 12) array-like); pure host logic, usable without a GPU."""
    records = np.asarray(records, dtype=np.float64).reshape(-1, REC_LEN)
    best = 0
    for r in range(1, len(records)):
        if better_record(records[r], records[best]):
            best = r
    return best, records[best].copy()


def shard_range(total, rank, world):
    """Contiguous hypothesis range [lo, hi) of `rank`; ranges differ in size by at most one."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_winner(local_record, group=None):
    """All-gather the local winner record over torch.distributed (RCCL for CUDA tensors, gloo for CPU tensors)
    and apply the selection rule; every rank returns the same (rank_of_winner, record)."""
    import torch
    import torch.distributed as dist
    t = local_record if isinstance(local_record, torch.Tensor) else torch.from_numpy(np.asarray(local_record, dtype=np.float64))
    world = dist.get_world_size(group)
    out = torch.empty((world, REC_LEN), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t.reshape(1, REC_LEN).contiguous(), group=group)
    return merge_records(out.cpu().numpy())


# ------------------------------------------------------------------------------------------------------------------
# the same exchange inside the library: RCCL collectives on the context's stream, merge on the device
# ------------------------------------------------------------------------------------------------------------------
COMM_ID_BYTES = 128


def comm_init(ctx, rank=0, world=1, group=None):
    """Give the context its own RCCL communicator over `world` ranks (collective: every rank calls it).  Rank 0 draws
    the identifier; with more than one rank it travels through a torch.distributed broadcast in `group` (any backend;
    a C host would hand the 128 bytes over by whatever means it has)."""
    import ctypes as C
    buf = (C.c_char * COMM_ID_BYTES)()
    if rank == 0:
        ctx.check(ctx.lib.slam_comm_unique_id(ctx.handle, C.addressof(buf)))
    if world > 1:
        import torch.distributed as dist
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        buf.raw = box[0]
    ctx.check(ctx.lib.slam_comm_init_rank(ctx.handle, C.addressof(buf), int(world), int(rank)))


def exchange_winner_dev(ctx, d_record, comm=None):
    """Hypothesis-sharded run: d_record (device address of this rank's 12-double winner record) holds the global
    winner on every rank afterwards.  All-gather plus the selection rule (src/epipolar.py:78-81) on the device, queued
    on the context's stream: nothing returns to the host.  comm: a caller's RCCL communicator, or None for the
    context's own (comm_init)."""
    ctx.check(ctx.lib.slam_exchange_winner_dev(ctx.handle, comm, d_record))


def allreduce_counts4_dev(ctx, d_counts4, comm=None):
    """Point-sharded pose recovery: sums the four in-front counts (int64[4] on the device) over the ranks in place,
    between slam_pose_candidates_dev and slam_pose_select_dev, on the context's stream."""
    ctx.check(ctx.lib.slam_allreduce_counts4_dev(ctx.handle, comm, d_counts4))
