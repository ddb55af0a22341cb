"""Descriptor matching on the GPU, with the interface of the reference's matching helpers (custom_slam.py:14-82).

Descriptors may be given the way the reference carries them, unpacked to one 0/1 byte per bit, shape (n, 256)
(custom_slam.py:51), or packed as ORB emits them, shape (n, 32) uint8.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import f64, ptr

MAX_K = 8   # SLAM_MATCH_MAX_K


def pack_descriptors(des):
    des = np.asarray(des)
    if des.ndim == 2 and des.shape[1] == 256:
        des = np.packbits(des.astype(np.uint8), axis=1)
    des = np.ascontiguousarray(des, dtype=np.uint8)
    assert des.ndim == 2 and des.shape[1] == 32, "descriptors must be (n,256) bits or (n,32) bytes"
    return des


def knn(img0_descriptors, img1_descriptors, num_matches=2, ctx=None):
    """(index, distance), each (n0, num_matches) int32, nearest first; equal distances in ascending image-1 index."""
    assert 1 <= num_matches <= MAX_K, "the GPU matcher keeps up to eight nearest descriptors"
    d0, d1 = pack_descriptors(img0_descriptors), pack_descriptors(img1_descriptors)
    index = np.empty((len(d0), num_matches), dtype=np.int32)
    distance = np.empty((len(d0), num_matches), dtype=np.int32)
    if len(d0) == 0:
        return index, distance
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.slam_match_knn(ctx.handle, ptr(d0), len(d0), ptr(d1), len(d1), num_matches, ptr(index), ptr(distance)))
    return index, distance


def match_knn_brute_force(img0_descriptors, img1_descriptors, num_matches, ctx=None):
    """Finds the num_matches nearest descriptors in img1 for each descriptor in img0   (reference:
    custom_slam.py:14-26).  Same return structure: a list per img0 descriptor of {"distance", "index"} dicts."""
    index, distance = knn(img0_descriptors, img1_descriptors, num_matches, ctx)
    return [[{"distance": int(d), "index": int(i)} for i, d in zip(ri, rd) if i >= 0] for ri, rd in zip(index, distance)]


def match_knn(img0_descriptors, img1_descriptors, num_matches=2, ctx=None):
    """The reference's default matcher is an approximate LSH index (custom_slam.py:37-43); this one is exact, i.e. it
    returns what the reference's own match_knn_brute_force returns."""
    return match_knn_brute_force(img0_descriptors, img1_descriptors, num_matches, ctx)


def matches_between_keypoints(img0_pts, img0_des, img1_pts, img1_des, ctx=None, return_knn=False):
    """Matched point lists from keypoints and descriptors of two images: 2-NN, ratio test (best < 0.5 * second best),
    kept matches in img0 keypoint order   (reference: custom_slam.py:60-73, everything after the ORB calls)."""
    p0 = f64(np.asarray(img0_pts)).reshape(-1, 2)
    p1 = f64(np.asarray(img1_pts)).reshape(-1, 2)
    d0, d1 = pack_descriptors(img0_des), pack_descriptors(img1_des)
    assert len(p0) == len(d0) and len(p1) == len(d1)
    m0 = np.empty((len(p0), 2))
    m1 = np.empty((len(p0), 2))
    n = np.zeros(1, dtype=np.int32)
    index = np.empty((len(p0), 2), dtype=np.int32) if return_knn else None
    distance = np.empty((len(p0), 2), dtype=np.int32) if return_knn else None
    if len(p0):
        ctx = ctx or _lib.default_context()
        ctx.check(ctx.lib.slam_match_points(ctx.handle, ptr(d0), ptr(p0), len(p0), ptr(d1), ptr(p1), len(p1), ptr(m0), ptr(m1),
                                            ptr(n), ptr(index), ptr(distance)))
    k = int(n[0])
    assert len(m0[:k]) == len(m1[:k])                # reference: custom_slam.py:71
    if return_knn:
        return m0[:k].copy(), m1[:k].copy(), index, distance
    return m0[:k].copy(), m1[:k].copy()
