"""Synthetic two-view correspondences for tests and benchmarks.

Deterministic (numpy PCG64 via ``default_rng(seed)``): a camera with KITTI-like
intrinsics looks at 3D points uniform in its frustum, moves by a small rotation
and a unit, mostly forward translation, and observes them again with Gaussian
pixel noise; a fraction of second-image points is replaced by uniform clutter.

World/pose conventions follow the reference's camera model so that the data is
a faithful input for its pose recovery: the second camera has rotation ``R`` and
centre ``t`` in world (=first camera) coordinates and projects with
``K [R^T | -R^T t]`` (reference: src/epipolar.py:117-118, src/util.py:30-37).
"""
from __future__ import annotations

import numpy as np

KITTI_K = np.array([[718.856, 0.0, 607.19],
                    [0.0, 718.856, 185.22],
                    [0.0, 0.0, 1.0]])
KITTI_SIZE = (1241.0, 376.0)

# intrinsics hard-coded by the reference driver (custom_slam.py:260-262)
REFERENCE_K = np.array([[984.2439, 0.0, 690.0],
                        [0.0, 980.8141, 233.1966],
                        [0.0, 0.0, 1.0]])
REFERENCE_SIZE = (1392.0, 512.0)


def _rotation(axis, angle):
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    C = 1.0 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def make_pair(n, outlier_frac=0.3, seed=0, K=KITTI_K, size=KITTI_SIZE,
              noise_px=0.5, max_rot_deg=3.0):
    """Return a dict with pixel matches ``pts0, pts1`` (n,2), the inlier flags,
    and the true motion (R, t as the reference defines them)."""
    rng = np.random.default_rng(seed)
    w, h = size
    # motion: rotation <= max_rot_deg about a random axis; unit translation mostly +z
    axis = rng.normal(size=3)
    angle = np.deg2rad(max_rot_deg) * rng.uniform(0.2, 1.0)
    R = _rotation(axis, angle)
    t = np.array([rng.normal(0, 0.15), rng.normal(0, 0.05), 1.0])
    t /= np.linalg.norm(t)

    Kinv = np.linalg.inv(K)
    pts0 = np.empty((n, 2))
    pts1 = np.empty((n, 2))
    X = np.empty((n, 3))
    filled = 0
    while filled < n:
        m = (n - filled) * 2 + 16
        uv = np.column_stack([rng.uniform(0, w, m), rng.uniform(0, h, m), np.ones(m)])
        depth = rng.uniform(5.0, 50.0, m)
        P = (uv @ Kinv.T) * depth[:, None]            # points in camera-0 frame
        q = (P - t) @ R                               # R^T (P - t), camera-1 frame
        ok = q[:, 2] > 1.0
        proj = q @ K.T
        uv1 = proj[:, :2] / proj[:, 2:3]
        ok &= (uv1[:, 0] >= 0) & (uv1[:, 0] < w) & (uv1[:, 1] >= 0) & (uv1[:, 1] < h)
        take = min(int(ok.sum()), n - filled)
        sel = np.flatnonzero(ok)[:take]
        pts0[filled:filled + take] = uv[sel, :2]
        pts1[filled:filled + take] = uv1[sel]
        X[filled:filled + take] = P[sel]
        filled += take
    pts0 += rng.normal(0, noise_px, pts0.shape)
    pts1 += rng.normal(0, noise_px, pts1.shape)
    inlier = np.ones(n, dtype=bool)
    n_out = int(round(outlier_frac * n))
    if n_out:
        out_idx = rng.choice(n, n_out, replace=False)
        pts1[out_idx] = np.column_stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)])
        inlier[out_idx] = False
    return {"pts0": pts0, "pts1": pts1, "inlier": inlier, "R": R, "t": t, "X": X, "K": np.array(K)}


def essential_from_motion(R, t):
    """E (Frobenius-normalised) with x1^T E x0 = 0 for the camera model above."""
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = R.T @ tx
    return E / np.linalg.norm(E)


def make_track(n_known, n_matches, outlier_frac=0.2, seed=0, K=KITTI_K, size=KITTI_SIZE,
               noise_px=0.5, noise_3d=0.05):
    """Inputs of one tracking step (reference: custom_slam.py:229-252, src/triangulation.py:135-142).

    The "previous" camera sits one unit step from the world origin and the "next" camera one further step;
    ``n_known`` matches carry a (noisy) known 3D position, ``n_matches`` is the size of the full match list
    (the known ones come first in it).  A fraction of the known matches gets a wrong next-image point.
    Returns previous_pose (4x4), prev_known, next_known (n_known,2), X_known (n_known,3), prev_all, next_all
    (n_matches,2), the true next pose and K.
    """
    assert n_matches >= n_known
    rng = np.random.default_rng(seed)
    w, h = size

    def pose_of(R, t):
        P = np.eye(4)
        P[:3, :3] = R.T
        P[:3, 3] = -R.T @ t
        return P

    R1 = _rotation(rng.normal(size=3), np.deg2rad(2.0) * rng.uniform(0.2, 1.0))
    t1 = np.array([rng.normal(0, 0.1), rng.normal(0, 0.03), 1.0])
    R2 = R1 @ _rotation(rng.normal(size=3), np.deg2rad(2.0) * rng.uniform(0.2, 1.0))
    t2 = t1 + R1 @ np.array([rng.normal(0, 0.1), rng.normal(0, 0.03), 1.0])
    Kinv = np.linalg.inv(K)
    prev = np.empty((n_matches, 2))
    nxt = np.empty((n_matches, 2))
    X = np.empty((n_matches, 3))
    filled = 0
    while filled < n_matches:
        m = (n_matches - filled) * 2 + 16
        uv = np.column_stack([rng.uniform(0, w, m), rng.uniform(0, h, m), np.ones(m)])
        depth = rng.uniform(5.0, 50.0, m)
        q1 = (uv @ Kinv.T) * depth[:, None]           # previous-camera frame
        P = q1 @ R1.T + t1                            # world
        q2 = (P - t2) @ R2                            # next-camera frame
        proj = q2 @ K.T
        uv2 = proj[:, :2] / proj[:, 2:3]
        ok = (q2[:, 2] > 1.0) & (uv2[:, 0] >= 0) & (uv2[:, 0] < w) & (uv2[:, 1] >= 0) & (uv2[:, 1] < h)
        sel = np.flatnonzero(ok)[:n_matches - filled]
        take = len(sel)
        prev[filled:filled + take] = uv[sel, :2]
        nxt[filled:filled + take] = uv2[sel]
        X[filled:filled + take] = P[sel]
        filled += take
    prev += rng.normal(0, noise_px, prev.shape)
    nxt += rng.normal(0, noise_px, nxt.shape)
    X_known = X[:n_known] + rng.normal(0, noise_3d, (n_known, 3))
    prev_known = prev[:n_known].copy()
    next_known = nxt[:n_known].copy()
    n_out = int(round(outlier_frac * n_known))
    inlier = np.ones(n_known, dtype=bool)
    if n_out:
        out_idx = rng.choice(n_known, n_out, replace=False)
        next_known[out_idx] = np.column_stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)])
        nxt[out_idx] = next_known[out_idx]
        inlier[out_idx] = False
    return {"previous_pose": pose_of(R1, t1), "next_pose": pose_of(R2, t2), "prev_known": prev_known,
            "next_known": next_known, "X_known": X_known, "prev_all": prev, "next_all": nxt,
            "inlier": inlier, "K": np.array(K)}


def make_feature_pair(n_true, n_clutter=0, flip_bits=20, seed=0, outlier_frac=0.0, **kw):
    """Keypoints with 256-bit descriptors for two views of one scene: ``n_true`` scene points seen in both images
    (descriptor of the second view = the first with ``flip_bits`` random bits flipped), ``n_clutter`` unrelated
    keypoints per image, image-1 keypoints in shuffled order.  Returns the dict of make_pair plus kp0, des0, kp1, des1
    (descriptors packed, 32 bytes) and ``truth``: for every image-0 keypoint the index of its image-1 partner or -1."""
    d = make_pair(n_true, outlier_frac, seed, **kw)
    rng = np.random.default_rng(seed + 7919)
    w, h = kw.get("size", KITTI_SIZE)
    bits0 = rng.integers(0, 2, (n_true, 256), dtype=np.uint8)
    bits1 = bits0.copy()
    for i in range(n_true):
        bits1[i, rng.choice(256, flip_bits, replace=False)] ^= 1
    c0 = rng.integers(0, 2, (n_clutter, 256), dtype=np.uint8)
    c1 = rng.integers(0, 2, (n_clutter, 256), dtype=np.uint8)
    kp0 = np.vstack([d["pts0"], np.column_stack([rng.uniform(0, w, n_clutter), rng.uniform(0, h, n_clutter)])])
    kp1 = np.vstack([d["pts1"], np.column_stack([rng.uniform(0, w, n_clutter), rng.uniform(0, h, n_clutter)])])
    des0 = np.packbits(np.vstack([bits0, c0]), axis=1)
    des1 = np.packbits(np.vstack([bits1, c1]), axis=1)
    order0 = rng.permutation(n_true + n_clutter)
    order1 = rng.permutation(n_true + n_clutter)
    where1 = np.empty_like(order1)
    where1[order1] = np.arange(len(order1))
    truth = np.where(order0 < n_true, where1[np.minimum(order0, n_true + n_clutter - 1)], -1)
    d.update(kp0=kp0[order0], des0=des0[order0], kp1=kp1[order1], des1=des1[order1], truth=truth)
    return d


def make_sequence(n_frames, n_landmarks=1500, n_clutter=200, flip_bits=16, seed=0, K=KITTI_K, size=KITTI_SIZE,
                  noise_px=0.3, step=1.0):
    """Keypoints and 256-bit descriptors of ``n_frames`` views of one static scene from a camera that drives forward
    (``step`` units per frame, small rotations): the input of the reference's driver loop (custom_slam.py:203-254)
    after its ORB calls.  Every frame lists the landmarks it sees (pixel noise ``noise_px``, descriptor = the
    landmark's with ``flip_bits`` random bits flipped) and ``n_clutter`` unrelated keypoints, in shuffled order.
    Returns frames [(keypoints (n,2), descriptors (n,32) uint8)], landmark [(n,) landmark id or -1 per keypoint],
    poses [(4,4) world -> camera, frame 0 = identity], X (landmarks in world = frame-0 coordinates) and K."""
    rng = np.random.default_rng(seed)
    w, h = size
    Kinv = np.linalg.inv(K)
    uv = np.column_stack([rng.uniform(0, w, n_landmarks), rng.uniform(0, h, n_landmarks), np.ones(n_landmarks)])
    depth = rng.uniform(8.0 + step * n_frames, 60.0, n_landmarks)
    X = (uv @ Kinv.T) * depth[:, None]
    base = rng.integers(0, 2, (n_landmarks, 256), dtype=np.uint8)
    R, t = np.eye(3), np.zeros(3)
    frames, landmark, poses = [], [], []
    for k in range(n_frames):
        if k > 0:
            t = t + R @ (step * np.array([rng.normal(0, 0.08), rng.normal(0, 0.03), 1.0]))
            R = R @ _rotation(rng.normal(size=3), np.deg2rad(1.5) * rng.uniform(0.2, 1.0))
        pose = np.eye(4)
        pose[:3, :3] = R.T
        pose[:3, 3] = -R.T @ t
        q = (X - t) @ R
        proj = q @ K.T
        px = proj[:, :2] / proj[:, 2:3]
        seen = np.flatnonzero((q[:, 2] > 1.0) & (px[:, 0] >= 0) & (px[:, 0] < w) & (px[:, 1] >= 0) & (px[:, 1] < h))
        kp = px[seen] + rng.normal(0, noise_px, (len(seen), 2))
        bits = base[seen].copy()
        for i in range(len(seen)):
            bits[i, rng.choice(256, flip_bits, replace=False)] ^= 1
        kp = np.vstack([kp, np.column_stack([rng.uniform(0, w, n_clutter), rng.uniform(0, h, n_clutter)])])
        bits = np.vstack([bits, rng.integers(0, 2, (n_clutter, 256), dtype=np.uint8)])
        ids = np.concatenate([seen, -np.ones(n_clutter, dtype=np.int64)])
        order = rng.permutation(len(kp))
        frames.append((np.ascontiguousarray(kp[order]), np.packbits(bits[order], axis=1)))
        landmark.append(ids[order])
        poses.append(pose)
    return {"frames": frames, "landmark": landmark, "poses": poses, "X": X, "K": np.array(K)}
