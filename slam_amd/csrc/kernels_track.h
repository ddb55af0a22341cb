// Tracking step: camera pose from 2D-3D matches (DLT resection) and its RANSAC loop, gfx950.
//   reference: src/triangulation.py:78-182, src/util.py:18-27,60-66
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_tri.h"
#include "linalg.h"

namespace slamk {

// Work block of one tracking call (float64 words)
enum {
    TW_P0 = 0,     // 12: K prev_pose[:3]
    TW_PREV = 12,  // 12: prev_pose[:3]
    TW_K = 24,     //  9: K
    TW_WIN = 36,   // 24: prev_pose[:3] | winning pose[:3]   (camera pair of the final triangulation)
    TW_BEST = 60,  //  3: winner position, its count, its error
    TW_LEN = 64
};
// Per-hypothesis block (float64 words)
enum {
    HP_POSE = 0,    // 16: [[R^T, -R^T t],[0,0,0,1]]
    HP_P1 = 16,     // 12: K pose[:3]
    HP_T = 28,      //  3: camera position recovered from the pose (src/util.py:24-27)
    HP_CZ = 31,     //  3: viewing direction, third column of pose[:3,:3]^T (src/util.py:18-21)
    HP_EXTRA = 34,  // 24: calculated K (9), rotation (9), translation (3), camera centre (3): what the reference prints
    HP_LEN = 58
};

#define SLAM_RESECT_LANES 32

__device__ __forceinline__ double det3x3(const double (&m)[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// From the least right singular vector of the design matrix to the hypothesis block   (reference:
// src/triangulation.py:113-132, src/util.py:18-37,60-66)
__device__ inline void resect_finish(double (&cam)[12], const double* __restrict__ tw, double* __restrict__ hp) {
    // :115  sign so that det of the left 3x3 block is positive
    const double m3[9] = {cam[0], cam[1], cam[2], cam[4], cam[5], cam[6], cam[8], cam[9], cam[10]};
    const double d = det3x3(m3);
    const double sg = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 12; ++i) cam[i] *= sg;
    // :117-118  the null vector of the 3x4 matrix by cofactors (only its ratios are used)
    double c4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double s[9];
        int cc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == k) continue;
            s[cc] = cam[j];
            s[3 + cc] = cam[4 + j];
            s[6 + cc] = cam[8 + j];
            ++cc;
        }
        c4[k] = ((k & 1) ? -1.0 : 1.0) * det3x3(s);
    }
    const double centre[3] = {c4[0] / c4[3], c4[1] / c4[3], c4[2] / c4[3]};
    // :120  rq_decomposition (src/util.py:60-66): J m = q' r' ; r = J r'^T J ; q = J q'^T.  Gram-Schmidt with one
    // re-orthogonalisation; whatever signs a QR routine picks for r' cancel in :122-124.
    double Mp[9], Q[9], Rp[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { Mp[3 * i + j] = sg * m3[3 * (2 - i) + j]; Rp[3 * i + j] = 0.0; Q[3 * i + j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double v[3] = {Mp[j], Mp[3 + j], Mp[6 + j]};
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const double dot = Q[k] * v[0] + Q[3 + k] * v[1] + Q[6 + k] * v[2];
                Rp[3 * k + j] += dot;
                v[0] -= dot * Q[k];
                v[1] -= dot * Q[3 + k];
                v[2] -= dot * Q[6 + k];
            }
        const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        Rp[3 * j + j] = nrm;
        Q[j] = v[0] / nrm;
        Q[3 + j] = v[1] / nrm;
        Q[6 + j] = v[2] / nrm;
    }
    double r[9], q[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[3 * i + j] = Rp[3 * (2 - j) + (2 - i)];
            q[3 * i + j] = Q[3 * j + (2 - i)];
        }
    double dg[3], Kc[9], Rm[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dg[i] = r[4 * i] > 0.0 ? 1.0 : (r[4 * i] < 0.0 ? -1.0 : 0.0);  // :122
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Kc[3 * i + j] = r[3 * i + j] * dg[j];  // :123
            Rm[3 * i + j] = dg[i] * q[3 * i + j];  // :124
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = -1.0 * (Rm[3 * i] * centre[0] + Rm[3 * i + 1] * centre[1] + Rm[3 * i + 2] * centre[2]);  // :125

    double pose[12];  // first three rows of [[R^T, -R^T t]]   (src/util.py:30-37)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pose[4 * i + j] = Rm[3 * j + i];
            s += Rm[3 * j + i] * t[j];
        }
        pose[4 * i + 3] = -s;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) hp[HP_POSE + i] = pose[i];
    hp[HP_POSE + 12] = 0.0; hp[HP_POSE + 13] = 0.0; hp[HP_POSE + 14] = 0.0; hp[HP_POSE + 15] = 1.0;
    const double* __restrict__ K = tw + TW_K;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            hp[HP_P1 + 4 * i + j] = K[3 * i] * pose[j] + K[3 * i + 1] * pose[4 + j] + K[3 * i + 2] * pose[8 + j];
    // what the scoring loop reads back from the pose: src/util.py:18-27
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        hp[HP_T + i] = -1.0 * (pose[i] * pose[3] + pose[4 + i] * pose[7] + pose[8 + i] * pose[11]);
        hp[HP_CZ + i] = pose[8 + i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { hp[HP_EXTRA + i] = Kc[i]; hp[HP_EXTRA + 9 + i] = Rm[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { hp[HP_EXTRA + 18 + i] = t[i]; hp[HP_EXTRA + 21 + i] = centre[i]; }
}

// ------------------------------------------------------------------------------------------------
// K5: resection, one lane per hypothesis   (reference: src/triangulation.py:98-132)
//   Rows [0,0,0,0,-X,-Y,-Z,-1, vX,vY,vZ,v] and [X,Y,Z,1, 0,0,0,0, -uX,-uY,-uZ,-u] of every chosen match are folded
//   with Givens rotations into a 12x12 triangular factor, whose least right singular vector (one-sided Jacobi with
//   accumulated V; the Gram matrix would square a condition number of ~1e7) is the camera matrix.  The matrices
//   live in LDS, element-major, so that the 32 lanes of a block touch consecutive words.
//   idx: H x 6 sample indices, or NULL: one solve (H = 1) on all M matches.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SLAM_RESECT_LANES) void k_resect(const double2* __restrict__ uv, const double* __restrict__ X3,
                                                             int M, const int32_t* __restrict__ idx, int H,
                                                             const double* __restrict__ tw, double* __restrict__ hyp) {
    __shared__ double sR[144][SLAM_RESECT_LANES];
    __shared__ double sV[144][SLAM_RESECT_LANES];
    const int lane = threadIdx.x;
    const int h = blockIdx.x * SLAM_RESECT_LANES + lane;
    if (h >= H) return;
#define R_(i, j) sR[(i) * 12 + (j)][lane]
#define V_(i, j) sV[(i) * 12 + (j)][lane]
    for (int e = 0; e < 144; ++e) { sR[e][lane] = 0.0; sV[e][lane] = 0.0; }
    for (int i = 0; i < 12; ++i) V_(i, i) = 1.0;

    const int rows = idx ? 6 : M;
    for (int p = 0; p < rows; ++p) {
        const int i = idx ? idx[(size_t)h * 6 + p] : p;
        const double2 q = uv[i];
        const double x = X3[3 * (size_t)i], y = X3[3 * (size_t)i + 1], z = X3[3 * (size_t)i + 2];
        const double xh[4] = {x, y, z, 1.0};
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            double row[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                row[j] = half ? xh[j] : 0.0;
                row[4 + j] = half ? 0.0 : -xh[j];
                row[8 + j] = half ? -q.x * xh[j] : q.y * xh[j];
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const double yk = row[k];
                if (yk != 0.0) {
                    const double xk = R_(k, k);
                    const double r = sqrt(xk * xk + yk * yk);
                    const double c = xk / r, s = yk / r;
                    R_(k, k) = r;
#pragma unroll
                    for (int j = k + 1; j < 12; ++j) {
                        const double t1 = R_(k, j), t2 = row[j];
                        R_(k, j) = c * t1 + s * t2;
                        row[j] = c * t2 - s * t1;
                    }
                }
            }
        }
    }

    double fro2 = 0.0;
    for (int e = 0; e < 144; ++e) fro2 += sR[e][lane] * sR[e][lane];
    const double floor2 = fro2 * 1e-32;  // columns at rounding level are left alone
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 11; ++p)
            for (int q = p + 1; q < 12; ++q) {
                double cp[12], cq[12];
                double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    cp[i] = R_(i, p);
                    cq[i] = R_(i, q);
                    a = fma(cp[i], cp[i], a);
                    b = fma(cq[i], cq[i], b);
                    c = fma(cp[i], cq[i], c);
                }
                if (!(a > floor2 && b > floor2 && c * c > 1e-30 * (a * b))) continue;
                rotated = true;
                double cs, sn;
                jacobi_angle(a, b, c, cs, sn);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    R_(i, p) = fma(cs, cp[i], -sn * cq[i]);
                    R_(i, q) = fma(sn, cp[i], cs * cq[i]);
                    const double vx = V_(i, p), vy = V_(i, q);
                    V_(i, p) = fma(cs, vx, -sn * vy);
                    V_(i, q) = fma(sn, vx, cs * vy);
                }
            }
        if (!rotated) break;
    }
    int least = 0;
    double best = 1e300;
    for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int i = 0; i < 12; ++i) s += R_(i, j) * R_(i, j);
        if (s < best) { best = s; least = j; }
    }
    double cam[12];
    for (int i = 0; i < 12; ++i) cam[i] = V_(i, least);  // vh[-1].reshape(3,4)   :113-114
#undef R_
#undef V_

    resect_finish(cam, tw, hyp + (size_t)h * HP_LEN);
}

// ------------------------------------------------------------------------------------------------
// K5b: resection of six-point samples, sixteen lanes per hypothesis (four hypotheses per wave).
//   The 12x12 design matrix needs no triangular factor here: lane c of a group holds column c of the matrix and
//   column c of V in registers, and the one-sided Jacobi runs in tournament order - in each of the 11 rounds of a
//   sweep the 12 columns form 6 disjoint pairs that rotate at the same time.  A lane fetches its partner's two
//   columns with 48 ds_bpermute, both lanes of a pair derive the same angle from the same three sums, and each
//   keeps its own side of the rotation.  Same thresholds as the one-lane kernel; sweeps end when no pair of the
//   wave rotates.  Lanes 12..15 of a group idle.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_resect6(const double2* __restrict__ uv, const double* __restrict__ X3,
                                                const int32_t* __restrict__ idx, int H, const double* __restrict__ tw,
                                                double* __restrict__ hyp) {
    const int lane = threadIdx.x;
    const int c = lane & 15;                  // column owned by this lane
    const int base = lane & 48;               // first lane of the group
    const int h = blockIdx.x * 4 + (lane >> 4);
    const bool owner = c < 12 && h < H;
    const int hh = h < H ? h : H - 1;
    double a[12], v[12];
    // column c of the design matrix: rows 2p = [0,0,0,0, -Xh, v Xh], rows 2p+1 = [Xh, 0,0,0,0, -u Xh]
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int i = idx[(size_t)hh * 6 + p];
        const double2 q = uv[i];
        const int k = c & 3;
        const double xk = k == 3 ? 1.0 : X3[3 * (size_t)i + k];
        const int blk = c >> 2;               // 0: first four columns, 1: middle, 2: last (3 for idle lanes)
        a[2 * p] = blk == 1 ? -xk : (blk == 2 ? q.y * xk : 0.0);
        a[2 * p + 1] = blk == 0 ? xk : (blk == 2 ? -q.x * xk : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = (i == c) ? 1.0 : 0.0;
    if (c >= 12) {
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] = 0.0;
    }
    // squared Frobenius norm of the group's matrix -> rounding floor
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) n2 = fma(a[i], a[i], n2);
    double fro2 = n2;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) fro2 += __shfl_xor(fro2, m, 64);
    const double floor2 = fro2 * 1e-32;

    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll 1
        for (int r = 0; r < 11; ++r) {
            // circle method: column 11 stays, the others meet the column at the mirrored position around r
            int partner;
            if (c >= 12) partner = c;
            else if (c == 11) partner = r;
            else if (c == r) partner = 11;
            else { partner = 2 * r - c; partner += partner < 0 ? 11 : 0; partner -= partner >= 11 ? 11 : 0; }
            const int src = base + partner;
            double pa[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) pa[i] = __shfl(a[i], src, 64);
            // squared norms and inner product of the pair; both lanes of a pair add the same numbers in the same order
            double own = 0.0, other = 0.0, spq = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                own = fma(a[i], a[i], own);
                other = fma(pa[i], pa[i], other);
                spq = fma(a[i], pa[i], spq);
            }
            // (lower column, higher column) order, the same on both lanes
            const bool low = c < partner;
            const double sp = low ? own : other, sq = low ? other : own;
            const bool go = c != partner && sp > floor2 && sq > floor2 && spq * spq > 1e-30 * (sp * sq);
            rotated |= go;
            if (__any(go)) {  // wave-uniform: a round in which nothing rotates costs the exchange of one column only
                double pv[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) pv[i] = __shfl(v[i], src, 64);
                double cs, sn;
                jacobi_angle(sp, sq, spq, cs, sn);
                cs = go ? cs : 1.0;
                sn = go ? sn : 0.0;
                // lower column: cs x - sn y ; higher column: sn x + cs y
                const double mine = cs, theirs = low ? -sn : sn;
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    a[i] = fma(mine, a[i], theirs * pa[i]);
                    v[i] = fma(mine, v[i], theirs * pv[i]);
                }
            }
        }
        if (!__any(rotated)) break;
    }
    // the column with the smallest norm; ties to the lower column
    n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) n2 = fma(a[i], a[i], n2);
    if (c >= 12) n2 = 1e300;
    double best = n2;
    int who = c;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const double ob = __shfl_xor(best, m, 64);
        const int ow = __shfl_xor(who, m, 64);
        const bool take = ob < best || (ob == best && ow < who);
        best = take ? ob : best;
        who = take ? ow : who;
    }
    if (owner && who == c) resect_finish(v, tw, hyp + (size_t)h * HP_LEN);  // vh[-1].reshape(3,4)   :113-114
}

// fills the call's work block: K prev_pose[:3], prev_pose[:3], K.  kp.A = K, kp.B unused; prev (12) by value too.
struct TrackSetup {
    double K[9];
    double prev[12];
};
__global__ __launch_bounds__(64) void k_track_setup(TrackSetup s, double* __restrict__ tw) {
    const int t = threadIdx.x;
    if (t < 12) {
        const int i = t >> 2, j = t & 3;
        tw[TW_P0 + t] = s.K[3 * i] * s.prev[j] + s.K[3 * i + 1] * s.prev[4 + j] + s.K[3 * i + 2] * s.prev[8 + j];
        tw[TW_PREV + t] = s.prev[t];
        tw[TW_WIN + t] = s.prev[t];
    }
    if (t < 9) tw[TW_K + t] = s.K[t];
}

// ------------------------------------------------------------------------------------------------
// K6: score one hypothesis per workgroup   (reference: src/triangulation.py:159-170)
//   triangulate the M known matches with (K prev_pose[:3], K pose[:3]); count points in front of both cameras;
//   error = sum of ||X_known - X||.  Fixed summation order: thread-strided partial sums, xor-shuffle tree, 4 waves.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_score(const double* __restrict__ tw, const double* __restrict__ hyp,
                                                     const double2* __restrict__ a, const double2* __restrict__ b,
                                                     const double* __restrict__ X3, int M, int32_t* __restrict__ counts,
                                                     double* __restrict__ errs) {
    __shared__ double s_err[4];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x;
    const double* __restrict__ hp = hyp + (size_t)blockIdx.x * HP_LEN;
    const double tx = hp[HP_T], ty = hp[HP_T + 1], tz = hp[HP_T + 2];
    const double cx = hp[HP_CZ], cy = hp[HP_CZ + 1], cz = hp[HP_CZ + 2];
    double err = 0.0;
    int cnt = 0;
    for (int i = tid; i < M; i += 256) {
        const double2 p = a[i], q = b[i];
        double X[3];
        triangulate_one(tw + TW_P0, hp + HP_P1, p.x, p.y, q.x, q.y, X);
        const bool f1 = X[2] > 0.0;                                                              // :167
        const bool f2 = ((X[0] - tx) * cx + (X[1] - ty) * cy + (X[2] - tz) * cz) > 0.0;          // :168
        cnt += (f1 && f2) ? 1 : 0;
        const double dx = X3[3 * (size_t)i] - X[0], dy = X3[3 * (size_t)i + 1] - X[1], dz = X3[3 * (size_t)i + 2] - X[2];
        err += sqrt(dx * dx + dy * dy + dz * dz);                                                // :170
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        err += __shfl_xor(err, m, 64);
        cnt += __shfl_xor(cnt, m, 64);
    }
    if ((tid & 63) == 0) { s_err[tid >> 6] = err; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        errs[blockIdx.x] = ((s_err[0] + s_err[1]) + s_err[2]) + s_err[3];
    }
}

// ------------------------------------------------------------------------------------------------
// K7: winner by the loop's rule   (reference: src/triangulation.py:172-176)
//   a later hypothesis replaces the incumbent iff it has more points in front, or as many and a strictly smaller
//   error.  Each lane scans a contiguous range in order, lane 0 then folds the 64 range winners in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_track_select(const int32_t* __restrict__ counts, const double* __restrict__ errs,
                                                     int H, const double* __restrict__ hyp, double* __restrict__ tw,
                                                     double* __restrict__ pose_out) {
    __shared__ int s_i[64];
    __shared__ int s_c[64];
    __shared__ double s_e[64];
    const int lane = threadIdx.x;
    const int per = (H + 63) / 64;
    const int lo = lane * per, hi = min(H, lo + per);
    int bi = -1, bc = -1;
    double be = 0.0;
    for (int i = lo; i < hi; ++i) {
        const int c = counts[i];
        const double e = errs[i];
        if (c > bc || (c == bc && be > e)) { bc = c; be = e; bi = i; }
    }
    s_i[lane] = bi; s_c[lane] = bc; s_e[lane] = be;
    __syncthreads();
    if (lane == 0) {
        for (int l = 1; l < 64; ++l) {
            if (s_i[l] < 0) continue;
            const int c = s_c[l];
            const double e = s_e[l];
            if (c > bc || (c == bc && be > e)) { bc = c; be = e; bi = s_i[l]; }
        }
        s_i[0] = bi;
        tw[TW_BEST] = (double)bi;
        tw[TW_BEST + 1] = (double)bc;
        tw[TW_BEST + 2] = be;
    }
    __syncthreads();
    const double* __restrict__ hp = hyp + (size_t)s_i[0] * HP_LEN;
    if (lane < 16) pose_out[lane] = hp[HP_POSE + lane];
    if (lane < 12) tw[TW_WIN + 12 + lane] = hp[HP_POSE + lane];
}

// triangulation whose camera pair sits in device memory (the tracking step's last call: src/triangulation.py:179-182)
__global__ __launch_bounds__(256) void k_triangulate_cams(const double* __restrict__ cams, const double2* __restrict__ p0,
                                                          const double2* __restrict__ p1, int64_t N,
                                                          double* __restrict__ Xout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double2 a = p0[i], b = p1[i];
    double X[3];
    triangulate_one(cams, cams + 12, a.x, a.y, b.x, b.y, X);
    Xout[3 * i + 0] = X[0];
    Xout[3 * i + 1] = X[1];
    Xout[3 * i + 2] = X[2];
}

}  // namespace slamk
