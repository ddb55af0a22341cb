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
__device__ inline void resect_finish(double (&cam)[12], const double* __restrict__ K, double* __restrict__ hp) {
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

    resect_finish(cam, tw + TW_K, hyp + (size_t)h * HP_LEN);
}

// ------------------------------------------------------------------------------------------------
// K5b: resection of six-point samples, sixteen lanes per hypothesis (four hypotheses per wave): lane c of a group
// holds column c of the 12x12 design matrix; lanes 12..15 of a group carry zeros.
//
// The least right singular vector comes from a block inverse iteration on the triangular factor:
//   1. Householder QR across the lanes (lane j forms reflector j, DPP row broadcasts carry it to the group, lane c > j
//      applies it to its column): lane c ends with column c of R; an LDS transpose gives lane i row i as well.
//   2. Two vectors x, y (one component per lane) go through (R^T R)^-1 = two triangular solves per step - at step j
//      lane j's component is final and is broadcast, the other lanes subtract their multiple - and are orthonormalised
//      after every step.  The pair converges to the span of the two least singular vectors at the rate
//      (sigma_11 / sigma_10)^2 per step (0.01 at the 90th percentile of six-point samples, 0.27 at worst), where a single
//      vector would converge at (sigma_12 / sigma_11)^2 (0.43 at the 99th percentile, 0.92 at worst).
//   3. One rotation of R [x y] (one-sided Jacobi, 2 columns) separates the least vector inside the span.
// Orthogonal transforms and triangular solves only: the Gram matrix, whose condition number would be the square of
// 1e5..1e7, never appears.  A wave in which some group's vector fails the residual test at the end (or is not finite)
// repeats the work with the one-sided Jacobi SVD below and keeps that result for those groups.
// ------------------------------------------------------------------------------------------------
#ifndef SLAM_RESECT_STEPS
#define SLAM_RESECT_STEPS 16
#endif

template <int J>
__device__ __forceinline__ double row16_bcast(double x) {  // x of lane J of each row of 16 lanes
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + J, 0xF, 0xF, false);  // row_newbcast:J
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + J, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row16_sum(double v) {  // sum over the row of 16 lanes, on every lane, fixed order
    v += dpp_f64<0xB1>(v);   // lane ^ 1
    v += dpp_f64<0x4E>(v);   // lane ^ 2
    v += dpp_f64<0x141>(v);  // 7 - l within 8
    v += dpp_f64<0x140>(v);  // 15 - l within 16
    return v;
}

template <int J>
__device__ __forceinline__ void resect_qr_step(double (&a)[12], int c) {
    double sigma = 0.0;
#pragma unroll
    for (int r = J + 1; r < 12; ++r) sigma = fma(a[r], a[r], sigma);
    const double x0 = a[J];
    const double s2 = fma(x0, x0, sigma);
    const double normx = (s2 > 0.0) ? s2 * rsqrt(s2) : 0.0;
    const double alpha = (x0 >= 0.0) ? -normx : normx;
    const double v0 = x0 - alpha;  // same sign as x0: no cancellation
    const double vtv = fma(v0, v0, sigma);
    const double bj = (vtv > 0.0) ? 2.0 * fast_rcp(vtv) : 0.0;
    const double rv0 = row16_bcast<J>(v0), rb = row16_bcast<J>(bj);
    double tail[12];
#pragma unroll
    for (int r = J + 1; r < 12; ++r) tail[r] = row16_bcast<J>(a[r]);
    if (c > J) {
        double s = rv0 * a[J];
#pragma unroll
        for (int r = J + 1; r < 12; ++r) s = fma(tail[r], a[r], s);
        s *= rb;
        a[J] = fma(-s, rv0, a[J]);
#pragma unroll
        for (int r = J + 1; r < 12; ++r) a[r] = fma(-s, tail[r], a[r]);
    }
    if (c == J) a[J] = alpha;
}

// R w = z for two right-hand sides at once, one component per lane: rowz[j] = R[i][j] for j > i (0 otherwise) on lane i
template <int J>
__device__ __forceinline__ void resect_back_step(const double (&rowz)[12], double rd, double& ax, double& ay) {
    const double bx = row16_bcast<J>(ax * rd), by = row16_bcast<J>(ay * rd);
    ax = fma(-rowz[J], bx, ax);
    ay = fma(-rowz[J], by, ay);
    if constexpr (J > 0) resect_back_step<J - 1>(rowz, rd, ax, ay);
}
// R^T z = x likewise: colz[j] = R[j][i] for j < i (0 otherwise) on lane i
template <int J>
__device__ __forceinline__ void resect_fwd_step(const double (&colz)[12], double rd, double& ax, double& ay) {
    const double bx = row16_bcast<J>(ax * rd), by = row16_bcast<J>(ay * rd);
    ax = fma(-colz[J], bx, ax);
    ay = fma(-colz[J], by, ay);
    if constexpr (J < 11) resect_fwd_step<J + 1>(colz, rd, ax, ay);
}

__device__ __forceinline__ void resect_orthonormalise(double& x, double& y) {
    const double xx = row16_sum(x * x);
    x *= rsqrt(xx);
    const double xy = row16_sum(x * y);
    y = fma(-xy, x, y);
    const double yy = row16_sum(y * y);
    y *= rsqrt(yy);
}

// a: this lane's column of the design matrix (destroyed).  cam: the least right singular vector, on every lane of the
// group.  Returns false for a group whose result is not to be trusted.
__device__ __forceinline__ bool resect6_block_iteration(double (&a)[12], int c, double (*s_R)[16], double (&cam)[12]) {
    resect_qr_step<0>(a, c); resect_qr_step<1>(a, c); resect_qr_step<2>(a, c); resect_qr_step<3>(a, c);
    resect_qr_step<4>(a, c); resect_qr_step<5>(a, c); resect_qr_step<6>(a, c); resect_qr_step<7>(a, c);
    resect_qr_step<8>(a, c); resect_qr_step<9>(a, c); resect_qr_step<10>(a, c);
    // lane c: a[r] = R[r][c] for r <= c.  Rows through LDS: s_R[r][c]
#pragma unroll
    for (int r = 0; r < 12; ++r) s_R[r][c] = a[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int i = c < 12 ? c : 11;
    double rowz[12], colz[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const double rij = s_R[i][j];
        rowz[j] = (c < 12 && j > c) ? rij : 0.0;
        colz[j] = (c < 12 && j < c) ? a[j] : 0.0;
    }
    double diag = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) diag = (j == c) ? a[j] : diag;
    // A (numerically) zero diagonal entry - a rank-deficient sample, whose null vector is exactly what is wanted -
    // is replaced by a tiny one, as inverse iteration does with zero pivots: the solves then blow the null direction up
    // by 1e18 per solve instead of dividing by zero (or, with a reciprocal of 0, losing that direction altogether).
    const double d2 = row16_sum(diag * diag);
    const double tiny = 1e-18 * sqrt(d2);
    const double diag_used = (c < 12 && fabs(diag) < tiny) ? tiny : diag;
    const double rd = (diag_used != 0.0) ? fast_rcp(diag_used) : 0.0;

    double x = c < 12 ? 1.0 : 0.0, y = c < 12 ? ((c & 1) ? -1.0 : 1.0) : 0.0;
    resect_back_step<11>(rowz, rd, x, y);
    x *= rd; y *= rd;
    resect_orthonormalise(x, y);
#pragma unroll 1
    for (int k = 0; k < SLAM_RESECT_STEPS; ++k) {
        resect_fwd_step<0>(colz, rd, x, y);
        x *= rd; y *= rd;
        resect_back_step<11>(rowz, rd, x, y);
        x *= rd; y *= rd;
        resect_orthonormalise(x, y);
    }
    // B = R [x y], one component per lane
    double xr[12], yr[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) xr[j] = 0.0, yr[j] = 0.0;
    xr[0] = row16_bcast<0>(x); yr[0] = row16_bcast<0>(y);    xr[1] = row16_bcast<1>(x); yr[1] = row16_bcast<1>(y);
    xr[2] = row16_bcast<2>(x); yr[2] = row16_bcast<2>(y);    xr[3] = row16_bcast<3>(x); yr[3] = row16_bcast<3>(y);
    xr[4] = row16_bcast<4>(x); yr[4] = row16_bcast<4>(y);    xr[5] = row16_bcast<5>(x); yr[5] = row16_bcast<5>(y);
    xr[6] = row16_bcast<6>(x); yr[6] = row16_bcast<6>(y);    xr[7] = row16_bcast<7>(x); yr[7] = row16_bcast<7>(y);
    xr[8] = row16_bcast<8>(x); yr[8] = row16_bcast<8>(y);    xr[9] = row16_bcast<9>(x); yr[9] = row16_bcast<9>(y);
    xr[10] = row16_bcast<10>(x); yr[10] = row16_bcast<10>(y); xr[11] = row16_bcast<11>(x); yr[11] = row16_bcast<11>(y);
    double bx = diag * x, by = diag * y;
#pragma unroll
    for (int j = 0; j < 12; ++j) { bx = fma(rowz[j], xr[j], bx); by = fma(rowz[j], yr[j], by); }
    const double sa = row16_sum(bx * bx), sb = row16_sum(by * by), sc = row16_sum(bx * by);
    double cs = 1.0, sn = 0.0;
    if (sc * sc > 1e-30 * (sa * sb)) jacobi_angle(sa, sb, sc, cs, sn);
    // rotated pair: p' = cs p - sn q, q' = sn p + cs q; the shorter image belongs to the least singular vector
    const double bu = fma(cs, bx, -sn * by), bw = fma(sn, bx, cs * by);
    const double nu = row16_sum(bu * bu), nw = row16_sum(bw * bw);
    const bool first = nu <= nw;
    const double v = first ? fma(cs, x, -sn * y) : fma(sn, x, cs * y);
    // One more step for v alone, as refinement and as test: if v = v12 + e v11, the step leaves e rho with
    // rho = (sigma_12 / sigma_11)^2 ~ nv / nother, so v moves by e (1 - rho).  (A residual R^T R v - nv v would carry
    // rounding errors of the size of sigma_1^2, the Gram matrix' trouble again; a solve does not.)
    const double nv = first ? nu : nw, nother = first ? nw : nu;
    double v1 = v, dummy = 0.0;
    resect_fwd_step<0>(colz, rd, v1, dummy);
    v1 *= rd; dummy *= rd;
    resect_back_step<11>(rowz, rd, v1, dummy);
    v1 *= rd;
    const double n1 = row16_sum(v1 * v1);
    v1 *= rsqrt(n1);
    const double dv = v1 - v;
    const double moved = row16_sum(dv * dv);
    const double slack = 1.0 - nv / nother;
    cam[0] = row16_bcast<0>(v1); cam[1] = row16_bcast<1>(v1); cam[2] = row16_bcast<2>(v1); cam[3] = row16_bcast<3>(v1);
    cam[4] = row16_bcast<4>(v1); cam[5] = row16_bcast<5>(v1); cam[6] = row16_bcast<6>(v1); cam[7] = row16_bcast<7>(v1);
    cam[8] = row16_bcast<8>(v1); cam[9] = row16_bcast<9>(v1); cam[10] = row16_bcast<10>(v1); cam[11] = row16_bcast<11>(v1);
    const bool fine = (moved < 1e-16 * slack * slack) && (n1 > 0.0) && (n1 < 1e300);  // false for NaN as well
#ifdef SLAM_RESECT_DEBUG
    if (c == 0 && !fine) printf("not fine: block %d group %d moved %.3e slack %.3e nu %.3e nw %.3e\n", (int)blockIdx.x, (int)(threadIdx.x >> 4), moved, slack, nu, nw);
#endif
    return fine;
}

// The one-sided Jacobi SVD of the same matrix (the fallback, and the reference point for the iteration above): the
// columns rotate in tournament order - in each of the 11 rounds of a sweep the 12 columns form 6 disjoint pairs that
// rotate at the same time.  A lane fetches its partner's columns with ds_bpermute, both lanes of a pair derive the
// same angle from the same three sums, and each keeps its own side of the rotation.  Works on the matrix itself; sweeps
// end when no pair of the wave rotates.  cam: the right singular vector of the smallest singular value, on every lane
// of the group.
__device__ __forceinline__ void resect6_jacobi(double (&a)[12], int c, int base, double (&cam)[12]) {
    double v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = (i == c) ? 1.0 : 0.0;
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) n2 = fma(a[i], a[i], n2);
    double fro2 = n2;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) fro2 += __shfl_xor(fro2, m, 64);
    const double floor2 = fro2 * 1e-32;  // columns at rounding level are left alone

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
#pragma unroll
    for (int i = 0; i < 12; ++i) cam[i] = __shfl(v[i], base + who, 64);  // vh[-1]   :113-114
}

// column c of the design matrix of hypothesis hh: rows 2p = [0,0,0,0, -Xh, v Xh], rows 2p+1 = [Xh, 0,0,0,0, -u Xh]
__device__ __forceinline__ void resect6_column(const double2* __restrict__ uv, const double* __restrict__ X3,
                                               const int32_t* idx6, int c, double (&a)[12]) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int i = idx6[p];
        const double2 q = uv[i];
        const int k = c & 3;
        const double xk = k == 3 ? 1.0 : X3[3 * (size_t)i + k];
        const int blk = c >> 2;               // 0: first four columns, 1: middle, 2: last (3 for idle lanes)
        a[2 * p] = blk == 1 ? -xk : (blk == 2 ? q.y * xk : 0.0);
        a[2 * p + 1] = blk == 0 ? xk : (blk == 2 ? -q.x * xk : 0.0);
    }
    if (c >= 12) {
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] = 0.0;
    }
}

// The call's work block: K prev_pose[:3], prev_pose[:3], K, and prev_pose[:3] as the first camera of the final
// triangulation.  The intrinsics and the previous pose travel as a kernel argument.
struct TrackSetup {
    double K[9];
    double prev[12];
};
__device__ __forceinline__ void track_setup(const TrackSetup& s, double* __restrict__ tw, int t) {
    if (t < 12) {
        const int i = t >> 2, j = t & 3;
        tw[TW_P0 + t] = s.K[3 * i] * s.prev[j] + s.K[3 * i + 1] * s.prev[4 + j] + s.K[3 * i + 2] * s.prev[8 + j];
        tw[TW_PREV + t] = s.prev[t];
        tw[TW_WIN + t] = s.prev[t];
    }
    if (t < 9) tw[TW_K + t] = s.K[t];
}
__global__ __launch_bounds__(64) void k_track_setup(TrackSetup s, double* __restrict__ tw) { track_setup(s, tw, threadIdx.x); }

// setup: the kernels after this one read the work block tw; workgroup 0 fills it on the way (this kernel itself takes
// the intrinsics from the argument), which saves the tracking step a launch.
// Sampling: lane 0 of each group draws the group's six indices itself (random.seed(0x1337BEEF + it);
// random.sample(range(M), 6), src/triangulation.py:151-153, replayed from the iteration's row of the MT19937 table: the
// lane-sequential form of the sampler, as the sampling kernel has it), which saves the step another launch: the draw
// costs a fraction of a microsecond here and the front end's four as a kernel of its own.
__global__ __launch_bounds__(64) void k_resect6(const double2* __restrict__ uv, const double* __restrict__ X3, int M,
                                                const uint32_t* __restrict__ raw, int64_t raw_it0, int H,
                                                TrackSetup setup, double* __restrict__ tw, double* __restrict__ hyp,
                                                int32_t* __restrict__ flags) {
    __shared__ double s_R[4][12][16];
    __shared__ uint8_t s_pool[SLAM_POOL_MAX * 4];
    __shared__ uint32_t s_words[SLAM_PREFETCH_WORDS * 4];
    __shared__ int32_t s_idx[4][8];
    const int lane = threadIdx.x;
    if (blockIdx.x == 0) track_setup(setup, tw, lane);
    const int c = lane & 15;                  // column owned by this lane
    const int base = lane & 48;               // first lane of the group
    const int h = blockIdx.x * 4 + (lane >> 4);
    const int hh = h < H ? h : H - 1;
    if (c == 0) {
        int32_t out[SLAM_MAX_K];
        const bool ok = sample_cpython(raw + (size_t)(hh - raw_it0) * SLAM_RAW_WORDS, M, 6, out, s_words, s_pool, lane >> 4, 4);
#pragma unroll
        for (int p = 0; p < 6; ++p) s_idx[lane >> 4][p] = min(max(out[p], 0), M - 1);  // never outside the arrays
        if (!ok) raise_flag(flags);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a[12], cam[12];
    resect6_column(uv, X3, s_idx[lane >> 4], c, a);
    const bool fine = resect6_block_iteration(a, c, s_R[lane >> 4], cam);
    if (__any(!fine)) {  // wave-uniform; a group that was fine keeps its numbers
        double cam2[12];
        resect6_column(uv, X3, s_idx[lane >> 4], c, a);
        resect6_jacobi(a, c, base, cam2);
#pragma unroll
        for (int i = 0; i < 12; ++i) cam[i] = fine ? cam[i] : cam2[i];
    }
    if (c == 0 && h < H) resect_finish(cam, setup.K, hyp + (size_t)h * HP_LEN);  // vh[-1].reshape(3,4)   :113-114
}

// ------------------------------------------------------------------------------------------------
// K6: score one hypothesis per workgroup   (reference: src/triangulation.py:159-170)
//   triangulate the M known matches with (K prev_pose[:3], K pose[:3]); count points in front of both cameras;
//   error = sum of ||X_known - X||.  Fixed summation order (a function of M only): thread-strided partial sums,
//   xor-shuffle tree, the waves in order.
// ------------------------------------------------------------------------------------------------
#define SLAM_TRACK_SCORE_MAX_THREADS 512
__global__ __launch_bounds__(SLAM_TRACK_SCORE_MAX_THREADS) void k_track_score(
    const double* __restrict__ tw, const double* __restrict__ hyp, const double2* __restrict__ a,
    const double2* __restrict__ b, const double* __restrict__ X3, int M, int32_t* __restrict__ counts,
    double* __restrict__ errs) {
    // The workgroup has as many waves as the M matches fill, eight at most: a thread's triangulations are one
    // dependent chain each, so the kernel lasts as long as the thread with the most matches.
    __shared__ double s_err[SLAM_TRACK_SCORE_MAX_THREADS / 64];
    __shared__ int s_cnt[SLAM_TRACK_SCORE_MAX_THREADS / 64];
    const int tid = threadIdx.x, T = blockDim.x;
    const double* __restrict__ hp = hyp + (size_t)blockIdx.x * HP_LEN;
    const double tx = hp[HP_T], ty = hp[HP_T + 1], tz = hp[HP_T + 2];
    const double cx = hp[HP_CZ], cy = hp[HP_CZ + 1], cz = hp[HP_CZ + 2];
    double err = 0.0;
    int cnt = 0;
    for (int i = tid; i < M; i += T) {
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
        int c = s_cnt[0];
        double e = s_err[0];
        for (int w = 1; w < T / 64; ++w) { c += s_cnt[w]; e += s_err[w]; }
        counts[blockIdx.x] = c;
        errs[blockIdx.x] = e;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: winner by the loop's rule   (reference: src/triangulation.py:172-176)
//   a later hypothesis replaces the incumbent iff it has more points in front, or as many and a strictly smaller
//   error.  Each lane scans a contiguous range in order, lane 0 then folds the 64 range winners in order.
// ------------------------------------------------------------------------------------------------
// (the first 64 threads of the workgroup take part; the others wait at the barriers; returns the winner's position)
__device__ __forceinline__ int track_select(const int32_t* __restrict__ counts, const double* __restrict__ errs, int H,
                                            int* s_i, int* s_c, double* s_e, bool write, double* __restrict__ tw) {
    const int lane = threadIdx.x;
    if (lane < 64) {
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
    }
    __syncthreads();
    if (lane == 0) {
        int bi = s_i[0], bc = s_c[0];
        double be = s_e[0];
        for (int l = 1; l < 64; ++l) {
            if (s_i[l] < 0) continue;
            const int c = s_c[l];
            const double e = s_e[l];
            if (c > bc || (c == bc && be > e)) { bc = c; be = e; bi = s_i[l]; }
        }
        s_i[0] = bi;
        if (write) {
            tw[TW_BEST] = (double)bi;
            tw[TW_BEST + 1] = (double)bc;
            tw[TW_BEST + 2] = be;
        }
    }
    __syncthreads();
    return s_i[0];
}

__global__ __launch_bounds__(64) void k_track_select(const int32_t* __restrict__ counts, const double* __restrict__ errs,
                                                     int H, const double* __restrict__ hyp, double* __restrict__ tw,
                                                     double* __restrict__ pose_out) {
    __shared__ int s_i[64];
    __shared__ int s_c[64];
    __shared__ double s_e[64];
    const int lane = threadIdx.x;
    const int win = track_select(counts, errs, H, s_i, s_c, s_e, true, tw);
    const double* __restrict__ hp = hyp + (size_t)win * HP_LEN;
    if (lane < 16) pose_out[lane] = hp[HP_POSE + lane];
    if (lane < 12) tw[TW_WIN + 12 + lane] = hp[HP_POSE + lane];
}

// K7 and the final triangulation in one launch (few hypotheses): every workgroup finds the winner for itself (12 bytes
// per hypothesis) and triangulates its 256 matches with (prev_pose[:3], winning pose[:3])   (src/triangulation.py:
// 172-182); workgroup 0 writes the winner's record and pose.  N may be 0 (one workgroup, selection only).
__global__ __launch_bounds__(256) void k_track_finish(const int32_t* __restrict__ counts, const double* __restrict__ errs,
                                                      int H, const double* __restrict__ hyp, double* __restrict__ tw,
                                                      double* __restrict__ pose_out, const double2* __restrict__ p0,
                                                      const double2* __restrict__ p1, int64_t N,
                                                      double* __restrict__ Xout) {
    __shared__ int s_i[64];
    __shared__ int s_c[64];
    __shared__ double s_e[64];
    __shared__ double s_cams[24];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool live = i < N;
    double2 a = make_double2(0.0, 0.0), b = a;
    if (live) { a = p0[i]; b = p1[i]; }  // in flight during the selection
    const int win = track_select(counts, errs, H, s_i, s_c, s_e, blockIdx.x == 0, tw);
    const double* __restrict__ hp = hyp + (size_t)win * HP_LEN;
    if (tid < 12) s_cams[tid] = tw[TW_WIN + tid];
    else if (tid < 24) s_cams[tid] = hp[HP_POSE + tid - 12];
    if (blockIdx.x == 0) {
        if (tid < 16) pose_out[tid] = hp[HP_POSE + tid];
        if (tid < 12) tw[TW_WIN + 12 + tid] = hp[HP_POSE + tid];
    }
    __syncthreads();
    if (!live) return;
    double X[3];
    triangulate_one(s_cams, s_cams + 12, a.x, a.y, b.x, b.y, X, SLAM_TRI_ANGLE_TOL_TIGHT);
    Xout[3 * i + 0] = X[0];
    Xout[3 * i + 1] = X[1];
    Xout[3 * i + 2] = X[2];
}

// triangulation whose camera pair sits in device memory (the tracking step's last call: src/triangulation.py:179-182)
__global__ __launch_bounds__(256) void k_triangulate_cams(const double* __restrict__ cams, const double2* __restrict__ p0,
                                                          const double2* __restrict__ p1, int64_t N,
                                                          double* __restrict__ Xout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double2 a = p0[i], b = p1[i];
    double X[3];
    triangulate_one(cams, cams + 12, a.x, a.y, b.x, b.y, X, SLAM_TRI_ANGLE_TOL_TIGHT);
    Xout[3 * i + 0] = X[0];
    Xout[3 * i + 1] = X[1];
    Xout[3 * i + 2] = X[2];
}

}  // namespace slamk
