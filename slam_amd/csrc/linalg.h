// Register-resident small dense linear algebra for one lane (float64), gfx950.
// Everything is fully unrolled with compile-time indices so that the arrays live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifndef SLAM_CUBIC_NEWTON_STEPS
#define SLAM_CUBIC_NEWTON_STEPS 9
#endif
#ifndef SLAM_QUARTIC_NEWTON_STEPS
#define SLAM_QUARTIC_NEWTON_STEPS 8
#endif
#ifndef SLAM_RANK2_TOL
#define SLAM_RANK2_TOL 1e-15
#endif
namespace slamk {
// The context's error flags live in page-locked host memory that the device writes directly (never, in practice: the
// one flag says that a sampler ran out of pre-generated words), so that the host reads them after a synchronisation
// without a copy.  A plain store of the only value the word ever takes: nothing is read back, so no atomic has to
// cross the host link.
__device__ __forceinline__ void raise_flag(int32_t* flags) { *(volatile int32_t*)flags = 1; }
#ifdef SLAM_STAMPS
__device__ unsigned long long g_dbg[8];  // sweep statistics of the rank-2 projection; counted only with SLAM_COUNT_SWEEPS
#endif
#if defined(SLAM_STAMPS) && defined(SLAM_COUNT_SWEEPS)
#define SLAM_DBG_ADD(slot, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dbg[slot], (unsigned long long)(v)); } while (0)  // per wave
#else
#define SLAM_DBG_ADD(slot, v) do { } while (0)
#endif

// Plane rotation that orthogonalises two columns with squared norms a, b and inner product c.
// Small-angle branch (|theta| <= pi/4):  cos 2t = |b-a|/h, sin 2t = sign(b-a) 2c/h, h = hypot(b-a, 2c).
// Uses two reciprocal square roots and no division.
__device__ __forceinline__ void jacobi_angle(double a, double b, double c, double& cs, double& sn) {
    const double d = b - a, c2 = c + c;
    const double rh = rsqrt(d * d + c2 * c2);
    const double cos2 = fabs(d) * rh;
    const double sin2 = (d >= 0.0 ? c2 : -c2) * rh;
    const double half = 0.5 + 0.5 * cos2;
    const double rc = rsqrt(half);
    cs = half * rc;
    sn = 0.5 * sin2 * rc;
}

// 1/x to about one ulp: hardware estimate plus two Newton steps (no IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// One-sided (Hestenes) Jacobi sweeps on an M x N matrix held as A[row][col], accumulating the rotations into V
// (which the caller initialises, normally to the identity).
// On return A = A_in * R and V = V_in * R for the accumulated rotation R.
//
// A pair of columns with squared norms a, b and inner product c is rotated while it is not orthogonal to rounding
// (c^2 > 1e-30 a b) AND the rotation would still turn the right vectors by more than angle_tol (|c| > angle_tol
// |a - b|, since the angle is about c / (b - a)).  Columns whose squared norm is below 1e-32 ||A||_F^2 are
// treated as converged (rounding-level null columns).
//
// LAST_ONLY = false: every pair must converge; the columns of A end up mutually orthogonal (column j = sigma_j u_j
//   when V_in was the identity).
// LAST_ONLY = true: only the LAST column has to become orthogonal to the others, which makes V's last column a
//   right singular vector (the least one when the caller started near it); the other pairs are rotated only
//   while that is not yet the case, as part of the same cyclic sweep.  Lanes with `full` set (no good start)
//   still converge every pair.
template <int M, int N, bool LAST_ONLY>
__device__ __forceinline__ void jacobi_sweeps(double (&A)[M][N], double (&V)[N][N], int max_sweeps, double angle_tol,
                                              bool full = false) {
    double fro2 = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) fro2 = fma(A[i][j], A[i][j], fro2);
    const double floor2 = fro2 * 1e-32;

    auto rotate_pair = [&](auto P, auto Q, bool allowed) -> bool {
        constexpr int p = decltype(P)::value, q = decltype(Q)::value;
        double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            a = fma(A[i][p], A[i][p], a);
            b = fma(A[i][q], A[i][q], b);
            c = fma(A[i][p], A[i][q], c);
        }
        const bool go = allowed && (a > floor2) && (b > floor2) && (c * c > 1e-30 * (a * b)) &&
                        (fabs(c) > angle_tol * fabs(a - b));
        if (__any(go)) {  // wave-uniform: a converged wave only pays for the three inner products
            double cs, sn;
            jacobi_angle(a, b, c, cs, sn);
            cs = go ? cs : 1.0;
            sn = go ? sn : 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double x = A[i][p], y = A[i][q];
                A[i][p] = fma(cs, x, -sn * y);
                A[i][q] = fma(sn, x, cs * y);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double x = V[i][p], y = V[i][q];
                V[i][p] = fma(cs, x, -sn * y);
                V[i][q] = fma(sn, x, cs * y);
            }
        }
        return go;
    };

    static_assert(N == 3 || N == 4, "pairs are enumerated by hand for 3 and 4 columns");
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // does a pair still need a rotation?  (the three inner products and the tests of rotate_pair, nothing else)
    auto needs_pair = [&](auto P, auto Q) -> bool {
        constexpr int p = decltype(P)::value, q = decltype(Q)::value;
        double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            a = fma(A[i][p], A[i][p], a);
            b = fma(A[i][q], A[i][q], b);
            c = fma(A[i][p], A[i][q], c);
        }
        return (a > floor2) && (b > floor2) && (c * c > 1e-30 * (a * b)) && (fabs(c) > angle_tol * fabs(a - b));
    };

    if constexpr (LAST_ONLY && N == 3) {
        // A sweep starts by looking at the two pairs with the last column; when neither needs a rotation the lane is
        // done.  Otherwise the other pair goes FIRST: with columns 0 and 1 orthogonal, rotating each against the last
        // column is a proper cyclic sweep whose errors square; with them oblique, the second rotation re-opens the
        // first pair's angle to first order and costs another rotating sweep.
        // Before any rotation: when the last column is off by a tiny angle only (the usual state after a warm start),
        // one joint first-order step makes it orthogonal to both other columns at once.  With a_j = |b_j|^2 and
        // c_ij = b_i . b_j, the step b2 -= t0 b0 + t1 b1, b0 += t0 b2, b1 += t1 b2 (the same on V) needs
        //   t0 (a0 - a2) + t1 c01 = c02,   t0 c01 + t1 (a1 - a2) = c12,
        // which does not ask for columns 0 and 1 to be orthogonal, and leaves errors of second order: it is taken
        // when |t0|, |t1| <= 1e-8 (then t^2 is below rounding).  Every lane decides for itself; the sweeps below
        // start with a look at the result and end at once for a lane that is done.
        {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, c01 = 0.0, c02 = 0.0, c12 = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                a0 = fma(A[i][0], A[i][0], a0);
                a1 = fma(A[i][1], A[i][1], a1);
                a2 = fma(A[i][2], A[i][2], a2);
                c01 = fma(A[i][0], A[i][1], c01);
                c02 = fma(A[i][0], A[i][2], c02);
                c12 = fma(A[i][1], A[i][2], c12);
            }
            const bool need0 = (a0 > floor2) && (a2 > floor2) && (c02 * c02 > 1e-30 * (a0 * a2)) && (fabs(c02) > angle_tol * fabs(a0 - a2));
            const bool need1 = (a1 > floor2) && (a2 > floor2) && (c12 * c12 > 1e-30 * (a1 * a2)) && (fabs(c12) > angle_tol * fabs(a1 - a2));
            if (!__any(need0 || need1 || full)) return;
            const double d0 = a0 - a2, d1 = a1 - a2;
            const double det = fma(d0, d1, -c01 * c01);
            const double inv = fast_rcp(det);
            double t0 = fma(c02, d1, -c12 * c01) * inv, t1 = fma(c12, d0, -c02 * c01) * inv;
            const bool tiny = (need0 || need1) && !full && det > 0.0 && fabs(t0) <= 1e-8 && fabs(t1) <= 1e-8;
#if defined(SLAM_STAMPS) && defined(SLAM_COUNT_SWEEPS)
            if (M == 3 && (need0 || need1) && !tiny) {  // per lane: why the first-order step was not taken
                const double tm = fmax(fabs(t0), fabs(t1));
                atomicAdd(&g_dbg[!(det > 0.0) ? 3 : (tm <= 1e-4 ? 1 : 2)], 1ull);
            }
#endif
            if (__any(tiny)) {
                t0 = tiny ? t0 : 0.0;  // other lanes: an exact no-op
                t1 = tiny ? t1 : 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const double x0 = A[i][0], x1 = A[i][1], x2 = A[i][2];
                    A[i][2] = fma(-t1, x1, fma(-t0, x0, x2));
                    A[i][0] = fma(t0, x2, x0);
                    A[i][1] = fma(t1, x2, x1);
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double x0 = V[i][0], x1 = V[i][1], x2 = V[i][2];
                    V[i][2] = fma(-t1, x1, fma(-t0, x0, x2));
                    V[i][0] = fma(t0, x2, x0);
                    V[i][1] = fma(t1, x2, x1);
                }
            }
        }
        bool full_active = full;
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            if (M == 3) SLAM_DBG_ADD(4, 1);
            const bool need0 = needs_pair(I0{}, I2{});
            const bool need1 = needs_pair(I1{}, I2{});
            const bool more = need0 || need1 || full_active;
            if (!__any(more)) { if (M == 3) SLAM_DBG_ADD(5 + (sweep > 2 ? 2 : sweep), 1); break; }
            const bool r01 = rotate_pair(I0{}, I1{}, more);
            const bool r02 = rotate_pair(I0{}, I2{}, true);
            const bool r12 = rotate_pair(I1{}, I2{}, true);
            full_active = full && (r01 || r02 || r12);
        }
    } else {
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            bool rotated = false;
            if constexpr (LAST_ONLY) {  // N == 4
                // pairs with the last column first; the rest only for lanes that are not done
                rotated |= rotate_pair(I0{}, I3{}, true);
                rotated |= rotate_pair(I1{}, I3{}, true);
                rotated |= rotate_pair(I2{}, I3{}, true);
                const bool more = rotated || full;
                if (!__any(more)) break;
                bool others = rotate_pair(I0{}, I1{}, more);
                others |= rotate_pair(I0{}, I2{}, more);
                others |= rotate_pair(I1{}, I2{}, more);
                rotated |= full && others;
            } else {
                rotated |= rotate_pair(I0{}, I1{}, true);
                rotated |= rotate_pair(I0{}, I2{}, true);
                if constexpr (N == 4) rotated |= rotate_pair(I0{}, I3{}, true);
                rotated |= rotate_pair(I1{}, I2{}, true);
                if constexpr (N == 4) {
                    rotated |= rotate_pair(I1{}, I3{}, true);
                    rotated |= rotate_pair(I2{}, I3{}, true);
                }
            }
            // leave when no lane of the wave still rotates (keeps the wave converged together)
            if (!__any(rotated)) break;
        }
    }
}

template <int M, int N>
__device__ __forceinline__ void jacobi_svd(double (&A)[M][N], double (&V)[N][N], int max_sweeps) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    jacobi_sweeps<M, N, false>(A, V, max_sweeps, 1e-15);
}

// Cross-lane helpers of the cooperative solve (eight lanes per system, eight systems per wave; all 64 lanes active).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

template <int J>
__device__ __forceinline__ double bcast8(double x) {  // x of lane J of each group of eight lanes
    constexpr int pattern = (J << 5) | 0x18;          // bit mode: lane' = (lane & 0b11000) | J
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), pattern);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), pattern);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bperm_f64(int byte_addr, double x) {  // x of lane byte_addr / 4
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ uint32_t group8_max_u32(uint32_t k) {  // maximum over the eight lanes of a group, on all of them
    k = max(k, dpp_u32<0xB1>(k));   // lane ^ 1
    k = max(k, dpp_u32<0x4E>(k));   // lane ^ 2
    k = max(k, dpp_u32<0x141>(k));  // 7 - l within 8
    return k;
}

// Null vector of the 8 x 9 design matrix by EIGHT lanes per system: lane i of a group of eight consecutive lanes holds
// row i (a[0..8], a[8] = 1).  Gauss-Jordan elimination with row pivoting over the columns 8, 0, 1, ..., 6; column 7 is
// the free one, so the result is n with n[7] = 1 (NOT normalised; the projection that follows normalises).
//   * Column 8 is all ones: row 0 is its pivot and every multiplier is 1 (rows 1..7 become differences).
//   * Column c: the pivot is the unused row with the largest |a[c]|, compared on the high word of the double with its
//     three lowest bits replaced by 7 - row (a pivot within 2^-17 of the largest serves as well, and the keys of the
//     unused rows are distinct, so exactly one lane is the pivot); the maximum travels by three DPP exchanges.
//     The pivot row and the reciprocal of its entry reach the group through ds_bpermute; every row but the pivot -
//     rows used earlier included, which is what makes the last step a read-out instead of a back substitution -
//     subtracts its multiple.
//   * After the eight steps row i holds its pivot entry and a[7] only: n[column of row i] = -a[7] / pivot.
// Elementwise arithmetic only (no sums across lanes), backward error of the order of the rounding unit times the
// condition number, like the SVD the reference calls (measured on 10^4 samples: at most 1.4 eps cond).
// A zero pivot (rank-deficient sample) leaves its column out (multiplier 0, n = 0 there): finite garbage where the
// reference returns an arbitrary null vector.
template <int C>
__device__ __forceinline__ void null8x9_step(double (&a)[9], int i, int group_addr, bool& used, double& rsave, int& addr_out) {
    const double x = a[C];
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const uint32_t key = used ? 0u : ((hi & 0x7ffffff8u) | (uint32_t)(7 - i) | 0x80000000u);
    const uint32_t mk = group8_max_u32(key);
    const bool is_p = key == mk;
    const int addr = group_addr | (int)((7u - (mk & 7u)) << 2);
    const double rinv_own = (x != 0.0) ? fast_rcp(x) : 0.0;
    const double rinv = bperm_f64(addr, rinv_own);
    double piv[8];
#pragma unroll
    for (int j = C + 1; j < 8; ++j) piv[j] = bperm_f64(addr, a[j]);
    const double f = is_p ? 0.0 : x * rinv;
#pragma unroll
    for (int j = C + 1; j < 8; ++j) a[j] = fma(-f, piv[j], a[j]);
    rsave = is_p ? rinv_own : rsave;
    used = used || is_p;
    addr_out = addr;
}

__device__ __forceinline__ void null_vector_8x9_by_eight_lanes(double (&a)[9], double (&n)[9]) {
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int i = lane & 7, group_addr = (lane & ~7) << 2;
    {
        double piv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) piv[j] = bcast8<0>(a[j]);
        const double f = (i == 0) ? 0.0 : 1.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(-f, piv[j], a[j]);
    }
    bool used = (i == 0);
    double rsave = 1.0;
    int addr[7];
    null8x9_step<0>(a, i, group_addr, used, rsave, addr[0]);
    null8x9_step<1>(a, i, group_addr, used, rsave, addr[1]);
    null8x9_step<2>(a, i, group_addr, used, rsave, addr[2]);
    null8x9_step<3>(a, i, group_addr, used, rsave, addr[3]);
    null8x9_step<4>(a, i, group_addr, used, rsave, addr[4]);
    null8x9_step<5>(a, i, group_addr, used, rsave, addr[5]);
    null8x9_step<6>(a, i, group_addr, used, rsave, addr[6]);
    const double xo = -a[7] * rsave;
#pragma unroll
    for (int c = 0; c < 7; ++c) n[c] = bperm_f64(addr[c], xo);
    n[7] = 1.0;
    n[8] = bcast8<0>(xo);
}

// The same elimination by ONE lane per system (A[r][c], r = 0..7 rows, c = 0..8 columns, A[r][8] = 1): every
// arithmetic operation of the eight-lane version is done once per row with the same operands, in the same order per
// matrix entry, and the pivot is chosen by the same integer keys, so the null vector has the same bits.  What the
// other version does with lane exchanges is done here with selections over the rows (the pivot row is the one-hot
// sum of the rows).  Throughput form: 64 systems per wave.
__device__ __forceinline__ void null_vector_8x9_one_lane(double (&A)[8][9], double (&n)[9]) {
    {
        double piv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) piv[j] = A[0][j];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double f = (r == 0) ? 0.0 : 1.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) A[r][j] = fma(-f, piv[j], A[r][j]);
        }
    }
    bool used[8];
    double rsave[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { used[r] = (r == 0); rsave[r] = 1.0; }
    int prow[7];
    auto step = [&](auto CC) {
        constexpr int C = decltype(CC)::value;
        uint32_t key[8], mk = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t hi = (uint32_t)__double2hiint(A[r][C]);
            key[r] = used[r] ? 0u : ((hi & 0x7ffffff8u) | (uint32_t)(7 - r) | 0x80000000u);
            mk = max(mk, key[r]);
        }
        bool isp[8];
        double xp = 0.0, piv[8];
#pragma unroll
        for (int j = C + 1; j < 8; ++j) piv[j] = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            isp[r] = key[r] == mk;
            xp = isp[r] ? A[r][C] : xp;
#pragma unroll
            for (int j = C + 1; j < 8; ++j) piv[j] = isp[r] ? A[r][j] : piv[j];
        }
        const double rinv = (xp != 0.0) ? fast_rcp(xp) : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double f = isp[r] ? 0.0 : A[r][C] * rinv;
#pragma unroll
            for (int j = C + 1; j < 8; ++j) A[r][j] = fma(-f, piv[j], A[r][j]);
            rsave[r] = isp[r] ? rinv : rsave[r];
            used[r] = used[r] || isp[r];
        }
        prow[C] = 7 - (int)(mk & 7u);
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    double xo[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) xo[r] = -A[r][7] * rsave[r];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        double v = xo[0];
#pragma unroll
        for (int r = 1; r < 8; ++r) v = (prow[c] == r) ? xo[r] : v;
        n[c] = v;
    }
    n[7] = 1.0;
    n[8] = xo[0];
}

// Closest rank-2 matrix (Frobenius) of f / ||f||_F :  U diag(s0, s1, 0) V^T.
// With B = f V (columns sigma_i u_i) this is the sum over the two largest columns of B_i v_i^T.
//
// The Jacobi iteration is started from a basis that already contains a good estimate of the least right singular
// vector v3, so that it converges in one sweep (two for the odd badly separated case) instead of six or seven:
//   M = f^T f;  lambda = smallest root of det(M - x I), by Newton's iteration from x = 0 (monotone from below,
//   because the characteristic cubic is positive, decreasing and convex left of its smallest root);
//   adj(M - lambda I) = (l1 - lambda)(l2 - lambda) v3 v3^T, so its largest column is parallel to v3.
// The estimate only has to be good to a few digits: the sweeps that follow work on f itself and restore full
// accuracy (they do not see the squared condition number of M).
__device__ __forceinline__ void rank2_project_sweeps(const double (&f_in)[9], double (&F)[9]) {
    double nrm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nrm2 = fma(f_in[i], f_in[i], nrm2);
    const double inv = rsqrt(nrm2);
    double f[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[r][c] = f_in[3 * r + c] * inv;

    // M = f^T f (symmetric)
    const double m00 = f[0][0] * f[0][0] + f[1][0] * f[1][0] + f[2][0] * f[2][0];
    const double m01 = f[0][0] * f[0][1] + f[1][0] * f[1][1] + f[2][0] * f[2][1];
    const double m02 = f[0][0] * f[0][2] + f[1][0] * f[1][2] + f[2][0] * f[2][2];
    const double m11 = f[0][1] * f[0][1] + f[1][1] * f[1][1] + f[2][1] * f[2][1];
    const double m12 = f[0][1] * f[0][2] + f[1][1] * f[1][2] + f[2][1] * f[2][2];
    const double m22 = f[0][2] * f[0][2] + f[1][2] * f[1][2] + f[2][2] * f[2][2];
    const double c2 = m00 + m11 + m22;
    const double c1 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12);
    const double c0 = m00 * (m11 * m22 - m12 * m12) - m01 * (m01 * m22 - m12 * m02) + m02 * (m01 * m12 - m11 * m02);
    double lam = 0.0;
#pragma unroll
    for (int it = 0; it < SLAM_CUBIC_NEWTON_STEPS; ++it) {
        const double p = fma(fma(c2 - lam, lam, -c1), lam, c0);
        const double dp = fma(fma(-3.0, lam, c2 + c2), lam, -c1);
        lam = fma(-p, __builtin_amdgcn_rcp(dp), lam);  // an approximate reciprocal is enough: the iteration self-corrects
    }
    const double a00 = m00 - lam, a11 = m11 - lam, a22 = m22 - lam;
    const double g00 = a11 * a22 - m12 * m12, g11 = a00 * a22 - m02 * m02, g22 = a00 * a11 - m01 * m01;
    const double g01 = m02 * m12 - m01 * a22, g02 = m01 * m12 - m02 * a11, g12 = m01 * m02 - a00 * m12;
    // adj is (a multiple of) v3 v3^T: its largest diagonal entry marks its largest column
    const bool k0 = (g00 >= g11) && (g00 >= g22);
    const bool k1 = !k0 && (g11 >= g22);
    double v0 = k0 ? g00 : (k1 ? g01 : g02);
    double v1 = k0 ? g01 : (k1 ? g11 : g12);
    double v2 = k0 ? g02 : (k1 ? g12 : g22);
    const double vn2 = v0 * v0 + v1 * v1 + v2 * v2;
    const bool usable = (vn2 > 0.0) && (vn2 < 1e300);  // false for NaN, zero or overflow: start from the identity
    const double vi = rsqrt(vn2);
    v0 *= vi; v1 *= vi; v2 *= vi;
    // a = v x e_k with k the smallest |v_k| (best conditioned), b = v x a
    const double q0 = fabs(v0), q1 = fabs(v1), q2 = fabs(v2);
    const bool s0 = (q0 <= q1) && (q0 <= q2);
    const bool s1 = !s0 && (q1 <= q2);
    double a0 = s0 ? 0.0 : (s1 ? -v2 : v1);
    double a1 = s0 ? v2 : (s1 ? 0.0 : -v0);
    double a2 = s0 ? -v1 : (s1 ? v0 : 0.0);
    const double ai = rsqrt(a0 * a0 + a1 * a1 + a2 * a2);
    a0 *= ai; a1 *= ai; a2 *= ai;
    const double b0 = v1 * a2 - v2 * a1, b1 = v2 * a0 - v0 * a2, b2 = v0 * a1 - v1 * a0;
    double V[3][3];
    V[0][0] = usable ? a0 : 1.0; V[0][1] = usable ? b0 : 0.0; V[0][2] = usable ? v0 : 0.0;
    V[1][0] = usable ? a1 : 0.0; V[1][1] = usable ? b1 : 1.0; V[1][2] = usable ? v1 : 0.0;
    V[2][0] = usable ? a2 : 0.0; V[2][1] = usable ? b2 : 0.0; V[2][2] = usable ? v2 : 1.0;
    double B[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) B[r][c] = f[r][0] * V[0][c] + f[r][1] * V[1][c] + f[r][2] * V[2][c];
    // only v3 enters the projection f (I - v3 v3^T): the third column alone has to converge (lanes without a usable
    // estimate converge every pair and pick the smallest column below)
    SLAM_DBG_ADD(0, 1);
    jacobi_sweeps<3, 3, true>(B, V, 30, SLAM_RANK2_TOL, !usable);
    double n2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) n2[j] = B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j];
    // ||f v|| >= sigma_3 for every unit v, so a converged third column can never be longer than another column;
    // if it is, the start was poor and the third column settled on the wrong singular vector: converge every pair
    if (__any(n2[2] > n2[0] || n2[2] > n2[1])) {
        jacobi_sweeps<3, 3, false>(B, V, 30, 1e-15);
#pragma unroll
        for (int j = 0; j < 3; ++j) n2[j] = B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j];
    }
    // weight 0 for the column with the smallest norm (first one on ties), 1 for the others
    const bool d0 = (n2[0] <= n2[1]) && (n2[0] <= n2[2]);
    const bool d1 = !d0 && (n2[1] <= n2[2]);
    const double w[3] = {d0 ? 0.0 : 1.0, d1 ? 0.0 : 1.0, (d0 || d1) ? 1.0 : 0.0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s = fma(w[j] * B[r][j], V[c][j], s);
            F[3 * r + c] = s;
        }
}

// The same projection on the short path every well-separated matrix takes: with v the least right singular vector,
//   U diag(s0, s1, 0) V^T = f - (f v) v^T,
// so v alone is needed, to full accuracy.  Start as above (least root of the characteristic cubic, largest column of the
// adjugate), complete v to an orthonormal basis with a Householder reflector (w = v + sign(v_2) e_2, beta = 1 / (1 +
// |v_2|): no cancellation, no case distinction), then ONE joint first-order step on f itself makes f v orthogonal to
// the images of the two other basis vectors: with b_j = f h_j, b_2 = f v, a_j = |b_j|^2, c_ij = b_i . b_j,
//   t0 (a0 - a2) + t1 c01 = c02,   t0 c01 + t1 (a1 - a2) = c12,   v <- v - t0 h0 - t1 h1,   b_2 <- b_2 - t0 b_0 - t1 b_1.
// The step leaves an error of order t^2, so it is final when |t0|, |t1| <= 1e-8; a wave in which some lane's start was
// worse than that (least and middle singular values close together, or no usable start) also runs the sweeping
// version, whose result replaces the short path's on those lanes only.
__device__ __forceinline__ void rank2_project(const double (&f_in)[9], double (&F)[9]) {
    double nrm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nrm2 = fma(f_in[i], f_in[i], nrm2);
    const double inv = rsqrt(nrm2);
    double f[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[r][c] = f_in[3 * r + c] * inv;
    const double m00 = fma(f[0][0], f[0][0], fma(f[1][0], f[1][0], f[2][0] * f[2][0]));
    const double m01 = fma(f[0][0], f[0][1], fma(f[1][0], f[1][1], f[2][0] * f[2][1]));
    const double m02 = fma(f[0][0], f[0][2], fma(f[1][0], f[1][2], f[2][0] * f[2][2]));
    const double m11 = fma(f[0][1], f[0][1], fma(f[1][1], f[1][1], f[2][1] * f[2][1]));
    const double m12 = fma(f[0][1], f[0][2], fma(f[1][1], f[1][2], f[2][1] * f[2][2]));
    const double m22 = fma(f[0][2], f[0][2], fma(f[1][2], f[1][2], f[2][2] * f[2][2]));
    const double c2 = m00 + m11 + m22;
    const double k12 = fma(m11, m22, -m12 * m12), k02 = fma(m01, m22, -m12 * m02), k01 = fma(m01, m12, -m11 * m02);
    const double c1 = fma(m00, m11, -m01 * m01) + fma(m00, m22, -m02 * m02) + k12;
    const double c0 = fma(m00, k12, fma(-m01, k02, m02 * k01));
    double lam = 0.0;
#pragma unroll
    for (int it = 0; it < SLAM_CUBIC_NEWTON_STEPS; ++it) {
        const double p = fma(fma(c2 - lam, lam, -c1), lam, c0);
        const double dp = fma(fma(-3.0, lam, c2 + c2), lam, -c1);
        lam = fma(-p, __builtin_amdgcn_rcp(dp), lam);  // an approximate reciprocal is enough: the iteration self-corrects
    }
    const double a00 = m00 - lam, a11 = m11 - lam, a22 = m22 - lam;
    const double g00 = fma(a11, a22, -m12 * m12), g11 = fma(a00, a22, -m02 * m02), g22 = fma(a00, a11, -m01 * m01);
    const double g01 = fma(m02, m12, -m01 * a22), g02 = fma(m01, m12, -m02 * a11), g12 = fma(m01, m02, -a00 * m12);
    const bool k0 = (g00 >= g11) && (g00 >= g22);
    const bool k1 = !k0 && (g11 >= g22);
    double v0 = k0 ? g00 : (k1 ? g01 : g02);
    double v1 = k0 ? g01 : (k1 ? g11 : g12);
    double v2 = k0 ? g02 : (k1 ? g12 : g22);
    const double vn2 = fma(v0, v0, fma(v1, v1, v2 * v2));
    const bool usable = (vn2 > 0.0) && (vn2 < 1e300);  // false for NaN, zero or overflow
    const double vi = rsqrt(vn2);
    v0 *= vi; v1 *= vi; v2 *= vi;
    // h0, h1: the first two columns of I - beta w w^T
    const double w2 = v2 + (v2 >= 0.0 ? 1.0 : -1.0);
    const double beta = fast_rcp(1.0 + fabs(v2));
    const double bw0 = beta * v0, bw1 = beta * v1;
    const double h00 = fma(-bw0, v0, 1.0), h01 = -bw0 * v1, h02 = -bw0 * w2;  // h0 = (h00, h01, h02)
    const double h10 = h01, h11 = fma(-bw1, v1, 1.0), h12 = -bw1 * w2;        // h1 = (h10, h11, h12)
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        b0[r] = fma(f[r][0], h00, fma(f[r][1], h01, f[r][2] * h02));
        b1[r] = fma(f[r][0], h10, fma(f[r][1], h11, f[r][2] * h12));
        b2[r] = fma(f[r][0], v0, fma(f[r][1], v1, f[r][2] * v2));
    }
    const double q0 = fma(b0[0], b0[0], fma(b0[1], b0[1], b0[2] * b0[2]));
    const double q1 = fma(b1[0], b1[0], fma(b1[1], b1[1], b1[2] * b1[2]));
    const double q2 = fma(b2[0], b2[0], fma(b2[1], b2[1], b2[2] * b2[2]));
    const double c01 = fma(b0[0], b1[0], fma(b0[1], b1[1], b0[2] * b1[2]));
    const double c02 = fma(b0[0], b2[0], fma(b0[1], b2[1], b0[2] * b2[2]));
    const double c12 = fma(b1[0], b2[0], fma(b1[1], b2[1], b1[2] * b2[2]));
    const double d0 = q0 - q2, d1 = q1 - q2;
    const double det = fma(d0, d1, -c01 * c01);
    const double idet = fast_rcp(det);
    const double t0 = fma(c02, d1, -c12 * c01) * idet, t1 = fma(c12, d0, -c02 * c01) * idet;
    const bool fine = usable && (det > 0.0) && (fabs(t0) <= 1e-8) && (fabs(t1) <= 1e-8);
    const double u0 = fma(-t1, h10, fma(-t0, h00, v0));
    const double u1 = fma(-t1, h11, fma(-t0, h01, v1));
    const double u2 = fma(-t1, h12, fma(-t0, h02, v2));
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double br = fma(-t1, b1[r], fma(-t0, b0[r], b2[r]));
        F[3 * r + 0] = fma(-br, u0, f[r][0]);
        F[3 * r + 1] = fma(-br, u1, f[r][1]);
        F[3 * r + 2] = fma(-br, u2, f[r][2]);
    }
    // a lane's result never depends on which other lanes share its wave: fine lanes keep the short path's numbers
    if (__any(!fine)) {  // wave-uniform
        double Fs[9];
        rank2_project_sweeps(f_in, Fs);
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = fine ? F[i] : Fs[i];
    }
}

// Right singular vector of the smallest singular value of a 4 x 4 matrix A (held as A[row][col]).
//
// Warm start as in rank2_project: G = A^T A; lambda = smallest root of its characteristic quartic by Newton from 0
// (monotone from below); the largest column of adj(G - lambda I) is parallel to the wanted vector v; a Householder
// reflector that maps the last unit axis onto v supplies an orthonormal basis with v as its last column.  The polish
// works on A itself (it does not see G's squared condition number): Newton-like steps, then one-sided Jacobi sweeps
// for whatever they leave.  A is not modified.
// angle_tol: rotation angle below which a pair counts as converged.
// MAX_STEPS: Newton-like steps a lane may take before the sweeps (0: sweeps only, which keeps 30 registers free where
// the start is known to be good enough).
#ifndef SLAM_LEAST4_STEPS
#define SLAM_LEAST4_STEPS 3
#endif
// Z03: the caller guarantees A[0][3] == A[1][3] == 0 (a first camera K [I|0] has a zero last column, so the two rows
//   of the design matrix that come from it end in zero); the terms they would contribute are left out - the same bits
//   as with the zeros multiplied through, fewer instructions.
// (Ending the Newton iteration of the warm start as soon as the wave has converged was measured and dropped: the loop with
//   its vote costs more than the steps it saves, here and in the cubic of rank2_project.)
template <int MAX_STEPS = SLAM_LEAST4_STEPS, bool Z03 = false>
__device__ __forceinline__ void least_right_vector_4x4(double (&A)[4][4], double angle_tol, double (&y)[4]) {
    // (the warm start is written with explicit fma: it only has to be good to a few digits)
    double g[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            if (Z03 && j == 3) g[i][j] = fma(A[2][i], A[2][j], A[3][i] * A[3][j]);
            else g[i][j] = fma(A[0][i], A[0][j], fma(A[1][i], A[1][j], fma(A[2][i], A[2][j], A[3][i] * A[3][j])));
            g[j][i] = g[i][j];
        }
    // 2 x 2 minors of rows (0,1) and rows (2,3) of a symmetric matrix with diagonal d0..d3
    double s0, s1, s2, s3, s4, s5, c0, c1, c2, c3, c4, c5;
    auto minors = [&](double d0, double d1, double d2, double d3) {
        s0 = fma(d0, d1, -g[0][1] * g[0][1]);
        s1 = fma(d0, g[1][2], -g[0][1] * g[0][2]);
        s2 = fma(d0, g[1][3], -g[0][1] * g[0][3]);
        s3 = fma(g[0][1], g[1][2], -d1 * g[0][2]);
        s4 = fma(g[0][1], g[1][3], -d1 * g[0][3]);
        s5 = fma(g[0][2], g[1][3], -g[1][2] * g[0][3]);
        c0 = s5;
        c1 = fma(g[0][2], g[2][3], -d2 * g[0][3]);
        c2 = fma(g[0][2], d3, -g[2][3] * g[0][3]);
        c3 = fma(g[1][2], g[2][3], -d2 * g[1][3]);
        c4 = fma(g[1][2], d3, -g[2][3] * g[1][3]);
        c5 = fma(d2, d3, -g[2][3] * g[2][3]);
    };
    minors(g[0][0], g[1][1], g[2][2], g[3][3]);
    // characteristic polynomial x^4 - k3 x^3 + k2 x^2 - k1 x + k0
    const double k0 = fma(s0, c5, fma(-s1, c4, fma(s2, c3, fma(s3, c2, fma(-s4, c1, s5 * c0)))));
    const double k1 = fma(g[1][1], c5, fma(-g[1][2], c4, g[1][3] * c3)) + fma(g[0][0], c5, fma(-g[0][2], c2, g[0][3] * c1)) +
                      fma(g[3][3], s0, fma(-g[1][3], s2, g[0][3] * s4)) + fma(g[2][2], s0, fma(-g[1][2], s1, g[0][2] * s3));
    const double k2 = s0 + c5 + fma(g[0][0], g[2][2], -g[0][2] * g[0][2]) + fma(g[0][0], g[3][3], -g[0][3] * g[0][3]) +
                      fma(g[1][1], g[2][2], -g[1][2] * g[1][2]) + fma(g[1][1], g[3][3], -g[1][3] * g[1][3]);
    const double k3 = g[0][0] + g[1][1] + g[2][2] + g[3][3];
    double lam = 0.0;
#pragma unroll
    for (int it = 0; it < SLAM_QUARTIC_NEWTON_STEPS; ++it) {
        const double p = fma(fma(fma(lam - k3, lam, k2), lam, -k1), lam, k0);
        const double dp = fma(fma(fma(4.0, lam, -3.0 * k3), lam, k2 + k2), lam, -k1);
        lam = fma(-p, __builtin_amdgcn_rcp(dp), lam);  // approximate reciprocal: the iteration self-corrects
    }
    const double d0 = g[0][0] - lam, d1 = g[1][1] - lam, d2 = g[2][2] - lam, d3 = g[3][3] - lam;
    minors(d0, d1, d2, d3);
    // adjugate of the shifted matrix (symmetric): a multiple of v v^T
    const double j00 = fma(d1, c5, fma(-g[1][2], c4, g[1][3] * c3));
    const double j01 = fma(-g[0][1], c5, fma(g[0][2], c4, -g[0][3] * c3));
    const double j02 = fma(g[1][3], s5, fma(-g[2][3], s4, d3 * s3));
    const double j03 = fma(-g[1][2], s5, fma(d2, s4, -g[2][3] * s3));
    const double j11 = fma(d0, c5, fma(-g[0][2], c2, g[0][3] * c1));
    const double j12 = fma(-g[0][3], s5, fma(g[2][3], s2, -d3 * s1));
    const double j13 = fma(g[0][2], s5, fma(-d2, s2, g[2][3] * s1));
    const double j22 = fma(g[0][3], s4, fma(-g[1][3], s2, d3 * s0));
    const double j23 = fma(-g[0][2], s4, fma(g[1][2], s2, -g[2][3] * s0));
    const double j33 = fma(g[0][2], s3, fma(-g[1][2], s1, d2 * s0));
    const double a0 = fabs(j00), a1 = fabs(j11), a2 = fabs(j22), a3 = fabs(j33);
    const bool p0 = (a0 >= a1) && (a0 >= a2) && (a0 >= a3);
    const bool p1 = !p0 && (a1 >= a2) && (a1 >= a3);
    const bool p2 = !p0 && !p1 && (a2 >= a3);
    double v[4];
    v[0] = p0 ? j00 : p1 ? j01 : p2 ? j02 : j03;
    v[1] = p0 ? j01 : p1 ? j11 : p2 ? j12 : j13;
    v[2] = p0 ? j02 : p1 ? j12 : p2 ? j22 : j23;
    v[3] = p0 ? j03 : p1 ? j13 : p2 ? j23 : j33;
    const double vn2 = fma(v[0], v[0], fma(v[1], v[1], fma(v[2], v[2], v[3] * v[3])));
    const bool usable = (vn2 > 0.0) && (vn2 < 1e300);  // false for NaN, zero or overflow: start from the identity
    const double vi = rsqrt(vn2);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = usable ? v[i] * vi : (i == 3 ? 1.0 : 0.0);
    if constexpr (MAX_STEPS == 0) {
        // Householder reflector H = I - w w^T / (1 + |v_3|) with w = v + sign(v_3) e_3: its last column is -sign(v_3) v and
        // the other columns complete an orthonormal basis.  w^T w = 2 (1 + |v_3|) >= 2 whatever v_3 is: no cancellation.
        const double sg = v[3] >= 0.0 ? 1.0 : -1.0;
        const double w[4] = {v[0], v[1], v[2], v[3] + sg};
        const double beta = fast_rcp(1.0 + fabs(v[3]));
        // B = A H = A - beta (A w) w^T, in place
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (Z03 && r < 2) {
                const double aw = beta * fma(A[r][0], w[0], fma(A[r][1], w[1], A[r][2] * w[2]));
#pragma unroll
                for (int j = 0; j < 3; ++j) A[r][j] = fma(-aw, w[j], A[r][j]);
                A[r][3] = -aw * w[3];
            } else {
                const double aw = beta * fma(A[r][0], w[0], fma(A[r][1], w[1], fma(A[r][2], w[2], A[r][3] * w[3])));
#pragma unroll
                for (int j = 0; j < 4; ++j) A[r][j] = fma(-aw, w[j], A[r][j]);
            }
        }
        // the test of the sweeps on the three pairs with the last column, and the safety net's test (below): a wave
        // whose lanes all pass has the last column of H as its answer and forms neither V nor a rotation
        double n2[4], c[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) n2[j] = fma(A[0][j], A[0][j], fma(A[1][j], A[1][j], fma(A[2][j], A[2][j], A[3][j] * A[3][j])));
#pragma unroll
        for (int j = 0; j < 3; ++j) c[j] = fma(A[0][j], A[0][3], fma(A[1][j], A[1][3], fma(A[2][j], A[2][3], A[3][j] * A[3][3])));
        const double floor2 = ((n2[0] + n2[1]) + (n2[2] + n2[3])) * 1e-32;
        bool need = !usable || n2[3] > n2[0] || n2[3] > n2[1] || n2[3] > n2[2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            need = need || ((n2[j] > floor2) && (n2[3] > floor2) && (c[j] * c[j] > 1e-30 * (n2[j] * n2[3])) &&
                            (fabs(c[j]) > angle_tol * fabs(n2[j] - n2[3])));
        if (!__any(need)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fma(-beta * w[i], w[3], (i == 3) ? 1.0 : 0.0);
            return;
        }
        double V[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) V[i][j] = fma(-beta * w[i], w[j], (i == j) ? 1.0 : 0.0);
        jacobi_sweeps<4, 4, true>(A, V, 30, angle_tol, !usable);
#pragma unroll
        for (int j = 0; j < 4; ++j) n2[j] = fma(A[0][j], A[0][j], fma(A[1][j], A[1][j], fma(A[2][j], A[2][j], A[3][j] * A[3][j])));
        // a converged least column cannot be longer than any other column (||A v|| >= sigma_min for unit v); if it is,
        // the start was poor: converge every pair and take the shortest column
        if (__any(n2[3] > n2[0] || n2[3] > n2[1] || n2[3] > n2[2] || !usable)) {
            jacobi_sweeps<4, 4, false>(A, V, 30, angle_tol);
#pragma unroll
            for (int j = 0; j < 4; ++j) n2[j] = A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j] + A[3][j] * A[3][j];
        }
        int least = 3;
        double best = n2[3];
#pragma unroll
        for (int j = 2; j >= 0; --j) {
            const bool lt = n2[j] < best;
            best = lt ? n2[j] : best;
            least = lt ? j : least;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = least == 0 ? V[i][0] : (least == 1 ? V[i][1] : (least == 2 ? V[i][2] : V[i][3]));
        return;
    } else {
        // Polish on A itself.  A Householder reflector H = I - w w^T / (1 + |v_3|) with w = v + sign(v_3) e_3 has
        // -sign(v_3) v as its last column and completes it to an orthonormal basis (w^T w = 2 (1 + |v_3|) >= 2 whatever
        // v_3 is: no cancellation).  With b_j the columns of B = A H, c_j = b_j . b_3 and a_j = |b_j|^2, the last column is
        // done when every pair (j, 3) passes the test of the Jacobi sweeps (|c_j| <= angle_tol |a_j - a_3|, or orthogonal
        // to rounding).  A lane that is not done takes a Newton-like step: z = (t, 1) with (M0 - a_3 I) t = -m, where
        // M0 = [b_i . b_j] (i, j < 3) and m = (c_0, c_1, c_2), is the least right singular vector of B up to
        // |(M0 - a_3 I)^-1| |m| |t|^2: the error is cubed.  v becomes H z, normalised, and the basis is rebuilt from it (B
        // comes from the untouched A every time, so nothing accumulates).  A sweep of rotations only squares the error
        // and costs three times the instructions.  Lanes that are done keep their v bit for bit, so a lane's result does
        // not depend on its wave mates.  What is not done after MAX_STEPS steps - singular values too close for
        // the step to make sense, or no usable start - goes through the sweeps, starting from the basis reached.
        double B[4][4], w[4], beta, n2[4];
        bool done = false, stuck = false;
#pragma unroll 1
        for (int step = 0;; ++step) {
            const double sg = v[3] >= 0.0 ? 1.0 : -1.0;
            w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3] + sg;
            beta = fast_rcp(1.0 + fabs(v[3]));
            // B = A H = A - beta (A w) w^T
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double aw = beta * fma(A[r][0], w[0], fma(A[r][1], w[1], fma(A[r][2], w[2], A[r][3] * w[3])));
#pragma unroll
                for (int j = 0; j < 4; ++j) B[r][j] = fma(-aw, w[j], A[r][j]);
            }
            double c[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) n2[j] = fma(B[0][j], B[0][j], fma(B[1][j], B[1][j], fma(B[2][j], B[2][j], B[3][j] * B[3][j])));
#pragma unroll
            for (int j = 0; j < 3; ++j) c[j] = fma(B[0][j], B[0][3], fma(B[1][j], B[1][3], fma(B[2][j], B[2][3], B[3][j] * B[3][3])));
            const double floor2 = ((n2[0] + n2[1]) + (n2[2] + n2[3])) * 1e-32;
            bool need = false;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                need = need || ((n2[j] > floor2) && (n2[3] > floor2) && (c[j] * c[j] > 1e-30 * (n2[j] * n2[3])) &&
                                (fabs(c[j]) > angle_tol * fabs(n2[j] - n2[3])));
            done = usable && !need;
            bool step_now = usable && need && !stuck && step < MAX_STEPS;
            if (!__any(step_now)) break;
            const double m01 = fma(B[0][0], B[0][1], fma(B[1][0], B[1][1], fma(B[2][0], B[2][1], B[3][0] * B[3][1])));
            const double m02 = fma(B[0][0], B[0][2], fma(B[1][0], B[1][2], fma(B[2][0], B[2][2], B[3][0] * B[3][2])));
            const double m12 = fma(B[0][1], B[0][2], fma(B[1][1], B[1][2], fma(B[2][1], B[2][2], B[3][1] * B[3][2])));
            const double s00 = n2[0] - n2[3], s11 = n2[1] - n2[3], s22 = n2[2] - n2[3];
            const double q00 = fma(s11, s22, -m12 * m12), q01 = fma(m02, m12, -m01 * s22), q02 = fma(m01, m12, -m02 * s11);
            const double q11 = fma(s00, s22, -m02 * m02), q12 = fma(m01, m02, -s00 * m12), q22 = fma(s00, s11, -m01 * m01);
            const double det = fma(s00, q00, fma(m01, q01, m02 * q02));
            const double ninv = -fast_rcp(det);
            const double t0 = ninv * fma(q00, c[0], fma(q01, c[1], q02 * c[2]));
            const double t1 = ninv * fma(q01, c[0], fma(q11, c[1], q12 * c[2]));
            const double t2 = ninv * fma(q02, c[0], fma(q12, c[1], q22 * c[2]));
            const double tt = fma(t0, t0, fma(t1, t1, t2 * t2));
            stuck = stuck || !(tt <= 0.25);  // (true for NaN too) such a lane waits for the sweeps
            step_now = step_now && !stuck;
            // H z with z = (t, 1): z - beta (w . z) w, then to unit length
            const double wz = beta * fma(w[0], t0, fma(w[1], t1, fma(w[2], t2, w[3])));
            const double y0 = fma(-wz, w[0], t0), y1 = fma(-wz, w[1], t1), y2 = fma(-wz, w[2], t2), y3 = fma(-wz, w[3], 1.0);
            const double yi = rsqrt(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3))));
            v[0] = step_now ? y0 * yi : v[0];
            v[1] = step_now ? y1 * yi : v[1];
            v[2] = step_now ? y2 * yi : v[2];
            v[3] = step_now ? y3 * yi : v[3];
        }
        // a converged least column cannot be longer than any other column (||A v|| >= sigma_min for unit v); if it is, the
        // start was poor: converge every pair and take the shortest column
        const bool poor = n2[3] > n2[0] || n2[3] > n2[1] || n2[3] > n2[2];
        if (__any(!done || poor)) {  // wave-uniform; the rotations leave a lane that is done as it is, bit for bit
            double V[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) V[i][j] = fma(-beta * w[i], w[j], (i == j) ? 1.0 : 0.0);
            if (__any(!done)) {
                jacobi_sweeps<4, 4, true>(B, V, 30, angle_tol, !usable);
#pragma unroll
                for (int j = 0; j < 4; ++j) n2[j] = fma(B[0][j], B[0][j], fma(B[1][j], B[1][j], fma(B[2][j], B[2][j], B[3][j] * B[3][j])));
            }
            if (__any(n2[3] > n2[0] || n2[3] > n2[1] || n2[3] > n2[2] || !usable)) {
                jacobi_sweeps<4, 4, false>(B, V, 30, angle_tol);
#pragma unroll
                for (int j = 0; j < 4; ++j) n2[j] = B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j] + B[3][j] * B[3][j];
            }
            int least = 3;
            double best = n2[3];
#pragma unroll
            for (int j = 2; j >= 0; --j) {
                const bool lt = n2[j] < best;
                best = lt ? n2[j] : best;
                least = lt ? j : least;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = least == 0 ? V[i][0] : (least == 1 ? V[i][1] : (least == 2 ? V[i][2] : V[i][3]));
            return;
        }
        // every lane of the wave is done: the last column of H (the expression the branch above has for it)
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = fma(-beta * w[i], w[3], (i == 3) ? 1.0 : 0.0);
    }
}

// Sum over the 64 lanes of a wave in a fixed order, without LDS traffic: xor-1 and xor-2 inside quads, then the
// mirrored halves of 8 and of 16 lanes (DPP), then the four rows of 16 as (r0 + r1) + (r2 + r3).
// Every lane must be active.  Returns the (wave-uniform) total.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// Wave totals of EIGHT per-lane values at once (v[j], j = 0..7), in a fixed order and with a third of the
// instructions of eight separate wave sums.  Each exchange step halves the number of values a lane carries: a lane
// keeps one half of its values, sends the other half to its partner and adds what the partner sent.
//   step 1: partner lane^1 (quad_perm), keep v[0..3] if c1 = b0^b2 is 0, else v[4..7]
//   step 2: partner lane^2 (quad_perm), keep the lower pair if c2 = b1^b2 is 0
//   step 3: partner 7-lane within 8 (row_half_mirror), keep the lower one if c3 = b2^b3 is 0
// (b0..b3 = lane bits; the predicates are chosen so that a lane's partner always carries the same subset).
// After step 3 a lane holds the sum over its 8 lanes of v[4 c1 + 2 c2 + c3]; then the usual all-reduce steps over
// 16 lanes (row_mirror), 32 lanes (xor 16) and 64 lanes (xor 32).  All 64 lanes must be active.
// Returns the total of v[slot] where slot = wave_sum8_slot(lane); lanes 0..7 cover slots {0,4,2,6,7,3,5,1}.
__device__ __forceinline__ int wave_sum8_slot(int lane) {
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    return 4 * (b0 ^ b2) + 2 * (b1 ^ b2) + (b2 ^ b3);
}

__device__ __forceinline__ double wave_sum8_f64(const double (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    const bool c1 = (b0 ^ b2) != 0, c2 = (b1 ^ b2) != 0, c3 = (b2 ^ b3) != 0;
    double t4[4], t2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = c1 ? v[4 + i] : v[i];
        const double send = c1 ? v[i] : v[4 + i];
        t4[i] = keep + dpp_f64<0xB1>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = c2 ? t4[2 + i] : t4[i];
        const double send = c2 ? t4[i] : t4[2 + i];
        t2[i] = keep + dpp_f64<0x4E>(send);
    }
    const double keep = c3 ? t2[1] : t2[0];
    const double send = c3 ? t2[0] : t2[1];
    double t = keep + dpp_f64<0x141>(send);
    t += dpp_f64<0x140>(t);                       // rows of 16
    {                                             // lane ^ 16 (ds_swizzle, bit mode: and 0x1F, xor 0x10)
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(t), 0x401F);
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(t), 0x401F);
        t += __hiloint2double(hi, lo);
    }
    t += __shfl_xor(t, 32, 64);                   // lane ^ 32
    return t;
}

// The same for FOUR per-lane values: two halving steps (lane^1, lane^2), then the all-reduce steps of 8, 16, 32 and 64
// lanes.  Returns the total of v[wave_sum4_slot(lane)]; lanes 0..3 cover slots {0,2,1,3}.  The summation tree of every
// value is the one of wave_sum_f64 (additions commute, so which partner adds does not matter): the results are
// bit-identical to four separate wave_sum_f64 calls.
__device__ __forceinline__ int wave_sum4_slot(int lane) {
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1;
    return 2 * (b0 ^ b2) + (b1 ^ b2);
}

__device__ __forceinline__ double wave_sum4_f64(const double (&v)[4]) {
    const int lane = threadIdx.x & 63;
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1;
    const bool c1 = (b0 ^ b2) != 0, c2 = (b1 ^ b2) != 0;
    double t2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = c1 ? v[2 + i] : v[i];
        const double send = c1 ? v[i] : v[2 + i];
        t2[i] = keep + dpp_f64<0xB1>(send);
    }
    const double keep = c2 ? t2[1] : t2[0];
    const double send = c2 ? t2[0] : t2[1];
    double t = keep + dpp_f64<0x4E>(send);
    t += dpp_f64<0x141>(t);                       // 7 - l within 8
    t += dpp_f64<0x140>(t);                       // 15 - l within 16
    {                                             // lane ^ 16 (ds_swizzle, bit mode: and 0x1F, xor 0x10)
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(t), 0x401F);
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(t), 0x401F);
        t += __hiloint2double(hi, lo);
    }
    t += __shfl_xor(t, 32, 64);                   // lane ^ 32
    return t;
}

}  // namespace slamk
