// Register-resident small dense linear algebra for one lane (float64), gfx950.
// Everything is fully unrolled with compile-time indices so that the arrays live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace slamk {

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
// On return A = A_in * R and V = V_in * R for the accumulated rotation R; the columns of A are mutually orthogonal
// (column j = sigma_j u_j when V_in was the identity).
// Columns whose squared norm is below 1e-32 ||A||_F^2 are treated as converged (rounding-level null columns).
template <int M, int N>
__device__ __forceinline__ void jacobi_sweeps(double (&A)[M][N], double (&V)[N][N], int max_sweeps) {
    double fro2 = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) fro2 = fma(A[i][j], A[i][j], fro2);
    const double floor2 = fro2 * 1e-32;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    a = fma(A[i][p], A[i][p], a);
                    b = fma(A[i][q], A[i][q], b);
                    c = fma(A[i][p], A[i][q], c);
                }
                // rotate while the columns are not orthogonal to ~1e-15 relative
                const bool go = (a > floor2) && (b > floor2) && (c * c > 1e-30 * (a * b));
                if (__any(go)) {  // wave-uniform: a converged wave only pays for the three inner products
                    double cs, sn;
                    jacobi_angle(a, b, c, cs, sn);
                    cs = go ? cs : 1.0;
                    sn = go ? sn : 0.0;
                    rotated |= go;
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
            }
        }
        // leave when no lane of the wave still rotates (keeps the wave converged together)
        if (!__any(rotated)) break;
    }
}

template <int M, int N>
__device__ __forceinline__ void jacobi_svd(double (&A)[M][N], double (&V)[N][N], int max_sweeps) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    jacobi_sweeps<M, N>(A, V, max_sweeps);
}

// Unit null vector of the 8 x 9 system  sum_r n[r] * A[r][c] = 0  for c = 0..7, where A (9 x 8) holds the
// design matrix transposed (row = monomial, column = correspondence).  Householder QR of A; the null vector is
// the last column of Q = H0 H1 ... H7.  Orthogonal transforms only, so the result is backward stable like an SVD's.
__device__ __forceinline__ void null_vector_9x8(double (&A)[9][8], double (&n)[9]) {
    double beta[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double sigma = 0.0;
#pragma unroll
        for (int r = j + 1; r < 9; ++r) sigma = fma(A[r][j], A[r][j], sigma);
        const double x0 = A[j][j];
        const double s2 = fma(x0, x0, sigma);
        const double normx = (s2 > 0.0) ? s2 * rsqrt(s2) : 0.0;
        const double alpha = (x0 >= 0.0) ? -normx : normx;
        const double v0 = x0 - alpha;                    // same sign as x0: no cancellation
        const double vtv = fma(v0, v0, sigma);
        // beta only has to match v to rounding for H = I - beta v v^T to be orthogonal; zero column: H = I
        const double bj = (vtv > 0.0) ? 2.0 * fast_rcp(vtv) : 0.0;
        beta[j] = bj;
        A[j][j] = v0;
#pragma unroll
        for (int c = j + 1; c < 8; ++c) {
            double s = v0 * A[j][c];
#pragma unroll
            for (int r = j + 1; r < 9; ++r) s = fma(A[r][j], A[r][c], s);
            s *= bj;
            A[j][c] = fma(-s, v0, A[j][c]);
#pragma unroll
            for (int r = j + 1; r < 9; ++r) A[r][c] = fma(-s, A[r][j], A[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) n[r] = (r == 8) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        double s = A[j][j] * n[j];
#pragma unroll
        for (int r = j + 1; r < 9; ++r) s = fma(A[r][j], n[r], s);
        s *= beta[j];
        n[j] = fma(-s, A[j][j], n[j]);
#pragma unroll
        for (int r = j + 1; r < 9; ++r) n[r] = fma(-s, A[r][j], n[r]);
    }
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
    for (int it = 0; it < 6; ++it) {
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
    jacobi_sweeps<3, 3>(B, V, 30);
    double n2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) n2[j] = B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j];
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

}  // namespace slamk
