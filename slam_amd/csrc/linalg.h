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

// One-sided (Hestenes) Jacobi SVD of an M x N matrix held as A[row][col].
// On return A = A_in * V (column j = sigma_j u_j) and V holds the right singular vectors as columns.
// Columns whose squared norm is below floor2 are treated as already converged (rounding-level null columns).
template <int M, int N>
__device__ __forceinline__ void jacobi_svd(double (&A)[M][N], double (&V)[N][N], int max_sweeps) {
    double fro2 = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) fro2 = fma(A[i][j], A[i][j], fro2);
    const double floor2 = fro2 * 1e-32;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;

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
        // leave when no lane of the wave still rotates (keeps the wave converged together)
        if (!__any(rotated)) break;
    }
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
        const double normx = sqrt(fma(x0, x0, sigma));
        const double alpha = (x0 >= 0.0) ? -normx : normx;
        const double v0 = x0 - alpha;                    // same sign as x0: no cancellation
        const double vtv = fma(v0, v0, sigma);
        const double bj = (vtv > 0.0) ? 2.0 / vtv : 0.0; // zero column: H = I
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
__device__ __forceinline__ void rank2_project(const double (&f_in)[9], double (&F)[9]) {
    double nrm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) nrm2 = fma(f_in[i], f_in[i], nrm2);
    const double inv = 1.0 / sqrt(nrm2);
    double B[3][3], V[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) B[r][c] = f_in[3 * r + c] * inv;
    jacobi_svd<3, 3>(B, V, 30);
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

}  // namespace slamk
