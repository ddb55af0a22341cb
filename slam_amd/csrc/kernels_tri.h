// Triangulation, pose recovery and the small per-pair matrix steps, gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "linalg.h"

#ifndef SLAM_TRI_ANGLE_TOL
#define SLAM_TRI_ANGLE_TOL 1e-12
#endif
namespace slamk {

// Layout of the per-context "pose work" block (float64 words) shared by the kernels below.
enum {
    PW_P0 = 0,            // 12: K [I|0]
    PW_CAND = 12,         // 4 candidates x PW_CAND_LEN
    PW_CAND_P1 = 0,       //   12: K [R^T | -R^T t]
    PW_CAND_T = 12,       //   3 : t
    PW_CAND_CZ = 15,      //   3 : R (0,0,1)^T = third column of R
    PW_CAND_POSE = 18,    //   16: [[R^T, -R^T t],[0,0,0,1]]
    PW_CAND_LEN = 34,
    PW_E = 12 + 4 * 34,   // 9
    PW_F = PW_E + 9,      // 9 (denormalised F of the pair pipeline)
    PW_T0 = PW_F + 9,     // 9
    PW_T1 = PW_T0 + 9,    // 9
    PW_K = PW_T1 + 9,     // 9
    PW_LEN = PW_K + 9 + 3
};

// ------------------------------------------------------------------------------------------------
// One DLT triangulation   (reference: src/triangulation.py:50-73)
//   D rows: -P0[1] + v P0[2];  P0[0] - u P0[2];  -P1[1] + v' P1[2];  P1[0] - u' P1[2]
//   m_i = max_j |D_ij| ; D' = D diag(1/m) (column j divided by m_j) ; y = right singular vector of the smallest
//   singular value of D' ; Y = diag(m) y ; X = Y[0..2] / Y[3]
// P0, P1 are wave-uniform pointers (scalar loads).
// ------------------------------------------------------------------------------------------------
// 1/x to within an ulp, with 1/(+-0) = +-inf like the division it stands for
__device__ __forceinline__ double rcp_like_division(double x) {
    const double r = fast_rcp(x);
    return x == 0.0 ? __builtin_amdgcn_rcp(x) : r;
}

// angle_tol: rotation angle below which the polish of the least singular vector stops.  1e-12 serves cameras of the
// form K [R|t] (pose recovery, hypothesis scoring: the points then agree with LAPACK's to 1e-11); callers whose
// cameras are arbitrary - the reference's tracking step triangulates with the POSES themselves against pixel
// coordinates (src/triangulation.py:179), which leaves the two smallest singular values close together and the
// homogeneous coordinate small - pass SLAM_TRI_ANGLE_TOL_TIGHT.
#define SLAM_TRI_ANGLE_TOL_TIGHT 1e-15
// P0_IS_K_I0: the first camera is K [I|0] (pose recovery): its last column is zero, so rows 0 and 1 of D end in zero.
//   The zeros are not computed, not scaled and not multiplied through (least_right_vector_4x4: Z03).
template <int MAX_STEPS = SLAM_LEAST4_STEPS, bool P0_IS_K_I0 = false>
__device__ __forceinline__ void triangulate_one(const double* __restrict__ P0, const double* __restrict__ P1, double u0,
                                                double v0, double u1, double v1, double (&X)[3],
                                                double angle_tol = SLAM_TRI_ANGLE_TOL) {
    double D[4][4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (P0_IS_K_I0 && j == 3) {
            D[0][j] = 0.0;
            D[1][j] = 0.0;
        } else {
            D[0][j] = fma(v0, P0[8 + j], -P0[4 + j]);
            D[1][j] = fma(-u0, P0[8 + j], P0[j]);
        }
        D[2][j] = fma(v1, P1[8 + j], -P1[4 + j]);
        D[3][j] = fma(-u1, P1[8 + j], P1[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (P0_IS_K_I0 && i < 2) m[i] = fmax(fmax(fabs(D[i][0]), fabs(D[i][1])), fabs(D[i][2]));
        else m[i] = fmax(fmax(fabs(D[i][0]), fabs(D[i][1])), fmax(fabs(D[i][2]), fabs(D[i][3])));
    }
    double im[4];  // reciprocals to within an ulp (the points are compared to 1e-5)
#pragma unroll
    // (a camera K [I|0] or K [R|t] has no zero row, so no row maximum is zero: the plain reciprocal serves there)
    for (int j = 0; j < 4; ++j) im[j] = P0_IS_K_I0 ? fast_rcp(m[j]) : rcp_like_division(m[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(P0_IS_K_I0 && i < 2 && j == 3)) D[i][j] *= im[j];
    // polished until further rotations would turn y by less than 1e-12 rad (the points are compared to 1e-5)
    double yv[4];
    least_right_vector_4x4<MAX_STEPS, P0_IS_K_I0>(D, angle_tol, yv);
    double Y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Y[i] = m[i] * yv[i];
    const double iy = rcp_like_division(Y[3]);
    X[0] = Y[0] * iy;
    X[1] = Y[1] * iy;
    X[2] = Y[2] * iy;
}

struct Cams {  // kernel argument: the first three rows of both camera matrices
    double P0[12];
    double P1[12];
};
struct Mat3x2 {  // kernel argument: two 3x3 matrices
    double A[9];
    double B[9];
};

// One pair of a batched call (slam_init_pairs): the arguments the pair pipeline's kernels would get for this pair.
// A kernel launched with a non-NULL `batch` takes its arguments from batch[blockIdx.y] instead of its parameters.
struct PairArgs {
    const double2 *r0, *r1;       // matches in pixel coordinates
    double2 *n0, *n1;             // their normalised copies
    int N, H;
    double* pw;                   // pose work block (PW_*)
    double* F_all;                // H x 9   }
    int32_t* counts;              // H       } hypothesis scratch
    double* errsums;              // H       }
    double* record;               // SLAM_REC_LEN
    unsigned long long* counts4;  // 4
    double* X4;                   // 2 x N x 3
    double* X;                    // N x 3
    uint8_t* good;                // N
    double* pose;                 // 16
    int32_t* winner;
};

// K4a: plain triangulation with one camera pair (slam_triangulate)
__global__ __launch_bounds__(256) void k_triangulate(Cams cams, const double2* __restrict__ p0,
                                                     const double2* __restrict__ p1, int64_t N,
                                                     double* __restrict__ Xout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double2 a = p0[i], b = p1[i];
    double X[3];
    triangulate_one(cams.P0, cams.P1, a.x, a.y, b.x, b.y, X, SLAM_TRI_ANGLE_TOL_TIGHT);
    Xout[3 * i + 0] = X[0];
    Xout[3 * i + 1] = X[1];
    Xout[3 * i + 2] = X[2];
}

// K4b: the four (R,t) candidates of one E   (reference: src/epipolar.py:113-127)
//   in front of camera 1: X_z > 0 ; in front of camera 2: (X - t) . (R e_z) > 0 ; count points passing both.
// The points go to two planes X4[r][N][3], r = 0, 1: plane r holds candidate 2 r, and candidate 2 r + 1 is its negative.
// counts4 (int64) must be zero on entry.
// A workgroup is ONE wave and takes per_wg consecutive matches (a multiple of 64), 64 at a time.  The Jacobi polish makes waves finish a
// match at different times; as four-wave workgroups the early waves' slots stayed empty until the last wave had
// ended (2.6 of 4 waves per SIMD on average).  The wave adds its four counts with one atomic instruction: all waves'
// atomics go to the same four words and complete one after the other (about 13 ns each), so per_wg grows with the
// number of matches - 64 for a single pair (every wave one batch: the shortest kernel), 1024 for millions of matches
// (at 256 the 65 536 atomics of 16 M matches were as long as the kernel's arithmetic).
__host__ __device__ inline int tri4_per_wg(int64_t matches_in_launch) {
    return matches_in_launch <= (1 << 16) ? 64 : (matches_in_launch <= (1 << 20) ? 256 : 1024);
}
// Beyond a million matches the waves' counts go to one of SLAM_TRI4_COUNT_SETS sets of four counters (the set is the
// workgroup number modulo that), so that the atomics of different waves rarely meet on an address and a workgroup can
// be short again; k_fold_counts4 adds the sets up.  (The launcher zeroes the sets.)
#define SLAM_TRI4_COUNT_SETS 256
#ifndef SLAM_TRI4_STEPS
#define SLAM_TRI4_STEPS 0
#endif
#ifndef SLAM_TRI4_SPECIAL   // the first camera of pose recovery is K [I|0] (pose_prepare): use what that gives away
#define SLAM_TRI4_SPECIAL true
#endif
#ifndef SLAM_TRI4_WAVES_PER_SIMD
#define SLAM_TRI4_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(64, SLAM_TRI4_WAVES_PER_SIMD) void k_triangulate4(const double* __restrict__ pw, const double2* __restrict__ p0,
                                                     const double2* __restrict__ p1, int64_t N, double* __restrict__ X4,
                                                     unsigned long long* __restrict__ counts4,
                                                     const PairArgs* __restrict__ batch, int per_wg,
                                                     unsigned long long* __restrict__ count_sets) {
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        pw = a.pw; p0 = a.r0; p1 = a.r1; N = a.N; X4 = a.X4; counts4 = a.counts4;
        if ((int64_t)blockIdx.x * per_wg >= N) return;  // the grid is sized for the largest pair
    }
    const int lane = threadIdx.x;
    // The cameras and the cheirality vectors of the four candidates are staged in LDS once: read from there they are
    // broadcast reads that wait for LDS traffic only.  Read from global memory inside the candidate loop they were vector
    // loads whose wait also covered the previous candidate's point stores (44 % of the wave cycles were such waits).
    __shared__ double s_pw[PW_E];
    for (int q = lane; q < PW_E; q += 64) s_pw[q] = pw[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int cnt[4] = {0, 0, 0, 0};  // wave-uniform
    // (all addresses are a wave-uniform 64-bit base plus a 32-bit lane offset: per-lane 64-bit indices and pointers kept
    // across the solve were what the register allocator spilled)
    const int64_t wg0 = (int64_t)blockIdx.x * per_wg;
#pragma unroll 1
    for (int k = 0; k < per_wg / 64; ++k) {
        const int64_t base = wg0 + k * 64;
        if (base >= N) break;  // uniform
        const int nlive = (int)(N - base < 64 ? N - base : 64);
        const bool live = lane < nlive;
        const int ll = live ? lane : nlive - 1;
        // Candidates 2 r and 2 r + 1 share the rotation and have opposite translations (pose_prepare): the last column
        // of the second camera changes sign and nothing else does, the first camera's last column is zero, so the whole
        // last column of D changes sign, the least singular vector changes the sign of its last component, and the
        // point of candidate 2 r + 1 is exactly the negative of candidate 2 r's (IEEE arithmetic is symmetric under a
        // change of sign).  Two triangulations serve the four candidates; (X' - t') . c = -((X - t) . c) likewise.
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            asm volatile("" ::: "memory");  // the LDS reads stay inside the loop: hoisted, they cost 40 registers
            // (the match is read again for the second rotation - a cache hit - instead of occupying eight registers
            // through the first solve)
            const double2 a = (p0 + base)[ll], b = (p1 + base)[ll];
            const double* cand = s_pw + PW_CAND + 2 * r * PW_CAND_LEN;
            double X[3];
            // (sweeps only: with cameras K [R|t] the warm start is within the tolerance for all but one match in
            // thousands, and the Newton-like steps would cost the fourth wave per SIMD their registers)
            triangulate_one<SLAM_TRI4_STEPS, SLAM_TRI4_SPECIAL>(s_pw + PW_P0, cand + PW_CAND_P1, a.x, a.y, b.x, b.y, X);
            const double tx = cand[PW_CAND_T], ty = cand[PW_CAND_T + 1], tz = cand[PW_CAND_T + 2];
            const double cx = cand[PW_CAND_CZ], cy = cand[PW_CAND_CZ + 1], cz = cand[PW_CAND_CZ + 2];
            const double s2 = (X[0] - tx) * cx + (X[1] - ty) * cy + (X[2] - tz) * cz;
            const int n_pos = __popcll(__ballot(live && X[2] > 0.0 && s2 > 0.0));
            const int n_neg = __popcll(__ballot(live && -X[2] > 0.0 && -s2 > 0.0));
            cnt[0] += r == 0 ? n_pos : 0; cnt[1] += r == 0 ? n_neg : 0;
            cnt[2] += r == 1 ? n_pos : 0; cnt[3] += r == 1 ? n_neg : 0;
            if (live) {  // plane r holds candidate 2 r; candidate 2 r + 1 is its negative (candidate_point below)
                double* o = X4 + ((size_t)r * (size_t)N + (size_t)base) * 3;  // uniform
                int l3;  // 3 * lane, formed here every time (volatile: kept across the solve it would be spilled)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %0, 1, %0" : "=v"(l3));
                o[l3 + 0] = X[0];
                o[l3 + 1] = X[1];
                o[l3 + 2] = X[2];
            }
        }
    }
    if (lane < 4) {
        const int n = lane == 0 ? cnt[0] : (lane == 1 ? cnt[1] : (lane == 2 ? cnt[2] : cnt[3]));
        unsigned long long* dst = count_sets ? count_sets + 4 * (blockIdx.x % SLAM_TRI4_COUNT_SETS) : counts4;
        if (n) atomicAdd(&dst[lane], (unsigned long long)n);
    }
}

// counts4[c] += sum over the sets of count_sets[set][c]   (one workgroup of SLAM_TRI4_COUNT_SETS * 4 threads)
__global__ __launch_bounds__(SLAM_TRI4_COUNT_SETS * 4) void k_fold_counts4(const unsigned long long* __restrict__ count_sets,
                                                                         unsigned long long* __restrict__ counts4) {
    __shared__ unsigned long long s_c[SLAM_TRI4_COUNT_SETS * 4];
    const int t = threadIdx.x;
    s_c[t] = count_sets[t];
    __syncthreads();
    if (t < 4) {
        unsigned long long n = 0;
        for (int k = 0; k < SLAM_TRI4_COUNT_SETS; ++k) n += s_c[4 * k + t];
        counts4[t] += n;
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate enumeration   (reference: src/epipolar.py:97-107,117-118; src/util.py:30-37)
//   u,s,vt = svd(E); R in {u W vt, u W^T vt} * sign(det(u W vt)); t in {u[:,2], -u[:,2]}
//   pose(R,t) = [[R^T, -R^T t],[0,0,0,1]]; P1 = K pose[:3]; P0 = K [I|0]
// The order of the two R (and of the two t) depends on the sign choices of the SVD routine; this one
// orders by its own Jacobi SVD with singular values sorted descending and u3 = u1 x u2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(bool doit, double& x, double& y) {
    const double t = x;
    x = doit ? y : x;
    y = doit ? t : y;
}

__device__ inline void pose_prepare(const double* __restrict__ E, const double* __restrict__ K, double* __restrict__ pw) {
    double B[3][3], V[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) B[r][c] = E[3 * r + c];
    jacobi_svd<3, 3>(B, V, 30);
    double n2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) n2[j] = B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j];
    // sort columns by descending norm (3-element network: (0,1) (1,2) (0,1))
#define SLAM_CSWAP_COL(p, q)                                                  \
    {                                                                         \
        const bool sw = n2[q] > n2[p];                                        \
        cswap(sw, n2[p], n2[q]);                                              \
        for (int r_ = 0; r_ < 3; ++r_) { cswap(sw, B[r_][p], B[r_][q]); cswap(sw, V[r_][p], V[r_][q]); } \
    }
    SLAM_CSWAP_COL(0, 1)
    SLAM_CSWAP_COL(1, 2)
    SLAM_CSWAP_COL(0, 1)
#undef SLAM_CSWAP_COL
    double U[3][3];
    const double i0 = 1.0 / sqrt(n2[0]), i1 = 1.0 / sqrt(n2[1]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { U[r][0] = B[r][0] * i0; U[r][1] = B[r][1] * i1; }
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    // U W = [u2, -u1, u3] ; U W^T = [-u2, u1, u3]   (W = [[0,-1,0],[1,0,0],[0,0,1]])
    double R[2][3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double x = U[r][1] * V[c][0] - U[r][0] * V[c][1];
            const double z = U[r][2] * V[c][2];
            R[0][r][c] = x + z;
            R[1][r][c] = -x + z;
        }
    const double det = R[0][0][0] * (R[0][1][1] * R[0][2][2] - R[0][1][2] * R[0][2][1]) -
                       R[0][0][1] * (R[0][1][0] * R[0][2][2] - R[0][1][2] * R[0][2][0]) +
                       R[0][0][2] * (R[0][1][0] * R[0][2][1] - R[0][1][1] * R[0][2][0]);
    const double mult = det > 0.0 ? 1.0 : -1.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pw[PW_P0 + 4 * r + c] = K[3 * r + c]; }
    pw[PW_P0 + 3] = 0.0; pw[PW_P0 + 7] = 0.0; pw[PW_P0 + 11] = 0.0;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int ri = ci >> 1;
        const double ts = (ci & 1) ? -1.0 : 1.0;
        double* __restrict__ cand = pw + PW_CAND + ci * PW_CAND_LEN;
        double t[3], pose[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r] = ts * U[r][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pose[r][c] = mult * R[ri][c][r];  // R^T
                s += pose[r][c] * t[c];
            }
            pose[r][3] = -s;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                cand[PW_CAND_P1 + 4 * r + c] = K[3 * r] * pose[0][c] + K[3 * r + 1] * pose[1][c] + K[3 * r + 2] * pose[2][c];
                cand[PW_CAND_POSE + 4 * r + c] = pose[r][c];
            }
        cand[PW_CAND_POSE + 12] = 0.0; cand[PW_CAND_POSE + 13] = 0.0; cand[PW_CAND_POSE + 14] = 0.0;
        cand[PW_CAND_POSE + 15] = 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { cand[PW_CAND_T + r] = t[r]; cand[PW_CAND_CZ + r] = mult * R[ri][r][2]; }
    }
}

// F -> E   (reference: src/epipolar.py:87-92):  E = K^T F K / ||K^T F K||_F
__device__ inline void f_to_e(const double* __restrict__ F, const double* __restrict__ K, double* __restrict__ E) {
    double T[3][3], G[3][3], n2 = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r][c] = K[r] * F[c] + K[3 + r] * F[3 + c] + K[6 + r] * F[6 + c];  // K^T F
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            G[r][c] = T[r][0] * K[c] + T[r][1] * K[3 + c] + T[r][2] * K[6 + c];
            n2 += G[r][c] * G[r][c];
        }
    const double nrm = sqrt(n2);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) E[3 * r + c] = G[r][c] / nrm;
}

// zero the candidate counters and build the candidates from E (ek.A) and K (ek.B)
__global__ __launch_bounds__(64) void k_pose_prepare(Mat3x2 ek, double* __restrict__ pw,
                                                     unsigned long long* __restrict__ counts4) {
    if (threadIdx.x < 4) counts4[threadIdx.x] = 0ull;
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 9; ++i) { pw[PW_E + i] = ek.A[i]; pw[PW_K + i] = ek.B[i]; }
    pose_prepare(ek.A, ek.B, pw);
}

// pair pipeline middle step: denormalise the RANSAC winner (custom_slam.py:101: T1^T F T0), F -> E, candidates
__global__ __launch_bounds__(64) void k_pair_mid(Mat3x2 kk /* A = K */, const double* __restrict__ record,
                                                 double* __restrict__ pw, unsigned long long* __restrict__ counts4,
                                                 const PairArgs* __restrict__ batch) {
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        record = a.record; pw = a.pw; counts4 = a.counts4;
    }
    if (threadIdx.x < 4) counts4[threadIdx.x] = 0ull;
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 9; ++i) pw[PW_K + i] = kk.A[i];
    const double* Fn = record + SLAM_REC_F;
    const double* T0 = pw + PW_T0;
    const double* T1 = pw + PW_T1;
    double A[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r][c] = T1[r] * Fn[c] + T1[3 + r] * Fn[3 + c] + T1[6 + r] * Fn[6 + c];  // T1^T F
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) pw[PW_F + 3 * r + c] = A[r][0] * T0[c] + A[r][1] * T0[3 + c] + A[r][2] * T0[6 + c];
    double E[9];
    f_to_e(pw + PW_F, kk.A, E);
    for (int i = 0; i < 9; ++i) pw[PW_E + i] = E[i];
    pose_prepare(E, kk.A, pw);
}

__global__ __launch_bounds__(64) void k_f_to_e(Mat3x2 fk /* A = F, B = K */, double* __restrict__ Eout) {
    if (threadIdx.x == 0) f_to_e(fk.A, fk.B, Eout);
}

// winner = first candidate with the strictly largest count   (reference: src/epipolar.py:129-133)
__global__ __launch_bounds__(64) void k_pose_select(const double* __restrict__ pw,
                                                    const unsigned long long* __restrict__ counts4,
                                                    double* __restrict__ pose, int32_t* __restrict__ winner) {
    __shared__ int s_w;
    if (threadIdx.x == 0) {
        long long best = -1;
        int w = 0;
        for (int c = 0; c < 4; ++c) {
            const long long n = (long long)counts4[c];
            if (n > best) { best = n; w = c; }
        }
        s_w = w;
        *winner = w;
    }
    __syncthreads();
    if (threadIdx.x < 16) pose[threadIdx.x] = pw[PW_CAND + s_w * PW_CAND_LEN + PW_CAND_POSE + threadIdx.x];
}

// copy the winner's plane to the output and optionally flag X_z > 0 (custom_slam.py:114)
__global__ __launch_bounds__(256) void k_emit_points(const double* __restrict__ X4, const int32_t* __restrict__ winner,
                                                     int64_t N, double* __restrict__ X, uint8_t* __restrict__ good) {
    const int w = *winner;
    const double* src = X4 + (size_t)(w >> 1) * (size_t)N * 3;  // two planes: candidate 2 r + 1 is the negative of 2 r
    const int64_t total = N * 3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const double v = (w & 1) ? -src[e] : src[e];
        X[e] = v;
        if (good && (e % 3) == 2) good[e / 3] = v > 0.0 ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// normalize_points   (reference: custom_slam.py:85-93)
//   avg = mean(p); s = mean(||p - avg||) / sqrt(2); out = (p - avg)/s; T = [[s,0,-s ax],[0,s,-s ay],[0,0,1]]
// Single workgroup, fixed-order reductions.
// ------------------------------------------------------------------------------------------------
// Sum over the 1024 threads of a workgroup, the same on every thread: butterfly inside the wave (wave_sum_f64: the
// operands of every addition are those of an xor-shuffle butterfly, so the bits are too), then the 16 wave totals in
// order.  Every call needs 16 doubles of LDS of its own (s_red), so that no barrier has to protect their reuse.
__device__ __forceinline__ double block_sum_1024(double v, double* s_red) {
    const int tid = threadIdx.x;
    v = wave_sum_f64(v);
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_red[w];
    return t;
}
// two sums behind one barrier
__device__ __forceinline__ void block_sum2_1024(double& a, double& b, double* s_red_a, double* s_red_b) {
    const int tid = threadIdx.x;
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if ((tid & 63) == 0) { s_red_a[tid >> 6] = a; s_red_b[tid >> 6] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) { ta += s_red_a[w]; tb += s_red_b[w]; }
    a = ta;
    b = tb;
}

__global__ __launch_bounds__(1024) void k_normalize(const double2* __restrict__ p, int N, double2* __restrict__ out,
                                                    double* __restrict__ T) {
    __shared__ double s_red[3][16];
    const int tid = threadIdx.x;
    double sx = 0.0, sy = 0.0;
    for (int i = tid; i < N; i += 1024) { const double2 v = p[i]; sx += v.x; sy += v.y; }
    block_sum2_1024(sx, sy, s_red[0], s_red[1]);
    const double ax = sx / (double)N;
    const double ay = sy / (double)N;
    double sd = 0.0;
    for (int i = tid; i < N; i += 1024) {
        const double2 v = p[i];
        const double dx = v.x - ax, dy = v.y - ay;
        sd += sqrt(dx * dx + dy * dy);
    }
    const double s = (block_sum_1024(sd, s_red[2]) / (double)N) / sqrt(2.0);
    for (int i = tid; i < N; i += 1024) {
        const double2 v = p[i];
        out[i] = make_double2((v.x - ax) / s, (v.y - ay) / s);
    }
    if (tid == 0) {
        T[0] = s;   T[1] = 0.0; T[2] = -1.0 * s * ax;
        T[3] = 0.0; T[4] = s;   T[5] = -1.0 * s * ay;
        T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
    }
}

// both images of a pair in one launch: workgroup 0 normalises pa into outa / Ta, workgroup 1 pb into outb / Tb
__global__ __launch_bounds__(1024) void k_normalize_pair(const double2* __restrict__ pa, const double2* __restrict__ pb,
                                                         int N, double2* __restrict__ outa, double2* __restrict__ outb,
                                                         double* __restrict__ Ta, double* __restrict__ Tb,
                                                         const PairArgs* __restrict__ batch) {
    __shared__ double s_red[3][16];
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        pa = a.r0; pb = a.r1; N = a.N; outa = a.n0; outb = a.n1; Ta = a.pw + PW_T0; Tb = a.pw + PW_T1;
    }
    const bool second = blockIdx.x != 0;
    const double2* __restrict__ p = second ? pb : pa;
    double2* __restrict__ out = second ? outb : outa;
    double* __restrict__ T = second ? Tb : Ta;
    const int tid = threadIdx.x;
    double sx = 0.0, sy = 0.0;
    for (int i = tid; i < N; i += 1024) { const double2 v = p[i]; sx += v.x; sy += v.y; }
    block_sum2_1024(sx, sy, s_red[0], s_red[1]);
    const double ax = sx / (double)N;
    const double ay = sy / (double)N;
    double sd = 0.0;
    for (int i = tid; i < N; i += 1024) {
        const double2 v = p[i];
        const double dx = v.x - ax, dy = v.y - ay;
        sd += sqrt(dx * dx + dy * dy);
    }
    const double s = (block_sum_1024(sd, s_red[2]) / (double)N) / sqrt(2.0);
    for (int i = tid; i < N; i += 1024) {
        const double2 v = p[i];
        out[i] = make_double2((v.x - ax) / s, (v.y - ay) / s);
    }
    if (tid == 0) {
        T[0] = s;   T[1] = 0.0; T[2] = -1.0 * s * ax;
        T[3] = 0.0; T[4] = s;   T[5] = -1.0 * s * ay;
        T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
    }
}

// selection and emission in one launch (pair pipelines): every workgroup finds the winner from the four counts, copies
// its share of the winner's plane and flags X_z > 0; workgroup 0 also writes the pose and the winner's position
__global__ __launch_bounds__(256) void k_select_emit(const double* __restrict__ pw,
                                                     const unsigned long long* __restrict__ counts4,
                                                     const double* __restrict__ X4, int64_t N, double* __restrict__ pose,
                                                     int32_t* __restrict__ winner, double* __restrict__ X,
                                                     uint8_t* __restrict__ good, const PairArgs* __restrict__ batch) {
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        pw = a.pw; counts4 = a.counts4; X4 = a.X4; N = a.N; pose = a.pose; winner = a.winner; X = a.X; good = a.good;
    }
    long long best = -1;
    int w = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long long n = (long long)counts4[c];
        if (n > best) { best = n; w = c; }  // strict: the first maximum   (reference: src/epipolar.py:129-133)
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *winner = w;
        if (threadIdx.x < 16) pose[threadIdx.x] = pw[PW_CAND + w * PW_CAND_LEN + PW_CAND_POSE + threadIdx.x];
    }
    const double* src = X4 + (size_t)(w >> 1) * (size_t)N * 3;  // two planes: candidate 2 r + 1 is the negative of 2 r
    const int64_t total = N * 3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const double v = (w & 1) ? -src[e] : src[e];
        X[e] = v;
        if (good && (e % 3) == 2) good[e / 3] = v > 0.0 ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Over-determined eight-point solve, M > 8 rows   (reference: src/epipolar.py:37-51 with len(points) > 8)
// The least right singular vector of the M x 9 design matrix equals that of its 9 x 9 triangular factor R:
// rows are folded into R with Givens rotations, then R goes through a one-sided Jacobi SVD.  One lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_eightpoint_general(const double2* __restrict__ p0, const double2* __restrict__ p1,
                                                           int M, double* __restrict__ Fout) {
    __shared__ double R[9][9];
    __shared__ double V[9][9];
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) { R[i][j] = 0.0; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int i = 0; i < M; ++i) {
        const double2 a = p0[i], b = p1[i];
        double row[9] = {b.x * a.x, b.x * a.y, b.x, b.y * a.x, b.y * a.y, b.y, a.x, a.y, 1.0};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double x = R[k][k], y = row[k];
            if (y != 0.0) {
                const double r = sqrt(x * x + y * y);
                const double c = x / r, s = y / r;
                R[k][k] = r;
#pragma unroll
                for (int j = k + 1; j < 9; ++j) {
                    const double t1 = R[k][j], t2 = row[j];
                    R[k][j] = c * t1 + s * t2;
                    row[j] = c * t2 - s * t1;
                }
            }
        }
    }
    double fro2 = 0.0;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) fro2 += R[i][j] * R[i][j];
    const double floor2 = fro2 * 1e-32;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                double a = 0.0, b = 0.0, c = 0.0;
                for (int i = 0; i < 9; ++i) {
                    a = fma(R[i][p], R[i][p], a);
                    b = fma(R[i][q], R[i][q], b);
                    c = fma(R[i][p], R[i][q], c);
                }
                if (!(a > floor2 && b > floor2 && c * c > 1e-30 * (a * b))) continue;
                rotated = true;
                double cs, sn;
                jacobi_angle(a, b, c, cs, sn);
                for (int i = 0; i < 9; ++i) {
                    const double x = R[i][p], y = R[i][q];
                    R[i][p] = fma(cs, x, -sn * y);
                    R[i][q] = fma(sn, x, cs * y);
                    const double vx = V[i][p], vy = V[i][q];
                    V[i][p] = fma(cs, vx, -sn * vy);
                    V[i][q] = fma(sn, vx, cs * vy);
                }
            }
        if (!rotated) break;
    }
    int least = 0;
    double best = 1e300;
    for (int j = 0; j < 9; ++j) {
        double s = 0.0;
        for (int i = 0; i < 9; ++i) s += R[i][j] * R[i][j];
        if (s < best) { best = s; least = j; }
    }
    double f[9], F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) f[i] = V[i][least];
    rank2_project(f, F);
#pragma unroll
    for (int i = 0; i < 9; ++i) Fout[i] = F[i];
}

}  // namespace slamk
