/*
 * CPU restatement in plain C (float64) of the reference's two-view initialisation path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, the smoke check and the
 * benchmark's CPU-baseline leg load this library; nothing under slam_amd/ links or calls it.
 * It is pinned against golden vectors produced by the reference itself (the .npz files under tests/golden).
 *
 * Where the reference calls LAPACK's SVD this file uses a one-sided (Hestenes) Jacobi SVD,
 * which delivers the same singular subspaces to a few ulps; everything else is the
 * reference's arithmetic, statement by statement.  Reference lines are cited per function
 * (paths relative to the reference tree).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MT_N 624
#define MT_M 397

/* ------------------------------------------------------------------------------------
 * CPython's random.seed(int) + random.sample(range(N), k)
 * src/epipolar.py:70-71 ; CPython Modules/_randommodule.c (init_genrand, init_by_array,
 * genrand_uint32) ; Lib/random.py (_randbelow_with_getrandbits, sample)
 * ---------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[MT_N]; int idx; } mt_t;

static void mt_init_genrand(mt_t* g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < MT_N; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = MT_N;
}

static void mt_init_by_array(mt_t* g, const uint32_t* key, int klen) {
    mt_init_genrand(g, 19650218u);
    uint32_t* mt = g->mt;
    int i = 1, j = 0;
    for (int k = (MT_N > klen ? MT_N : klen); k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

static uint32_t mt_next(mt_t* g) {
    uint32_t* mt = g->mt;
    if (g->idx >= MT_N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g->idx = 0;
    }
    uint32_t y = mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static int bit_length(uint32_t n) { int b = 0; while (n) { b++; n >>= 1; } return b; }

static uint32_t mt_randbelow(mt_t* g, uint32_t n) {
    int k = bit_length(n);
    uint32_t r = mt_next(g) >> (32 - k);
    while (r >= n) r = mt_next(g) >> (32 - k);
    return r;
}

/* idx[h*k .. h*k+k) = sample after seed(seed_base + it0 + h).  N < 2^31, k <= 64. */
int oracle_sample(uint64_t seed_base, int64_t it0, int H, int N, int k, int32_t* idx) {
    if (N < k || k < 1 || k > 64) return -1;
    /* setsize rule of Lib/random.py sample(): 21, plus 4**ceil(log(3k, 4)) when k > 5 */
    int setsize = 21;
    if (k > 5) { int p = 1; while (p < 3 * k) p *= 4; setsize += p; }
    int32_t* pool = (N <= setsize) ? (int32_t*)malloc(sizeof(int32_t) * (size_t)N) : NULL;
    mt_t g;
    for (int h = 0; h < H; h++) {
        uint64_t seed = seed_base + (uint64_t)(it0 + h);
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
        mt_init_by_array(&g, key, key[1] ? 2 : 1);
        int32_t* out = idx + (size_t)h * k;
        if (pool) {
            for (int i = 0; i < N; i++) pool[i] = i;
            for (int i = 0; i < k; i++) {
                uint32_t j = mt_randbelow(&g, (uint32_t)(N - i));
                out[i] = pool[j];
                pool[j] = pool[N - i - 1];
            }
        } else {
            for (int i = 0; i < k; i++) {
                uint32_t j;
                int dup;
                do {
                    j = mt_randbelow(&g, (uint32_t)N);
                    dup = 0;
                    for (int q = 0; q < i; q++) if (out[q] == (int32_t)j) { dup = 1; break; }
                } while (dup);
                out[i] = (int32_t)j;
            }
        }
    }
    free(pool);
    return 0;
}

/* raw tempered outputs following seed(seed_base + it): out[h*count .. ) */
int oracle_mt_raw(uint64_t seed_base, int64_t it0, int H, int count, uint32_t* out) {
    mt_t g;
    for (int h = 0; h < H; h++) {
        uint64_t seed = seed_base + (uint64_t)(it0 + h);
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
        mt_init_by_array(&g, key, key[1] ? 2 : 1);
        for (int c = 0; c < count; c++) out[(size_t)h * count + c] = mt_next(&g);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * One-sided Jacobi SVD.  A is m x n row-major and is overwritten by A*V (its columns are
 * then sigma_i * u_i); V (n x n row-major) receives the right singular vectors as columns.
 * ---------------------------------------------------------------------------------- */
static void jacobi_svd(double* A, int m, int n, double* V) {
    double fro2 = 0;
    for (int i = 0; i < m * n; i++) fro2 += A[i] * A[i];
    const double floor2 = fro2 * 1e-32;          /* columns at rounding level are left alone */
    for (int i = 0; i < n * n; i++) V[i] = 0;
    for (int i = 0; i < n; i++) V[i * n + i] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < m; i++) {
                    double x = A[i * n + p], y = A[i * n + q];
                    a += x * x; b += y * y; c += x * y;
                }
                if (a <= floor2 || b <= floor2) continue;
                if (fabs(c) <= 1e-15 * sqrt(a * b)) continue;
                rotated = 1;
                double zeta = (b - a) / (2.0 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; i++) {
                    double x = A[i * n + p], y = A[i * n + q];
                    A[i * n + p] = cs * x - sn * y;
                    A[i * n + q] = sn * x + cs * y;
                }
                for (int i = 0; i < n; i++) {
                    double x = V[i * n + p], y = V[i * n + q];
                    V[i * n + p] = cs * x - sn * y;
                    V[i * n + q] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
}

static void column_norms2(const double* A, int m, int n, double* out) {
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < m; i++) s += A[i * n + j] * A[i * n + j];
        out[j] = s;
    }
}

/* ------------------------------------------------------------------------------------
 * eight-point solve: src/epipolar.py:18-51
 * ---------------------------------------------------------------------------------- */
static void rank2_project(const double* f_in, double* F) {
    /* :43 normalise by the Frobenius norm; :47-49 zero the smallest singular value.
       With B = f V (columns sigma_i u_i):  U diag(s0,s1,0) V^T = sum over the two largest of B_i v_i^T */
    double f[9], V[9], nrm = 0, n2[3];
    for (int i = 0; i < 9; i++) nrm += f_in[i] * f_in[i];
    nrm = sqrt(nrm);
    for (int i = 0; i < 9; i++) f[i] = f_in[i] / nrm;
    jacobi_svd(f, 3, 3, V);
    column_norms2(f, 3, 3, n2);
    int drop = 0;
    for (int j = 1; j < 3; j++) if (n2[j] < n2[drop]) drop = j;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int j = 0; j < 3; j++) if (j != drop) s += f[r * 3 + j] * V[c * 3 + j];
            F[r * 3 + c] = s;
        }
}

/* p0, p1: M x 2 (img0 = a, img1 = b); F: 3x3 row-major */
int oracle_eight_point(const double* p0, const double* p1, int M, double* F) {
    if (M < 8) return -1;                                         /* :23 */
    int rows = M < 9 ? 9 : M;                                     /* pad 8x9 with a zero row so m >= n */
    double* C = (double*)calloc((size_t)rows * 9, sizeof(double));
    double V[81], n2[9];
    for (int i = 0; i < M; i++) {                                  /* :37 */
        double a0 = p0[2 * i], a1 = p0[2 * i + 1], b0 = p1[2 * i], b1 = p1[2 * i + 1];
        double* c = C + (size_t)i * 9;
        c[0] = b0 * a0; c[1] = b0 * a1; c[2] = b0;
        c[3] = b1 * a0; c[4] = b1 * a1; c[5] = b1;
        c[6] = a0; c[7] = a1; c[8] = 1.0;
    }
    jacobi_svd(C, rows, 9, V);                                    /* :39 */
    column_norms2(C, rows, 9, n2);
    int least = 0;
    for (int j = 1; j < 9; j++) if (n2[j] < n2[least]) least = j;
    double f[9];
    for (int i = 0; i < 9; i++) f[i] = V[i * 9 + least];          /* vh[-1]  :40 */
    free(C);
    rank2_project(f, F);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * RANSAC scoring and selection: src/epipolar.py:54-84 (the index sets are an input)
 * ---------------------------------------------------------------------------------- */
static void score_one(const double* F, const double* p0, const double* p1, int N, double thr,
                      int32_t* count, double* errsum, uint8_t* mask) {
    int32_t c = 0;
    double e = 0;
    for (int i = 0; i < N; i++) {
        double a0 = p0[2 * i], a1 = p0[2 * i + 1], b0 = p1[2 * i], b1 = p1[2 * i + 1];
        double l0 = b0 * F[0] + b1 * F[3] + F[6];                 /* x1^T F        :73 */
        double l1 = b0 * F[1] + b1 * F[4] + F[7];
        double l2 = b0 * F[2] + b1 * F[5] + F[8];
        double r = l0 * a0 + l1 * a1 + l2;                        /* ... @ x0      */
        double err = r * r;                                       /* np.square     */
        int in = err < thr;                                       /* :75 strict    */
        if (in) { c++; e += err; }                                /* :76-77 */
        if (mask) mask[i] = (uint8_t)in;
    }
    *count = c; *errsum = e;
}

int oracle_ransac(const double* p0, const double* p1, int N, const int32_t* idx, int H, double thr,
                  double* F_all /* H*9 or NULL */, int32_t* counts /* H */, double* errsums /* H */,
                  int32_t* best_index, double* best_F /* 9 */, uint8_t* best_mask /* N or NULL */, int threads) {
    if (N < 8 || H < 1) return -1;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int h = 0; h < H; h++) {
        double a[16], b[16], F[9];
        for (int k = 0; k < 8; k++) {
            int32_t i = idx[(size_t)h * 8 + k];
            a[2 * k] = p0[2 * i]; a[2 * k + 1] = p0[2 * i + 1];
            b[2 * k] = p1[2 * i]; b[2 * k + 1] = p1[2 * i + 1];
        }
        oracle_eight_point(a, b, 8, F);                           /* :72 */
        if (F_all) memcpy(F_all + (size_t)h * 9, F, sizeof F);
        score_one(F, p0, p1, N, thr, &counts[h], &errsums[h], NULL);
    }
    int32_t best = -1, bi = -1;
    double be = 0;
    for (int h = 0; h < H; h++)                                   /* :78-81, in iteration order */
        if (counts[h] > best || (counts[h] == best && errsums[h] < be)) { best = counts[h]; be = errsums[h]; bi = h; }
    *best_index = bi;
    if (best_F || best_mask) {
        double a[16], b[16], F[9];
        int32_t c; double e;
        for (int k = 0; k < 8; k++) {
            int32_t i = idx[(size_t)bi * 8 + k];
            a[2 * k] = p0[2 * i]; a[2 * k + 1] = p0[2 * i + 1];
            b[2 * k] = p1[2 * i]; b[2 * k + 1] = p1[2 * i + 1];
        }
        oracle_eight_point(a, b, 8, F);
        if (best_F) memcpy(best_F, F, sizeof F);
        if (best_mask) score_one(F, p0, p1, N, thr, &c, &e, best_mask);
    }
    return 0;
}

/* Timing leg of the benchmark's CPU baseline: one untimed pass, then `reps` timed passes of oracle_ransac on the
 * monotonic clock; returns the best and the mean time of a pass in seconds (what a stopwatch around the foreign-function
 * call would add is left out). */
#include <time.h>
int oracle_ransac_timed(const double* p0, const double* p1, int N, const int32_t* idx, int H, double thr, int threads,
                        int reps, int32_t* counts /* H */, double* errsums /* H */, int32_t* best_index,
                        double* best_seconds, double* mean_seconds) {
    if (reps < 1) return -1;
    int rc = oracle_ransac(p0, p1, N, idx, H, thr, NULL, counts, errsums, best_index, NULL, NULL, threads);
    if (rc) return rc;
    double best = 1e300, total = 0.0;
    for (int r = 0; r < reps; r++) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        rc = oracle_ransac(p0, p1, N, idx, H, thr, NULL, counts, errsums, best_index, NULL, NULL, threads);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc) return rc;
        const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (dt < best) best = dt;
        total += dt;
    }
    *best_seconds = best;
    *mean_seconds = total / reps;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * caller-side normalisation: custom_slam.py:85-93
 * ---------------------------------------------------------------------------------- */
int oracle_normalize_points(const double* p, int N, double* out /* N*2 */, double* T /* 9 */) {
    double ax = 0, ay = 0, d = 0;
    for (int i = 0; i < N; i++) { ax += p[2 * i]; ay += p[2 * i + 1]; }
    ax /= N; ay /= N;
    for (int i = 0; i < N; i++) d += hypot(p[2 * i] - ax, p[2 * i + 1] - ay);
    double s = (d / N) / sqrt(2.0);
    for (int i = 0; i < N; i++) { out[2 * i] = (p[2 * i] - ax) / s; out[2 * i + 1] = (p[2 * i + 1] - ay) / s; }
    double t[9] = { s, 0, -s * ax, 0, s, -s * ay, 0, 0, 1 };      /* s where 1/s would be expected: :90-92 */
    memcpy(T, t, sizeof t);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * triangulation: src/triangulation.py:49-73.  P0, P1: 3x4 row-major (first three rows of
 * whatever the caller holds); X: N x 3.
 * ---------------------------------------------------------------------------------- */
static void triangulate_one(const double* P0, const double* P1, double u0, double v0, double u1, double v1, double* X) {
    double D[16], V[16], m[4], n2[4];
    for (int j = 0; j < 4; j++) {
        D[0 + j] = -1 * P0[4 + j] + v0 * P0[8 + j];               /* :50 */
        D[4 + j] = P0[0 + j] - u0 * P0[8 + j];                    /* :51 */
        D[8 + j] = -1 * P1[4 + j] + v1 * P1[8 + j];               /* :52 */
        D[12 + j] = P1[0 + j] - u1 * P1[8 + j];                   /* :53 */
    }
    for (int i = 0; i < 4; i++) {                                  /* row maxima :58 */
        double mx = 0;
        for (int j = 0; j < 4; j++) { double a = fabs(D[4 * i + j]); if (a > mx) mx = a; }
        m[i] = mx;
    }
    for (int i = 0; i < 4; i++)                                    /* D @ diag(1/m): column j times 1/m_j  :60-62 */
        for (int j = 0; j < 4; j++) D[4 * i + j] *= (1.0 / m[j]);
    jacobi_svd(D, 4, 4, V);                                        /* :64 */
    column_norms2(D, 4, 4, n2);
    int least = 0;
    for (int j = 1; j < 4; j++) if (n2[j] < n2[least]) least = j;
    double Y[4];
    for (int i = 0; i < 4; i++) Y[i] = m[i] * V[4 * i + least];    /* diag(m) @ vh[-1]  :66-67 */
    X[0] = Y[0] / Y[3]; X[1] = Y[1] / Y[3]; X[2] = Y[2] / Y[3];    /* :70-73 */
}

int oracle_triangulate(const double* P0, const double* P1, const double* p0, const double* p1, int64_t N,
                       double* X, int threads) {
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; i++)
        triangulate_one(P0, P1, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], X + 3 * i);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * pose recovery: src/epipolar.py:95-136, src/util.py:30-37
 * The candidate ORDER depends on the sign choices of the SVD routine; this file orders them by
 * its own SVD, so only order-independent results (winner pose, points, sorted counts) are
 * comparable with LAPACK-based runs.
 * ---------------------------------------------------------------------------------- */
static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* pose = [[R^T, -R^T t],[0,0,0,1]]  (src/util.py:30-37) */
static void rt_to_pose(const double* R, const double* t, double* pose) {
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int j = 0; j < 3; j++) { pose[4 * i + j] = R[3 * j + i]; s += R[3 * j + i] * t[j]; }
        pose[4 * i + 3] = -s;
    }
    pose[12] = pose[13] = pose[14] = 0; pose[15] = 1;
}

/* Rs: 4 x 9, ts: 4 x 3 in the order (R0,t0),(R0,t1),(R1,t0),(R1,t1) */
int oracle_pose_candidates(const double* E, double* Rs, double* ts) {
    double B[9], V[9], n2[3], U[9], s[3];
    memcpy(B, E, sizeof B);
    jacobi_svd(B, 3, 3, V);
    column_norms2(B, 3, 3, n2);
    int o[3] = { 0, 1, 2 };                                        /* descending singular values */
    for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (n2[o[j]] > n2[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    double Vs[9];
    for (int c = 0; c < 3; c++) {
        s[c] = sqrt(n2[o[c]]);
        for (int r = 0; r < 3; r++) { Vs[3 * r + c] = V[3 * r + o[c]]; U[3 * r + c] = B[3 * r + o[c]] / s[c]; }
    }
    /* sigma_3 is zero to rounding: complete U with the cross product of its first two columns */
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    const double W[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 };
    double Vt[9], tmp[9], R0[9], R1[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vt[3 * r + c] = Vs[3 * c + r];
    mat3_mul(U, W, tmp); mat3_mul(tmp, Vt, R0);                    /* :102 */
    mat3_mul(U, Wt, tmp); mat3_mul(tmp, Vt, R1);
    double mult = det3(R0) > 0 ? 1.0 : -1.0;                       /* :103-104 */
    for (int i = 0; i < 9; i++) { R0[i] *= mult; R1[i] *= mult; }
    double t0[3] = { U[2], U[5], U[8] }, t1[3] = { -U[2], -U[5], -U[8] };   /* :107 */
    memcpy(Rs, R0, sizeof R0); memcpy(Rs + 9, R0, sizeof R0); memcpy(Rs + 18, R1, sizeof R1); memcpy(Rs + 27, R1, sizeof R1);
    memcpy(ts, t0, sizeof t0); memcpy(ts + 3, t1, sizeof t1); memcpy(ts + 6, t0, sizeof t0); memcpy(ts + 9, t1, sizeof t1);
    return 0;
}

int oracle_pose_from_E(const double* E, const double* K, const double* p0, const double* p1, int64_t N,
                       double* pose /* 16 */, double* X /* N*3 */, int32_t* counts4, int32_t* winner, int threads) {
    double Rs[36], ts[12], P0[12], P1[12], pose1[16];
    oracle_pose_candidates(E, Rs, ts);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) P0[4 * i + j] = j < 3 ? K[3 * i + j] : 0.0;   /* K [I|0]  :117 */
    double* Xc = (double*)malloc(sizeof(double) * 3 * (size_t)(N > 0 ? N : 1));
    int32_t best = -1;
    *winner = -1;
    for (int c = 0; c < 4; c++) {
        const double* R = Rs + 9 * c;
        const double* t = ts + 3 * c;
        rt_to_pose(R, t, pose1);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) {                                       /* K pose[:3]  :118 */
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[3 * i + k] * pose1[4 * k + j];
            P1[4 * i + j] = s;
        }
        oracle_triangulate(P0, P1, p0, p1, N, Xc, threads);                                              /* :119 */
        double cz[3] = { R[2], R[5], R[8] };                                                             /* R @ (0,0,1)  :123 */
        int32_t cnt = 0;
        for (int64_t i = 0; i < N; i++) {
            const double* x = Xc + 3 * i;
            int f1 = x[2] > 0;                                                                           /* :125 */
            int f2 = ((x[0] - t[0]) * cz[0] + (x[1] - t[1]) * cz[1] + (x[2] - t[2]) * cz[2]) > 0;        /* :126 */
            cnt += f1 && f2;
        }
        counts4[c] = cnt;
        if (cnt > best) {                                                                                /* :129 strict */
            best = cnt; *winner = c;
            memcpy(pose, pose1, sizeof pose1);
            memcpy(X, Xc, sizeof(double) * 3 * (size_t)N);
        }
    }
    free(Xc);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * tracking step: camera pose from 2D-3D matches and its RANSAC loop
 * src/triangulation.py:78-182, src/util.py:18-27,60-66
 * ---------------------------------------------------------------------------------- */
/* uv: M x 2 image points, X: M x 3 known points.  pose: 4x4.  extra (may be NULL): 9 (calculated K) + 9 (rotation)
 * + 3 (translation) + 3 (camera centre), the quantities the reference prints. */
int oracle_pose_from_points(const double* uv, const double* X, int M, double* pose, double* extra) {
    if (M < 6) return -1;                                                                                /* :82 */
    const int rows = 2 * M;
    double* A = (double*)calloc((size_t)rows * 12, sizeof(double));
    double V[144], n2[12];
    for (int i = 0; i < M; i++) {                                                                        /* :98-100 */
        const double u = uv[2 * i], v = uv[2 * i + 1];
        const double xh[4] = { X[3 * i], X[3 * i + 1], X[3 * i + 2], 1.0 };
        double* e = A + (size_t)(2 * i) * 12;
        double* o = e + 12;
        for (int j = 0; j < 4; j++) {
            e[4 + j] = -xh[j]; e[8 + j] = v * xh[j];
            o[j] = xh[j];      o[8 + j] = -u * xh[j];
        }
    }
    jacobi_svd(A, rows, 12, V);                                                                          /* :112 */
    column_norms2(A, rows, 12, n2);
    free(A);
    int least = 0;
    for (int j = 1; j < 12; j++) if (n2[j] < n2[least]) least = j;
    double cam[12];
    for (int i = 0; i < 12; i++) cam[i] = V[i * 12 + least];                                             /* :113-114 */
    const double m3[9] = { cam[0], cam[1], cam[2], cam[4], cam[5], cam[6], cam[8], cam[9], cam[10] };
    const double d = det3(m3);
    const double sg = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);                                                /* :115 */
    for (int i = 0; i < 12; i++) cam[i] *= sg;
    /* :117-118  the null vector of the 3x4 matrix, by cofactors; only its ratios are used */
    double c4[4];
    for (int k = 0; k < 4; k++) {
        double s[9];
        int cc = 0;
        for (int j = 0; j < 4; j++) {
            if (j == k) continue;
            s[cc] = cam[j]; s[3 + cc] = cam[4 + j]; s[6 + cc] = cam[8 + j];
            cc++;
        }
        c4[k] = ((k & 1) ? -1.0 : 1.0) * det3(s);
    }
    const double centre[3] = { c4[0] / c4[3], c4[1] / c4[3], c4[2] / c4[3] };
    /* :120  rq_decomposition(m) of src/util.py:60-66: J m = q' r' (Gram-Schmidt here; the signs LAPACK would pick
       cancel in :122-124), r = J r'^T J, q = J q'^T */
    double Mp[9], Q[9], Rp[9] = { 0 };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mp[3 * i + j] = sg * m3[3 * (2 - i) + j];
    for (int j = 0; j < 3; j++) {
        double v[3] = { Mp[j], Mp[3 + j], Mp[6 + j] };
        for (int pass = 0; pass < 2; pass++)                         /* twice: re-orthogonalisation */
            for (int k = 0; k < j; k++) {
                double dot = 0;
                for (int i = 0; i < 3; i++) dot += Q[3 * i + k] * v[i];
                Rp[3 * k + j] += dot;
                for (int i = 0; i < 3; i++) v[i] -= dot * Q[3 * i + k];
            }
        const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        Rp[3 * j + j] = nrm;
        for (int i = 0; i < 3; i++) Q[3 * i + j] = v[i] / nrm;
    }
    double r[9], q[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        r[3 * i + j] = Rp[3 * (2 - j) + (2 - i)];                     /* (J r'^T J)_ij = r'_(2-j)(2-i) */
        q[3 * i + j] = Q[3 * j + (2 - i)];                            /* (J q'^T)_ij = q'_j(2-i) */
    }
    double dg[3], Kc[9], R[9], t[3];
    for (int i = 0; i < 3; i++) dg[i] = r[4 * i] > 0 ? 1.0 : (r[4 * i] < 0 ? -1.0 : 0.0);                /* :122 */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Kc[3 * i + j] = r[3 * i + j] * dg[j]; R[3 * i + j] = dg[i] * q[3 * i + j]; }
    for (int i = 0; i < 3; i++) t[i] = -1 * (R[3 * i] * centre[0] + R[3 * i + 1] * centre[1] + R[3 * i + 2] * centre[2]);   /* :125 */
    rt_to_pose(R, t, pose);                                                                              /* :132 */
    if (extra) { memcpy(extra, Kc, sizeof Kc); memcpy(extra + 9, R, sizeof R); memcpy(extra + 18, t, sizeof t); memcpy(extra + 21, centre, sizeof centre); }
    return 0;
}

static void mat34_from_K_pose(const double* K, const double* pose, double* P) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += K[3 * i + k] * pose[4 * k + j];
        P[4 * i + j] = s;
    }
}

/* idx: H x 6 sample indices.  a, b: M x 2 previous/next image points of the matches with known 3D position X3 (M x 3).
 * Outputs: poses (H x 16), counts (H), errs (H), best (winner position), any of the first three may be NULL. */
int oracle_tracking_ransac(const double* prev_pose, const double* a, const double* b, const double* X3, int M,
                           const double* K, const int32_t* idx, int H, double* poses, int32_t* counts, double* errs,
                           int32_t* best) {
    if (M < 6 || H < 1) return -1;
    double P0[12], P1[12], pose[16];
    double* Xt = (double*)malloc(sizeof(double) * 3 * (size_t)M);
    mat34_from_K_pose(K, prev_pose, P0);
    int32_t bc = -1, bi = -1;
    double be = 0;
    for (int h = 0; h < H; h++) {
        double uv[12], xs[18];
        for (int j = 0; j < 6; j++) {
            const int32_t i = idx[6 * h + j];
            uv[2 * j] = b[2 * i]; uv[2 * j + 1] = b[2 * i + 1];
            xs[3 * j] = X3[3 * i]; xs[3 * j + 1] = X3[3 * i + 1]; xs[3 * j + 2] = X3[3 * i + 2];
        }
        oracle_pose_from_points(uv, xs, 6, pose, NULL);                                                  /* :156-158 */
        mat34_from_K_pose(K, pose, P1);
        oracle_triangulate(P0, P1, a, b, M, Xt, 1);                                                      /* :159-162 */
        /* rotation = pose[:3,:3]^T, translation = -pose[:3,:3]^T pose[:3,3]   (src/util.py:18-27) */
        const double cz[3] = { pose[8], pose[9], pose[10] };                                             /* R (0,0,1)^T  :166 */
        double t[3];
        for (int i = 0; i < 3; i++) t[i] = -1 * (pose[i] * pose[3] + pose[4 + i] * pose[7] + pose[8 + i] * pose[11]);
        int32_t c = 0;
        double e = 0;
        for (int i = 0; i < M; i++) {
            const double* x = Xt + 3 * i;
            const int f1 = x[2] > 0;                                                                     /* :167 */
            const int f2 = ((x[0] - t[0]) * cz[0] + (x[1] - t[1]) * cz[1] + (x[2] - t[2]) * cz[2]) > 0;  /* :168 */
            c += f1 && f2;
            const double dx = X3[3 * i] - x[0], dy = X3[3 * i + 1] - x[1], dz = X3[3 * i + 2] - x[2];
            e += sqrt(dx * dx + dy * dy + dz * dz);                                                      /* :170 */
        }
        if (poses) memcpy(poses + 16 * (size_t)h, pose, sizeof pose);
        if (counts) counts[h] = c;
        if (errs) errs[h] = e;
        if (c > bc || (c == bc && be > e)) { bc = c; be = e; bi = h; }                                   /* :172-173 */
    }
    free(Xt);
    *best = bi;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * descriptor matching: custom_slam.py:14-26 (brute-force Hamming k-NN, stable order) and :62-73 (ratio test)
 * des0: n0 x 32 bytes, des1: n1 x 32 bytes (packed 256-bit descriptors).  index/distance: n0 x k, entries beyond
 * the number of train descriptors are -1.
 * ---------------------------------------------------------------------------------- */
int oracle_match_knn(const uint8_t* des0, const uint8_t* des1, int n0, int n1, int k, int32_t* index, int32_t* distance) {
    if (k < 1 || k > 16) return -1;
    for (int q = 0; q < n0; q++) {
        int32_t bi[16], bd[16];
        int have = 0;
        uint64_t a[4];
        memcpy(a, des0 + 32 * (size_t)q, 32);
        for (int j = 0; j < n1; j++) {
            uint64_t b[4];
            memcpy(b, des1 + 32 * (size_t)j, 32);
            int32_t d = 0;
            for (int w = 0; w < 4; w++) d += __builtin_popcountll(a[w] ^ b[w]);                 /* :21 */
            /* insertion keeps ascending (distance, index): a later equal distance never overtakes  :23 */
            int pos = have;
            while (pos > 0 && d < bd[pos - 1]) pos--;
            if (pos >= k) continue;
            const int last = have < k ? have : k - 1;
            for (int m = last; m > pos; m--) { bd[m] = bd[m - 1]; bi[m] = bi[m - 1]; }
            bd[pos] = d; bi[pos] = j;
            if (have < k) have++;
        }
        for (int m = 0; m < k; m++) {
            index[(size_t)q * k + m] = m < have ? bi[m] : -1;
            distance[(size_t)q * k + m] = m < have ? bd[m] : -1;
        }
    }
    return 0;
}

/* returns the number of matches kept; m0/m1: up to n0 x 2 */
int oracle_select_matches(const double* pts0, const double* pts1, const int32_t* index, const int32_t* distance, int n0,
                          double* m0, double* m1) {
    int kept = 0;
    for (int q = 0; q < n0; q++) {
        if ((double)distance[2 * q] < 0.5 * (double)distance[2 * q + 1]) {                      /* :67 */
            const int32_t j = index[2 * q];
            m0[2 * kept] = pts0[2 * q]; m0[2 * kept + 1] = pts0[2 * q + 1];
            m1[2 * kept] = pts1[2 * j]; m1[2 * kept + 1] = pts1[2 * j + 1];
            kept++;
        }
    }
    return kept;
}
