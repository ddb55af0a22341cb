// RANSAC kernels for gfx950: sample + eight-point solve, residual scoring, selection.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "linalg.h"

namespace slamk {

// ------------------------------------------------------------------------------------------------
// Samplers.  Both produce k distinct indices in [0, N) per iteration, in draw order.
// ------------------------------------------------------------------------------------------------

// CPython's  random.seed(s); random.sample(range(N), k)  (reference: src/epipolar.py:70-71) replayed from the
// first SLAM_RAW_WORDS tempered MT19937 words of that seed.  getrandbits(b) is word >> (32 - b); _randbelow
// rejects values >= n; sample() swaps out of a pool when N <= setsize and rejects repeats otherwise.
struct RawStream {
    const uint32_t* w;
    int pos;
    int limit;
    bool overflow;
    __device__ uint32_t below(uint32_t n) {
        const int bits = 32 - __clz(n);
        uint32_t r = n;  // invalid until a word is accepted
        while (pos < limit) {
            r = w[pos++] >> (32 - bits);
            if (r < n) return r;
        }
        overflow = true;
        return 0;
    }
};

#define SLAM_MAX_K 16
#define SLAM_POOL_MAX 85  // setsize of sample() for 6 <= k <= 21: 21 + 4^ceil(log4(3k)) = 85

__device__ __forceinline__ int sample_setsize(int k) {
    int s = 21;
    if (k > 5) {
        int p = 1;
        while (p < 3 * k) p *= 4;
        s += p;
    }
    return s;
}

// pool: LDS bytes [SLAM_POOL_MAX][blockDim.x] (lane-minor), only touched when N <= setsize
__device__ __forceinline__ bool sample_cpython(const uint32_t* raw_row, int raw_words, int N, int k, int32_t* out,
                                               uint8_t* pool, int lane, int lanes) {
    RawStream rs{raw_row, 0, raw_words, false};
    if (N <= sample_setsize(k)) {
        for (int i = 0; i < N; ++i) pool[i * lanes + lane] = (uint8_t)i;
        for (int i = 0; i < k; ++i) {
            const uint32_t j = rs.below((uint32_t)(N - i));
            out[i] = pool[j * lanes + lane];
            pool[j * lanes + lane] = pool[(N - i - 1) * lanes + lane];
        }
    } else {
        for (int i = 0; i < k; ++i) {
            uint32_t j;
            bool dup;
            do {
                j = rs.below((uint32_t)N);
                dup = false;
                for (int q = 0; q < i; ++q) dup |= (out[q] == (int32_t)j);
            } while (dup && !rs.overflow);
            out[i] = (int32_t)j;
        }
    }
    return !rs.overflow;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so any device can draw any iteration's sample.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// word stream of iteration `it`: block b = philox(counter = (b, 0, it_lo, it_hi), key = seed); index = word >>
// (32 - bitlen(N)), rejected when >= N or already drawn.
__device__ __forceinline__ bool sample_counter(uint64_t seed, uint64_t it, int N, int k, int32_t* out) {
    const int bits = 32 - __clz((uint32_t)N);
    uint32_t w[4];
    int have = 0, i = 0;
    for (uint32_t blk = 0; blk < 4096u && i < k; ++blk) {
        philox4x32_10(blk, 0u, (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
        have = 4;
        for (int u = 0; u < have && i < k; ++u) {
            const uint32_t j = w[u] >> (32 - bits);
            bool bad = j >= (uint32_t)N;
            for (int q = 0; q < i; ++q) bad |= (out[q] == (int32_t)j);
            if (!bad) out[i++] = (int32_t)j;
        }
    }
    return i == k;
}

// standalone sampling kernel (slam_sample_indices): one lane per iteration
__global__ __launch_bounds__(64) void k_sample(int kind, uint64_t seed, int64_t it0, int H, int N, int k,
                                               const uint32_t* __restrict__ raw, int64_t raw_it0,
                                               int32_t* __restrict__ idx, int32_t* __restrict__ flags) {
    __shared__ uint8_t pool[SLAM_POOL_MAX * 64];
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    int32_t out[SLAM_MAX_K];
    bool ok;
    if (kind == SLAM_SAMPLER_CPYTHON)
        ok = sample_cpython(raw + (size_t)(it0 + h - raw_it0) * SLAM_RAW_WORDS, SLAM_RAW_WORDS, N, k, out, pool,
                            threadIdx.x, 64);
    else
        ok = sample_counter(seed, (uint64_t)(it0 + h), N, k, out);
    if (!ok) atomicOr(&flags[0], 1);
    for (int i = 0; i < k; ++i) idx[(size_t)h * k + i] = out[i];
}

// ------------------------------------------------------------------------------------------------
// K1: eight-point solve, one lane per hypothesis   (reference: src/epipolar.py:37-51)
//   rows c = [b0 a0, b0 a1, b0, b1 a0, b1 a1, b1, a0, a1, 1]; f = null vector; f/||f||; rank-2 projection.
// SAMPLER: -1 = indices given, otherwise SLAM_SAMPLER_* drawn in-kernel (and stored to idx_out when non-null).
// ------------------------------------------------------------------------------------------------
template <int SAMPLER>
__global__ __launch_bounds__(64) void k_solve(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                              const int32_t* __restrict__ idx_in, int32_t* __restrict__ idx_out,
                                              uint64_t seed, int64_t it0, int H, const uint32_t* __restrict__ raw,
                                              int64_t raw_it0, double* __restrict__ Fout,
                                              int32_t* __restrict__ flags) {
    __shared__ uint8_t pool[SAMPLER == SLAM_SAMPLER_CPYTHON ? SLAM_POOL_MAX * 64 : 1];
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;

    int32_t sel[8];
    if (SAMPLER < 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sel[i] = idx_in[(size_t)h * 8 + i];
    } else {
        bool ok;
        if (SAMPLER == SLAM_SAMPLER_CPYTHON)
            ok = sample_cpython(raw + (size_t)(it0 + h - raw_it0) * SLAM_RAW_WORDS, SLAM_RAW_WORDS, N, 8, sel, pool,
                                threadIdx.x, 64);
        else
            ok = sample_counter(seed, (uint64_t)(it0 + h), N, 8, sel);
        if (!ok) atomicOr(&flags[0], 1);
        if (idx_out) {
#pragma unroll
            for (int i = 0; i < 8; ++i) idx_out[(size_t)h * 8 + i] = sel[i];
        }
    }

    double A[9][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = min(max(sel[c], 0), N - 1);  // never read outside the arrays, whatever idx holds
        const double2 a = p0[i], b = p1[i];
        A[0][c] = b.x * a.x; A[1][c] = b.x * a.y; A[2][c] = b.x;
        A[3][c] = b.y * a.x; A[4][c] = b.y * a.y; A[5][c] = b.y;
        A[6][c] = a.x;       A[7][c] = a.y;       A[8][c] = 1.0;
    }
    double f[9], F[9];
    null_vector_9x8(A, f);
    rank2_project(f, F);
#pragma unroll
    for (int i = 0; i < 9; ++i) Fout[(size_t)h * 9 + i] = F[i];
}

// ------------------------------------------------------------------------------------------------
// K2: scoring   (reference: src/epipolar.py:73-77)
//   r = (x1^T F) x0 with x = (u, v, 1); err = r^2; inlier iff err < thr (strict); count and sum of inlier err.
// One workgroup scores HB hypotheses against ALL N matches: each lane keeps PPL matches of a chunk in registers
// and loops over the hypotheses, whose F arrive through uniform (scalar) loads.  Per-hypothesis sums are combined
// in a fixed order (lane partials over chunks, xor-butterfly over the 64 lanes, waves in order), so a hypothesis
// gets the same errsum whichever workgroup, launch or GPU scores it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int HB, int PPL>
__global__ __launch_bounds__(256) void k_score(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                               const double* __restrict__ Fall, int H, double thr,
                                               int32_t* __restrict__ counts, double* __restrict__ errsums) {
    __shared__ double s_err[HB][4];
    __shared__ int s_cnt[HB][4];
    const int tid = threadIdx.x;
    const int h0 = blockIdx.x * HB;

    double es[HB];
    int cnt[HB];  // wave-uniform: ballot popcounts
#pragma unroll
    for (int j = 0; j < HB; ++j) { es[j] = 0.0; cnt[j] = 0; }

    const double nan = __builtin_nan("");
    for (int base = 0; base < N; base += 256 * PPL) {
        double a0[PPL], a1[PPL], b0[PPL], b1[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int i = base + k * 256 + tid;
            const bool ok = i < N;
            const double2 a = ok ? p0[i] : make_double2(nan, nan);  // NaN residual: never an inlier
            const double2 b = ok ? p1[i] : make_double2(nan, nan);
            a0[k] = a.x; a1[k] = a.y; b0[k] = b.x; b1[k] = b.y;
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const int h = min(h0 + j, H - 1);  // uniform; surplus slots rescore the last hypothesis and are dropped
            const double* __restrict__ F = Fall + (size_t)h * 9;
            const double f00 = F[0], f01 = F[1], f02 = F[2], f10 = F[3], f11 = F[4], f12 = F[5], f20 = F[6],
                         f21 = F[7], f22 = F[8];
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const double l0 = fma(b0[k], f00, fma(b1[k], f10, f20));
                const double l1 = fma(b0[k], f01, fma(b1[k], f11, f21));
                const double l2 = fma(b0[k], f02, fma(b1[k], f12, f22));
                const double r = fma(l0, a0[k], fma(l1, a1[k], l2));
                const double err = r * r;
                const bool in = err < thr;
                cnt[j] += __popcll(__ballot(in));
                es[j] += in ? err : 0.0;
            }
        }
    }

    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < HB; ++j) {
        const double s = wave_sum_f64(es[j]);
        if (lane == 0) { s_err[j][wave] = s; s_cnt[j][wave] = cnt[j]; }
    }
    __syncthreads();
    if (tid < HB && h0 + tid < H) {
        counts[h0 + tid] = s_cnt[tid][0] + s_cnt[tid][1] + s_cnt[tid][2] + s_cnt[tid][3];
        errsums[h0 + tid] = ((s_err[tid][0] + s_err[tid][1]) + s_err[tid][2]) + s_err[tid][3];
    }
}

// ------------------------------------------------------------------------------------------------
// K3: selection   (reference: src/epipolar.py:78-81 applied in iteration order)
//   a later hypothesis replaces the incumbent iff it has more inliers, or as many and a strictly smaller errsum;
//   hence the winner is the lexicographic best of (count desc, errsum asc, index asc).
// Single workgroup; also emits the winner's record and inlier mask.
// ------------------------------------------------------------------------------------------------
struct Best {
    int count;
    int index;
    double errsum;
};

__device__ __forceinline__ bool better(const Best& x, const Best& y) {  // x beats y
    if (x.count != y.count) return x.count > y.count;
    if (x.errsum != y.errsum) return x.errsum < y.errsum;
    return x.index < y.index;
}

__global__ __launch_bounds__(1024) void k_select(const int32_t* __restrict__ counts, const double* __restrict__ errsums,
                                                 const double* __restrict__ Fall, int H, int64_t it0,
                                                 const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                 double thr, double* __restrict__ record, uint8_t* __restrict__ mask) {
    __shared__ Best s_best[1024];
    const int tid = threadIdx.x;
    Best b{-1, 0x7fffffff, 0.0};
    for (int h = tid; h < H; h += 1024) {
        const Best c{counts[h], h, errsums[h]};
        if (better(c, b)) b = c;
    }
    s_best[tid] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const Best o = s_best[tid + s];
            if (better(o, s_best[tid])) s_best[tid] = o;
        }
        __syncthreads();
    }
    const Best w = s_best[0];
    const double* F = Fall + (size_t)w.index * 9;
    if (tid == 0) {
        record[SLAM_REC_COUNT] = (double)w.count;
        record[SLAM_REC_ERRSUM] = w.errsum;
        record[SLAM_REC_INDEX] = (double)(it0 + w.index);
    }
    if (tid < 9) record[SLAM_REC_F + tid] = F[tid];
    if (mask) {
        const double f00 = F[0], f01 = F[1], f02 = F[2], f10 = F[3], f11 = F[4], f12 = F[5], f20 = F[6], f21 = F[7],
                     f22 = F[8];
        for (int i = tid; i < N; i += 1024) {
            const double2 a = p0[i], bb = p1[i];
            const double l0 = fma(bb.x, f00, fma(bb.y, f10, f20));
            const double l1 = fma(bb.x, f01, fma(bb.y, f11, f21));
            const double l2 = fma(bb.x, f02, fma(bb.y, f12, f22));
            const double r = fma(l0, a.x, fma(l1, a.y, l2));
            mask[i] = (r * r < thr) ? 1 : 0;
        }
    }
}

}  // namespace slamk
