// Inlier COUNTS of many hypotheses in float32 with a float64 re-check band, and the float64 error sums of the few
// hypotheses that can still win (large H, caller does not ask for every hypothesis' error sum).   gfx950.
//
// The reference decides  fl64(r*r) < 0.005  with r = (x1^T F) x0 in float64 (src/epipolar.py:73-75); the float64 kernels
// evaluate that as |r| < rcut.  Here r is first evaluated in float32 (packed: two matches per instruction), which
// differs from the float64 value by at most  delta = SLAM_F32_BAND * sum|F_ij| * max|a| * max|b|  (derivation below), so
//     |r32| <  rcut - delta   =>  inlier for certain,
//     |r32| >= rcut + delta   =>  outlier for certain,
// and only a lane in between is evaluated again in float64, with the expression and the operand order of the float64
// kernels.  The counts are therefore exactly theirs.  The error sum of a hypothesis is only needed where it decides the
// selection (hypotheses that tie on the largest count) and for the winner's record: those few are scored again in
// float64 in the canonical summation order (kernels_ransac.h), so the record has the same bits as on the float64 path.
//
// Error bound.  R = sum_ij f_ij bx_i ax_j over the reals, bx = (b0, b1, 1), ax = (a0, a1, 1).  The float32 evaluation
// rounds the nine f and the four coordinates (relative error u = 2^-24 each) and runs the four-deep fma chain of
// `residual`: every term of the sum picks up at most 7 factors (1 + e), |e| <= u (the worst: b1 f1j a1, two roundings
// of its inputs' product chain, three in the chain), so |r32 - R| <= 7u / (1 - 7u) * S = 4.2e-7 S with
// S = sum |f_ij| |bx_i| |ax_j| <= (sum |f_ij|) max(|b0|,|b1|,1) max(|a0|,|a1|,1); the float64 value is within 5e-16 S of
// R.  SLAM_F32_BAND = 5e-7 leaves a fifth of margin, and the three factors are themselves rounded upwards.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_ransac.h"

#ifndef SLAM_F32_BAND
#define SLAM_F32_BAND 5.0e-7
#endif
#define SLAM_F32_ROW 12   // floats per hypothesis in the float32 table: F (9), sum |F_ij| rounded up, two unused
namespace slamk {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Packed fma whose second and third operands are ONE half of a register pair, used for both results (a wave-uniform
// entry of F against two matches): v_pk_fma_f32 with op_sel / op_sel_hi naming the half.  Written as asm because the
// compiler, given a splat, spends a register pair per entry (72 registers for the F of four hypotheses) where two entries
// can share a pair.  B, C: 0 = low half, 1 = high half of the pair.
template <int B, int C>
__device__ __forceinline__ f32x2 pk_fma_bc(f32x2 a, f32x2 bpair, f32x2 cpair) {
    f32x2 d;
    if (B == 0 && C == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(bpair), "v"(cpair));
    if (B == 0 && C == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(bpair), "v"(cpair));
    if (B == 1 && C == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(bpair), "v"(cpair));
    if (B == 1 && C == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(bpair), "v"(cpair));
    return d;
}
// the same with a full (per-match) third operand
template <int B>
__device__ __forceinline__ f32x2 pk_fma_b(f32x2 a, f32x2 bpair, f32x2 c) {
    f32x2 d;
    if (B == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(bpair), "v"(c));
    if (B == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(bpair), "v"(c));
    return d;
}

// float32 row of one hypothesis from its float64 F: the nine entries rounded to nearest, then sum |F_ij| rounded up
__device__ __forceinline__ void write_f32_row(const double (&F)[9], float* __restrict__ row) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { row[i] = (float)F[i]; s += fabs(F[i]); }
    row[9] = (float)(s * (1.0 + 1e-6));   // (the conversion rounds to nearest: at most 6e-8 below, the factor covers it)
    row[10] = 0.f;
    row[11] = 0.f;
}

// k_solve with the float32 table as a second output (same F bits as k_solve)
template <int SAMPLER>
__global__ __launch_bounds__(64) void k_solve_f32(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                  const int32_t* __restrict__ idx_in, uint64_t seed, int64_t it0, int H,
                                                  const uint32_t* __restrict__ raw, int64_t raw_it0,
                                                  double* __restrict__ Fout, float* __restrict__ F32out,
                                                  int32_t* __restrict__ flags) {
    __shared__ uint32_t s_words[SAMPLER == SLAM_SAMPLER_CPYTHON ? SLAM_PREFETCH_WORDS * 64 : 1];
    __shared__ uint8_t s_pool[SAMPLER == SLAM_SAMPLER_CPYTHON ? SLAM_POOL_MAX * 64 : 1];
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    int32_t sel[8];
    const bool ok = indices_for<SAMPLER>(h, N, idx_in, seed, it0, raw, raw_it0, s_words, s_pool, threadIdx.x, 64, sel);
    if (!ok) raise_flag(flags);
    double F[9];
    solve_from_indices(p0, p1, N, sel, F);
#pragma unroll
    for (int i = 0; i < 9; ++i) Fout[(size_t)h * 9 + i] = F[i];
    write_f32_row(F, F32out + (size_t)h * SLAM_F32_ROW);
}

// max(|a0|,|a1|,1) * max(|b0|,|b1|,1) of one match as a float32 rounded upwards; +infinity when a coordinate is not
// finite or the product does not fit (the band then covers everything and the chunk is counted in float64)
__device__ __forceinline__ float coordinate_scale(const double2 a, const double2 b) {
    const double m = fmax(fmax(fabs(a.x), fabs(a.y)), 1.0) * fmax(fmax(fabs(b.x), fabs(b.y)), 1.0);
    const bool finite = (fabs(a.x) + fabs(a.y) + fabs(b.x) + fabs(b.y)) < 1e150;  // (false for NaN)
    return finite ? (float)(m * (1.0 + 1e-6)) : __builtin_inff();
}

// Device words shared by the kernels of one call (zeroed by the launcher before the count kernel)
struct CountState {
    unsigned int max_count;   // largest inlier count over the hypotheses
    unsigned int n_ties;      // hypotheses that have it
    unsigned int band_evals;  // (statistics) wave-level evaluations that had a lane inside the re-check band
    unsigned int pad;
};

// Counting kernel: the layout of k_score_lds (one chunk of 512 * PPL matches per workgroup in LDS, every wave scores
// groups of HG hypotheses of its own against all of it), in float32, two turns of a slice per packed instruction.
// LDS image: [slice w][turn pair q][image 0 / 1][lane] float4 = {x(turn 2q), x(turn 2q+1), y(turn 2q), y(turn 2q+1)};
// matches beyond N are staged as NaN, which fails every comparison (no lane masks in the loop).
// counts: accumulate != 0 adds this chunk's count with an atomic (several chunks; zeroed by the launcher), else stores.
template <int HG, int PPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_count_f32(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                          const float* __restrict__ F32, const double* __restrict__ Fall,
                                                          int H, double rcut, int nchunks, int nslots,
                                                          int32_t* __restrict__ counts, int accumulate,
                                                          CountState* __restrict__ state) {
    static_assert(PPL % 2 == 0, "turns are taken two at a time");
    constexpr int CH = 64 * SLAM_SLICES * PPL, THREADS = 64 * WAVES, PAIRS = PPL / 2;
    __shared__ float4 s_m[SLAM_SLICES][PAIRS][2][64];
    __shared__ unsigned int s_top;  // largest count this workgroup has stored
    __shared__ unsigned int s_band; // (statistics) its evaluations with a lane inside the band
    __shared__ unsigned int s_M;   // bits of max over the chunk of max(|a0|,|a1|,1) * max(|b0|,|b1|,1), rounded up
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = (H + HG - 1) / HG;
    unsigned int band_seen = 0;
    int top = 0;  // largest final count this wave has stored (wave-uniform)
    if (tid == 0) { s_top = 0u; s_band = 0u; }
    int vzero;  // a zero the compiler cannot see through: keeps the (wave-uniform) loads of F on the vector side.  As scalar
                // operands the 4 x 11 values and the 16 lane masks of a step do not fit the scalar registers, and the
                // compiler parks scalars in vector lanes (a v_readlane and its wait states per use)
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    SLAM_STAMP_WAVE(0, clock64());
    SLAM_STAMP_WAVE(6, wall_clock64());
    int stamp_group = 0;
    (void)stamp_group;
    for (int u = blockIdx.x; u < nchunks * nslots; u += gridDim.x) {
        const int chunk = __builtin_amdgcn_readfirstlane(u / nslots);
        const int slot = u - chunk * nslots;
        const int base = chunk * CH;
        if (u != (int)blockIdx.x) __syncthreads();  // everybody has finished with the previous chunk
        if (tid == 0) s_M = 0u;
        __syncthreads();
        float mmax = 0.f;
        for (int e = tid; e < CH / 2; e += THREADS) {  // e = q * 512 + w * 64 + l: the two turns 2q and 2q+1 of (w, l)
            const int q = e / (64 * SLAM_SLICES), w = (e / 64) % SLAM_SLICES, l = e % 64;
            const int i0 = base + (2 * q) * 64 * SLAM_SLICES + w * 64 + l, i1 = i0 + 64 * SLAM_SLICES;
            const float nanf_ = __builtin_nanf("");
            float4 A = make_float4(nanf_, nanf_, nanf_, nanf_), B = A;
            if (i0 < N) {
                const double2 a = p0[i0], b = p1[i0];
                A.x = (float)a.x; A.z = (float)a.y; B.x = (float)b.x; B.z = (float)b.y;
                mmax = fmaxf(mmax, coordinate_scale(a, b));
            }
            if (i1 < N) {
                const double2 a = p0[i1], b = p1[i1];
                A.y = (float)a.x; A.w = (float)a.y; B.y = (float)b.x; B.w = (float)b.y;
                mmax = fmaxf(mmax, coordinate_scale(a, b));
            }
            s_m[w][q][0][l] = A;
            s_m[w][q][1][l] = B;
        }
        atomicMax(&s_M, __float_as_uint(mmax));  // non-negative floats order like their bits (an infinity or NaN product
                                                 // makes the band cover everything: every lane is then re-checked)
        __syncthreads();
        const float Mf = __uint_as_float(s_M);
        SLAM_STAMP_WAVE(1, clock64());
        for (int g = slot * WAVES + wave; g < ngroups; g += nslots * WAVES) {
            const int h0 = g * HG;
            // Per hypothesis: w2 = half-width of the band in the SQUARED domain.  With d = fl32(r32^2 - rc2), rc2 = fl32(rcut^2)
            // (one fma, one rounding; |rc2 - rcut^2| <= 3e-10):  d >= w2 excludes |r64| < rcut and d <= -w2 excludes
            // |r64| >= rcut as soon as  w2 (1 - 2^-24) >= 2 rcut delta + delta^2 + 3e-10  (|r32| is within delta of |r64|),
            // so a lane is certain when |d| >= w2, and then the SIGN of d is the answer.
            float w2[HG];
            f32x2 fp[HG][5];  // F as pairs (f0,f1) (f2,f3) (f4,f5) (f6,f7) (f8,-)
            int cl[HG];       // per-lane inlier counts of this lane's matches (summed over the wave at the end)
#pragma unroll
            for (int h = 0; h < HG; ++h) {
                const float* Fh = F32 + (size_t)min(h0 + h, H - 1) * SLAM_F32_ROW + vzero;  // surplus slots rescore the last hypothesis
#pragma unroll
                for (int i = 0; i < 5; ++i) fp[h][i] = (f32x2){Fh[2 * i], Fh[2 * i + 1]};  // (entry 9 is not an F entry; unused as such)
                const double delta = SLAM_F32_BAND * (double)Fh[9] * (double)Mf;
                const double w = (2.0 * rcut * delta + delta * delta) * (1.0 + 1e-6) + 1e-9;
                w2[h] = (float)w * (1.0f + 1.2e-7f);                      // rounded upwards
                if (!(delta < rcut)) w2[h] = __builtin_inff();            // band covers everything (also for NaN)
                cl[h] = 0;
            }
            const f32x2 nrc2 = {-(float)(rcut * rcut), 0.f};
            const int left = (ngroups - 1 - g) / (nslots * WAVES);  // groups after this one
            // Sign bits of d, one per turn: each slice shifts its PPL bits in from below (turn k lands on bit PPL - 1 - k
            // of the lowest PPL), and the word is counted when it is full, every 32 / PPL slices, not per slice.
            constexpr int FLUSH_EVERY = 32 / PPL;
            constexpr unsigned int SLICE_MASK = (1u << PPL) - 1u;
            unsigned int bits[HG];
#pragma unroll
            for (int h = 0; h < HG; ++h) bits[h] = 0u;
            const bool full_chunk = base + CH <= N;  // (uniform) every turn of every lane holds a match
#pragma unroll 1
            for (int w = 0; w < SLAM_SLICES; ++w) {
                {   // wave priority by the work that is left, as in k_score_lds
                    const int rem = left * SLAM_SLICES + (SLAM_SLICES - 1 - w);
                    if (rem >= 16) __builtin_amdgcn_s_setprio(3);
                    else if (rem >= 6) __builtin_amdgcn_s_setprio(2);
                    else if (rem >= 2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                float mind[HG];         // smallest |d| of this lane's turns of the slice
#pragma unroll
                for (int h = 0; h < HG; ++h) mind[h] = __builtin_inff();
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) {
                    const float4 A = s_m[w][q][0][lane], B = s_m[w][q][1][lane];
                    const f32x2 a0 = {A.x, A.y}, a1 = {A.z, A.w}, b0 = {B.x, B.y}, b1 = {B.z, B.w};
#pragma unroll
                    for (int h = 0; h < HG; ++h) {
                        // l0 = b0 f0 + (b1 f3 + f6), l1 = b0 f1 + (b1 f4 + f7), l2 = b0 f2 + (b1 f5 + f8): the chain of `residual`
                        const f32x2 l0 = pk_fma_b<0>(b0, fp[h][0], pk_fma_bc<1, 0>(b1, fp[h][1], fp[h][3]));
                        const f32x2 l1 = pk_fma_b<1>(b0, fp[h][0], pk_fma_bc<0, 1>(b1, fp[h][2], fp[h][3]));
                        const f32x2 l2 = pk_fma_b<0>(b0, fp[h][1], pk_fma_bc<1, 0>(b1, fp[h][2], fp[h][4]));
                        const f32x2 r = pk_fma(l0, a0, pk_fma(l1, a1, l2));
                        f32x2 d;
                        asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(r), "v"(nrc2));
                        asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(mind[h]) : "v"(d.x), "v"(d.y));   // (a NaN operand is ignored)
                        bits[h] = __builtin_amdgcn_alignbit(bits[h], __float_as_uint(d.x), 31);      // bits << 1 | sign(d.x)
                        bits[h] = __builtin_amdgcn_alignbit(bits[h], __float_as_uint(d.y), 31);
                    }
                }
                if (!full_chunk) {
                    // Turns of this lane that hold no match (beyond N; staged as NaN, which min3 skips) are cleared from
                    // the slice's bits.  (Non-finite REAL coordinates make the chunk's band infinite at staging, so every
                    // slice of such a chunk is counted in float64 below.)
                    const int nl = N - base - w * 64 - lane;  // > k * 512  <=>  turn k holds a match
                    const int live_turns = max(0, min(PPL, (nl + 64 * SLAM_SLICES - 1) / (64 * SLAM_SLICES)));
                    const unsigned int live_bits = ((1u << live_turns) - 1u) << (PPL - live_turns);
#pragma unroll
                    for (int h = 0; h < HG; ++h) bits[h] &= ~SLICE_MASK | live_bits;
                }
#pragma unroll
                for (int h = 0; h < HG; ++h) {
                    if (__any(mind[h] < w2[h])) {  // (wave-uniform, rare) some lane is inside the band: the slice again in
                        ++band_seen;               // float64, as the float64 kernels evaluate it; its bits stay in the word
                        const double* Fd = Fall + (size_t)min(h0 + h, H - 1) * 9;   // and the difference is added
                        const double f0 = Fd[0], f1 = Fd[1], f2 = Fd[2], f3 = Fd[3], f4 = Fd[4], f5 = Fd[5], f6 = Fd[6], f7 = Fd[7],
                                     f8 = Fd[8];
                        int n = 0;
#pragma unroll 1
                        for (int k = 0; k < PPL; ++k) {
                            const int i = base + k * 64 * SLAM_SLICES + w * 64 + lane;
                            const double2 a = p0[min(i, N - 1)], b = p1[min(i, N - 1)];
                            const double r = residual<false>(a.x, a.y, b.x, b.y, f0, f1, f2, f3, f4, f5, f6, f7, f8);
                            n += (i < N && fabs(r) < rcut) ? 1 : 0;
                        }
                        cl[h] += n - __popc(bits[h] & SLICE_MASK);
                    }
                }
                if ((w + 1) % FLUSH_EVERY == 0 || w == SLAM_SLICES - 1) {  // (uniform)
#pragma unroll
                    for (int h = 0; h < HG; ++h) { cl[h] += __popc(bits[h]); bits[h] = 0u; }
                }
            }
            int cnt[HG];  // wave totals
#pragma unroll
            for (int h = 0; h < HG; ++h) {
                int v = cl[h];
                v += dpp_i32<0xB1>(v);   // lane ^ 1
                v += dpp_i32<0x4E>(v);   // lane ^ 2
                v += dpp_i32<0x141>(v);  // 7 - l within 8
                v += dpp_i32<0x140>(v);  // 15 - l within 16
                cnt[h] = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
                         __builtin_amdgcn_readlane(v, 48);
            }
            if (lane < HG) {
                int c = cnt[0];
#pragma unroll
                for (int h = 1; h < HG; ++h) c = lane == h ? cnt[h] : c;
                if (h0 + lane < H) {
                    if (accumulate) atomicAdd(&counts[h0 + lane], c);
                    else counts[h0 + lane] = c;
                }
            }
#pragma unroll
            for (int h = 0; h < HG; ++h)
                if (h0 + h < H) top = max(top, cnt[h]);
#ifdef SLAM_STAMPS
            if (stamp_group < 3) SLAM_STAMP_WAVE(2 + stamp_group, clock64());
            ++stamp_group;
#endif
        }
    }
    __builtin_amdgcn_s_setprio(0);
    SLAM_STAMP_WAVE(5, clock64());
    SLAM_STAMP_WAVE(7, wall_clock64());
    // One atomic per WORKGROUP on the shared word: atomics on one address complete one after the other (about 13 ns each),
    // and one per group of hypotheses made them as long as the whole kernel.
    // (the same for the band statistics)
    if (lane == 0) {
        if (!accumulate) atomicMax(&s_top, (unsigned int)top);
        if (band_seen) atomicAdd(&s_band, band_seen);
    }
    __syncthreads();
    if (tid == 0) {
        if (!accumulate) atomicMax(&state->max_count, s_top);
        if (s_band) atomicAdd(&state->band_evals, s_band);
    }
}

// several chunks: the largest count, once every chunk has been added   (one thread per hypothesis)
// (a workgroup takes 256 * 32 hypotheses and issues one atomic)
__global__ __launch_bounds__(256) void k_max_count(const int32_t* __restrict__ counts, int H, CountState* __restrict__ state) {
    __shared__ unsigned int s_top;
    if (threadIdx.x == 0) s_top = 0u;
    __syncthreads();
    int c = 0;
    for (int k = 0; k < 32; ++k) {
        const int h = (blockIdx.x * 32 + k) * 256 + threadIdx.x;
        if (h < H) c = max(c, counts[h]);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) c = max(c, __shfl_xor(c, m, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_top, (unsigned int)c);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&state->max_count, s_top);
}

// the hypotheses that have the largest count, in no particular order (the selection that follows is a total order)
// (tie_index holds H entries: every hypothesis may tie)
__global__ __launch_bounds__(256) void k_collect_ties(const int32_t* __restrict__ counts, int H, CountState* __restrict__ state,
                                                      int32_t* __restrict__ tie_index) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    if ((unsigned int)counts[h] == state->max_count) tie_index[atomicAdd(&state->n_ties, 1u)] = h;
}

// Float64 error sum of tie t over chunk c, in the canonical order (kernels_ransac.h): eight waves = the eight slices,
// a lane's matches of the chunk in turn order, the slices in order per lane position, the tree over the lane positions.
// A launch handles the ties t_base .. t_base + cap - 1 (the launcher walks t_base over the H possible ties; a launch whose
// range lies beyond the actual number ends at once): workgroup (c, y) takes ties t_base + y, + gridDim.y, ...;
// partial[c * cap + (t - t_base)] receives the chunk's sum.
template <int PPL>
__global__ __launch_bounds__(512) void k_rescore_ties(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                      const double* __restrict__ Fall, double rcut,
                                                      const CountState* __restrict__ state,
                                                      const int32_t* __restrict__ tie_index, int t_base, int cap,
                                                      double* __restrict__ partial) {
    constexpr int CH = 64 * SLAM_SLICES * PPL;
    __shared__ double s_part[SLAM_SLICES][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, base = chunk * CH;
    const int n_ties = (int)min((long long)state->n_ties - t_base, (long long)cap);  // ties of this launch
    if ((int)blockIdx.y >= n_ties) return;
    double a0[PPL], a1[PPL], b0[PPL], b1[PPL];
    bool live[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int i = base + k * 64 * SLAM_SLICES + tid;
        live[k] = i < N;
        const double2 a = p0[live[k] ? i : N - 1], b = p1[live[k] ? i : N - 1];
        a0[k] = a.x; a1[k] = a.y; b0[k] = b.x; b1[k] = b.y;
    }
    for (int t = blockIdx.y; t < n_ties; t += gridDim.y) {
        const double* F = Fall + (size_t)tie_index[t_base + t] * 9;
        const double f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3], f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7], f8 = F[8];
        double es = 0.0;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const double r = residual<false>(a0[k], a1[k], b0[k], b1[k], f0, f1, f2, f3, f4, f5, f6, f7, f8);
            if (live[k] && fabs(r) < rcut) es = fma(r, r, es);
        }
        if (t != (int)blockIdx.y) __syncthreads();  // the previous tie's partial sums have been read
        s_part[wave][lane] = es;
        __syncthreads();
        if (wave == 0) {
            double tl = s_part[0][lane];
#pragma unroll
            for (int w = 1; w < SLAM_SLICES; ++w) tl += s_part[w][lane];
            const double total = wave_sum_f64(tl);
            if (lane == 0) partial[(size_t)chunk * cap + t] = total;
        }
    }
}

// counts and error sums of the ties as dense arrays for the selection kernel: the chunks folded in order
__global__ __launch_bounds__(256) void k_fold_ties(const CountState* __restrict__ state, int t_base, int cap, int nchunks,
                                                   const double* __restrict__ partial, int32_t* __restrict__ tie_counts,
                                                   double* __restrict__ tie_errsums, int32_t* __restrict__ n_out) {
    const int n_ties = (int)min((long long)state->n_ties - t_base, (long long)cap);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0 && t_base == 0) *n_out = (int32_t)state->n_ties;
    if (t >= n_ties) return;
    double e = 0.0;
    for (int c = 0; c < nchunks; ++c) e += partial[(size_t)c * cap + t];
    tie_counts[t_base + t] = (int32_t)state->max_count;
    tie_errsums[t_base + t] = e;
}

// the selection kernel over the ties: entry t is hypothesis tie_index[t]; their number comes from device memory
__global__ __launch_bounds__(SLAM_SELECT_THREADS) void k_select_ties(const int32_t* __restrict__ tie_counts,
                                                                     const double* __restrict__ tie_errsums,
                                                                     const int32_t* __restrict__ tie_index,
                                                                     const int32_t* __restrict__ n_ties,
                                                                     const double* __restrict__ Fall, int64_t it0,
                                                                     const double2* __restrict__ p0, const double2* __restrict__ p1,
                                                                     int N, double rcut, double* __restrict__ record,
                                                                     uint8_t* __restrict__ mask) {
    __shared__ Best s_best[SLAM_SELECT_THREADS / 64];
    select_body<SLAM_SELECT_THREADS, false>(tie_counts, tie_errsums, tie_index, Fall, *n_ties, it0, p0, p1, N, rcut, record, mask,
                                            (int)blockIdx.x, (int)gridDim.x, s_best);
}

}  // namespace slamk
