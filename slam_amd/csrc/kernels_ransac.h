// RANSAC kernels for gfx950: sample + eight-point solve, residual scoring, selection.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctx.h"          // slam_carried_select, SLAM_RAW_WORDS
#include "kernels_tri.h"  // PairArgs
#include "linalg.h"

// waves per CU the fused kernel is compiled for when a lane keeps at most four matches (N <= 2048): 40 = five per
// SIMD = 96 registers.  48 (80 registers) spills 76 bytes per lane inside the solve and makes the kernel 20 % slower
// (11.8 against 9.5 us at N=2048, H=1000), also for a batch of pairs that fills the chip.
#ifndef SLAM_SMALL_SHAPE_WAVES
#define SLAM_SMALL_SHAPE_WAVES 40
#endif
namespace slamk {

#ifdef SLAM_STAMPS
// diagnostic build only: shader-clock stamps of workgroup phases, read back by slam_debug_stamps()
__device__ long long g_stamps[4096 * 8];
#ifdef SLAM_NO_PHASE_STAMPS  // log-only diagnostic build: the workgroup log below without the phase stamps' stores
#define SLAM_STAMP(slot) do { } while (0)
#else
#define SLAM_STAMP(slot)                                                                        \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (slot)] = clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#endif
// Log of workgroup lifetimes across launches (constant 100 MHz clock, comparable between CUs and launches): while
// g_wg_log_on is set, every workgroup of the fused kernel appends {tag of its launch (bits of its F pointer and its
// first iteration), workgroup index, XCC/CU id, lifetime in shader cycles, end on the constant clock (read once, at
// the very end: that read takes about a microsecond)}; read back by slam_debug_wg_log().
#define SLAM_WG_LOG_MAX (1 << 17)
__device__ unsigned int g_wg_log_on, g_wg_log_n;
__device__ long long g_wg_log[SLAM_WG_LOG_MAX * 4];
#define SLAM_WG_LOG_BEGIN() const long long wg_log_t0_ = clock64()  /* shader clock: cheap to read */
#define SLAM_WG_LOG_END(tag)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && g_wg_log_on) {                                                        \
            const unsigned int slot_ = atomicAdd(&g_wg_log_n, 1u);                                    \
            if (slot_ < SLAM_WG_LOG_MAX) {                                                            \
                unsigned int hw_;                                                                     \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                     \
                unsigned int xcc_;                                                                    \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                   \
                g_wg_log[4 * slot_ + 0] = (long long)(tag);                                           \
                g_wg_log[4 * slot_ + 1] = ((long long)blockIdx.x << 32) | ((xcc_ & 0xf) << 16) | (hw_ & 0xffff); \
                g_wg_log[4 * slot_ + 2] = clock64() - wg_log_t0_; /* lifetime in shader cycles */      \
                g_wg_log[4 * slot_ + 3] = wall_clock64();         /* end, constant 100 MHz clock */   \
            }                                                                                         \
        }                                                                                             \
    } while (0)
// with SLAM_WALL_STAMPS: start and end of every workgroup of the fused kernel on the constant 100 MHz clock (the same
// for all compute dies, unlike the shader clock's counters): row 3072 + workgroup, slots 0 and 1.  Each read stalls the
// wave for about a microsecond, so this build answers "when do workgroups start and end", not "how long is a phase".
#ifdef SLAM_WALL_STAMPS
#define SLAM_WALL_STAMP(slot)                                                                                  \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_stamps[(3072 + (int)blockIdx.x) * 8 + (slot)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#else
#define SLAM_WALL_STAMP(slot) do { } while (0)
#endif
// the selection kernel's stamps go to the last row
#ifdef SLAM_NO_PHASE_STAMPS
#define SLAM_STAMP_SELECT(slot) do { } while (0)
#else
#define SLAM_STAMP_SELECT(slot)                                                       \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[4095 * 8 + (slot)] = clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#endif
// per-wave stamps of the large-H scoring kernel: row = workgroup * 16 + wave (rows beyond 4095 are dropped)
#define SLAM_STAMP_WAVE(slot, value)                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        const int row_ = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 6);                              \
        if ((threadIdx.x & 63) == 0 && row_ < 4096) g_stamps[row_ * 8 + (slot)] = (value);            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
// per-wave stamps of the fused kernel's score phase: row = 1024 + workgroup * 8 + wave, workgroups 0..255
#define SLAM_STAMP_W8(slot)                                                                           \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)                                              \
            g_stamps[(1024 + (int)blockIdx.x * 8 + (int)(threadIdx.x >> 6)) * 8 + (slot)] = clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define SLAM_STAMP_W8(slot) do { } while (0)
#define SLAM_WALL_STAMP(slot) do { } while (0)
#define SLAM_STAMP_WAVE(slot, value) do { } while (0)
#define SLAM_STAMP(slot) do { } while (0)
#define SLAM_STAMP_SELECT(slot) do { } while (0)
#define SLAM_WG_LOG_BEGIN() do { } while (0)
#define SLAM_WG_LOG_END(tag) do { } while (0)
#endif


// ------------------------------------------------------------------------------------------------
// Samplers.  Both produce k distinct indices in [0, N) per iteration, in draw order.
// ------------------------------------------------------------------------------------------------

// CPython's  random.seed(s); random.sample(range(N), k)  (reference: src/epipolar.py:70-71) replayed from the
// first SLAM_RAW_WORDS tempered MT19937 words of that seed.  getrandbits(b) is word >> (32 - b); _randbelow
// rejects values >= n; sample() swaps out of a pool when N <= setsize and rejects repeats otherwise.
// The first SLAM_PREFETCH_WORDS words of a lane's row are staged in LDS with one round of wide loads, so the
// data-dependent consumption loop does not pay a global-memory latency per word.
#define SLAM_PREFETCH_WORDS 32
#define SLAM_MAX_K 16
#define SLAM_POOL_MAX 85  // setsize of sample() for 6 <= k <= 21: 21 + 4^ceil(log4(3k)) = 85

struct RawStream {
    const uint32_t* row;  // this lane's row in the global table
    const uint32_t* lds;  // this lane's column in LDS: word j at lds[j * stride]
    int stride;
    int pos;
    bool overflow;
    __device__ __forceinline__ uint32_t below(uint32_t n) {
        const int shift = __clz(n);  // 32 - bit_length(n)
        while (pos < SLAM_RAW_WORDS) {
            const uint32_t w = pos < SLAM_PREFETCH_WORDS ? lds[pos * stride] : row[pos];
            ++pos;
            const uint32_t r = w >> shift;
            if (r < n) return r;
        }
        overflow = true;
        return 0;
    }
};

__device__ __forceinline__ int sample_setsize(int k) {
    int s = 21;
    if (k > 5) {
        int p = 1;
        while (p < 3 * k) p *= 4;
        s += p;
    }
    return s;
}

// stage the first SLAM_PREFETCH_WORDS words of `row` into this lane's LDS column (8 x 16-byte loads in flight)
__device__ __forceinline__ void stage_words(const uint32_t* row, uint32_t* lds, int stride) {
    const uint4* src = (const uint4*)row;  // rows are 384 B apart and the table is 256-B aligned
    uint4 v[SLAM_PREFETCH_WORDS / 4];
#pragma unroll
    for (int q = 0; q < SLAM_PREFETCH_WORDS / 4; ++q) v[q] = src[q];
#pragma unroll
    for (int q = 0; q < SLAM_PREFETCH_WORDS / 4; ++q) {
        lds[(4 * q + 0) * stride] = v[q].x;
        lds[(4 * q + 1) * stride] = v[q].y;
        lds[(4 * q + 2) * stride] = v[q].z;
        lds[(4 * q + 3) * stride] = v[q].w;
    }
}

// words: LDS [SLAM_PREFETCH_WORDS][lanes]; pool: LDS bytes [SLAM_POOL_MAX][lanes] (lane-minor), touched only
// when N <= setsize.  `lane` is this lane's column, `lanes` the number of columns.
__device__ __forceinline__ bool sample_cpython(const uint32_t* raw_row, int N, int k, int32_t* out, uint32_t* words,
                                               uint8_t* pool, int lane, int lanes) {
    stage_words(raw_row, words + lane, lanes);
    RawStream rs{raw_row, words + lane, lanes, 0, false};
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
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0;
        const uint32_t n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// word stream of iteration `it`: block b = philox(counter = (b, 0, it_lo, it_hi), key = seed); index = word >>
// (32 - bitlen(N)), rejected when >= N or already drawn.  k <= 8 indices go to out[0..7] (static indexing only).
__device__ __forceinline__ bool sample_counter8(uint64_t seed, uint64_t it, int N, int k, int32_t (&out)[8]) {
    const int shift = __clz((uint32_t)N);
    int i = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) out[q] = -1;
    for (uint32_t blk = 0; blk < 4096u && i < k; ++blk) {
        uint32_t w[4];
        philox4x32_10(blk, 0u, (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t j = (int32_t)(w[u] >> shift);
            bool ok = (j < N) && (i < k);
#pragma unroll
            for (int q = 0; q < 8; ++q) ok &= (out[q] != j);  // unused slots hold -1
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] = (ok && q == i) ? j : out[q];
            i += ok ? 1 : 0;
        }
    }
    return i == k;
}

// ------------------------------------------------------------------------------------------------
// Wave-parallel draw of ONE iteration's 8 indices: lane l looks at word l of the iteration's stream.
// Sequential semantics being reproduced (set path of random.sample, and the counter sampler): walk the words in
// order, drop a word whose value is >= N or equals an already accepted value, stop at 8 accepted values.
// A word is dropped as a repeat exactly when an EARLIER in-range word has the same value, so acceptance can be
// decided for all lanes at once.  Only the first 16 in-range words are considered; if they hold fewer than 8
// distinct values (or the 64 words run out) the caller falls back to the sequential walk.
// All 64 lanes must be active and pass the same arguments apart from `word`.  The indices go to LDS (out[0..7]).
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

// scratch: 16 LDS words private to the calling wave; out: 8 LDS words that receive the indices
__device__ __forceinline__ bool select8_from_words(uint32_t word, int N, int32_t* scratch, int32_t* out) {
    const int lane = threadIdx.x & 63;
    const int32_t r = (int32_t)(word >> __clz((uint32_t)N));
    const bool valid = r < N;
    const unsigned long long vmask = __ballot(valid);
    const int my_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vmask, 0));
    const int nvalid = min((int)__popcll(vmask), 16);
    // compact the first 16 in-range values into lanes 0..15 (through LDS; same-wave LDS traffic is ordered)
    if (valid && my_rank < 16) scratch[my_rank] = r;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int32_t c = scratch[lane & 15];
    // lane q < 16 repeats an earlier value iff c_q == c_{q-k} for some 1 <= k <= q: rotate the row of 16 by k
    bool repeat = false;
    const int q = lane & 15;
    // (the rotation runs with every lane active: a DPP read from a disabled lane returns nothing)
#define SLAM_ROT(k) { const int o_ = dpp_i32<0x120 + (k)>(c); repeat |= ((k) <= q) & (o_ == c); }  // row_ror:k: lane q-k
    SLAM_ROT(1) SLAM_ROT(2) SLAM_ROT(3) SLAM_ROT(4) SLAM_ROT(5) SLAM_ROT(6) SLAM_ROT(7) SLAM_ROT(8)
    SLAM_ROT(9) SLAM_ROT(10) SLAM_ROT(11) SLAM_ROT(12) SLAM_ROT(13) SLAM_ROT(14) SLAM_ROT(15)
#undef SLAM_ROT
    const bool accepted = (lane < nvalid) && !repeat;
    const unsigned long long amask = __ballot(accepted);
    const int arank = __builtin_amdgcn_mbcnt_lo((unsigned)amask, 0);  // accepted lanes are all below 16
    if (accepted && arank < 8) out[arank] = c;
    return __popcll(amask) >= 8;
}

// sequential walk over a lane-private copy of the stream (used when the wave-parallel draw does not apply)
__device__ __forceinline__ bool sample_cpython_seq(const uint32_t* row, int N, int k, int32_t* out, uint8_t* pool,
                                                   int lane, int lanes) {
    int pos = 0;
    bool overflow = false;
    auto below = [&](uint32_t n) -> uint32_t {
        const int shift = __clz(n);
        while (pos < SLAM_RAW_WORDS) {
            const uint32_t r = row[pos++] >> shift;
            if (r < n) return r;
        }
        overflow = true;
        return 0u;
    };
    if (N <= sample_setsize(k)) {
        for (int i = 0; i < N; ++i) pool[i * lanes + lane] = (uint8_t)i;
        for (int i = 0; i < k; ++i) {
            const uint32_t j = below((uint32_t)(N - i));
            out[i] = pool[j * lanes + lane];
            pool[j * lanes + lane] = pool[(N - i - 1) * lanes + lane];
        }
    } else {
        for (int i = 0; i < k; ++i) {
            uint32_t j;
            bool dup;
            do {
                j = below((uint32_t)N);
                dup = false;
                for (int q = 0; q < i; ++q) dup |= (out[q] == (int32_t)j);
            } while (dup && !overflow);
            out[i] = (int32_t)j;
        }
    }
    return !overflow;
}

// One iteration's 8 indices drawn by a whole wave into LDS (out[0..7]).  SAMPLER: -1 = given, else SLAM_SAMPLER_*.
// word0: this lane's word of the iteration's MT19937 row (CPython sampler only).
// pool: LDS bytes [SLAM_POOL_MAX] and scratch: 16 LDS words, both private to this wave.
template <int SAMPLER>
__device__ __forceinline__ bool indices_for_wave(int h, int N, uint32_t word0, const int32_t* __restrict__ idx_in,
                                                 uint64_t seed, int64_t it0, const uint32_t* __restrict__ raw,
                                                 int64_t raw_it0, uint8_t* pool, int32_t* scratch, int32_t* out) {
    const int lane = threadIdx.x & 63;
    if (SAMPLER < 0) {
        if (lane < 8) out[lane] = idx_in[(size_t)h * 8 + lane];
        return true;
    } else if (SAMPLER == SLAM_SAMPLER_CPYTHON) {
        const uint32_t* row = raw + (size_t)(it0 + h - raw_it0) * SLAM_RAW_WORDS;
        if (N > sample_setsize(8)) {  // set path of random.sample; word0 = row[lane], loaded by the caller
            if (select8_from_words(word0, N, scratch, out)) return true;
        }
        // pool path (N <= 85) or a draw that needs more than the first 16 in-range words: sequential, on lane 0
        bool ok = true;
        if (lane == 0) {
            int32_t sel[8];
            ok = sample_cpython_seq(row, N, 8, sel, pool, 0, 1);
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] = sel[q];
        }
        return __builtin_amdgcn_readlane((int)ok, 0) != 0;
    } else {
        uint32_t w[4];
        philox4x32_10((uint32_t)(lane >> 2), 0u, (uint32_t)(it0 + h), (uint32_t)((uint64_t)(it0 + h) >> 32),
                      (uint32_t)seed, (uint32_t)(seed >> 32), w);
        const int u = lane & 3;
        const uint32_t word = u == 0 ? w[0] : (u == 1 ? w[1] : (u == 2 ? w[2] : w[3]));
        if (select8_from_words(word, N, scratch, out)) return true;
        int32_t sel[8];
        const bool ok = sample_counter8(seed, (uint64_t)(it0 + h), N, 8, sel);  // every lane walks the same stream
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] = sel[q];
        }
        return ok;
    }
}

// standalone sampling kernel (slam_sample_indices): one lane per iteration
__global__ __launch_bounds__(64) void k_sample(int kind, uint64_t seed, int64_t it0, int H, int N, int k,
                                               const uint32_t* __restrict__ raw, int64_t raw_it0,
                                               int32_t* __restrict__ idx, int32_t* __restrict__ flags) {
    __shared__ uint8_t pool[SLAM_POOL_MAX * 64];
    __shared__ uint32_t words[SLAM_PREFETCH_WORDS * 64];
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    bool ok;
    if (kind == SLAM_SAMPLER_CPYTHON) {
        int32_t out[SLAM_MAX_K];
        ok = sample_cpython(raw + (size_t)(it0 + h - raw_it0) * SLAM_RAW_WORDS, N, k, out, words, pool, threadIdx.x, 64);
        for (int i = 0; i < k; ++i) idx[(size_t)h * k + i] = out[i];
    } else {
        int32_t out[8];
        ok = sample_counter8(seed, (uint64_t)(it0 + h), N, k, out);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < k) idx[(size_t)h * k + i] = out[i];
    }
    if (!ok) raise_flag(flags);
}

// ------------------------------------------------------------------------------------------------
// Eight-point solve of EIGHT hypotheses by one wave   (reference: src/epipolar.py:37-51)
//   rows [b0 a0, b0 a1, b0, b1 a0, b1 a1, b1, a0, a1, 1]; f = null vector; f/||f||; rank-2 projection.
// Lane 8 g + i holds correspondence i of hypothesis g (a in image 0, b in image 1) and so row i of its design matrix;
// F comes back on all eight lanes of the group.  Every kernel that solves a minimal sample goes through here, so a
// hypothesis has the same F bit for bit wherever it is solved.  All 64 lanes must be active.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void solve8_by_eight_lanes(const double2 a, const double2 b, double (&F)[9]) {
    double row[9] = {b.x * a.x, b.x * a.y, b.x, b.y * a.x, b.y * a.y, b.y, a.x, a.y, 1.0};
    double f[9];
    null_vector_8x9_by_eight_lanes(row, f);
    SLAM_STAMP(7);
    rank2_project(f, F);
}

// The same solve by one lane (throughput form: 64 hypotheses per wave); bit-identical to solve8_by_eight_lanes.
__device__ __forceinline__ void solve_from_indices(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                   const int32_t (&sel)[8], double (&F)[9]) {
    double A[8][9];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = min(max(sel[r], 0), N - 1);  // never read outside the arrays, whatever idx holds
        const double2 a = p0[i], b = p1[i];
        A[r][0] = b.x * a.x; A[r][1] = b.x * a.y; A[r][2] = b.x;
        A[r][3] = b.y * a.x; A[r][4] = b.y * a.y; A[r][5] = b.y;
        A[r][6] = a.x;       A[r][7] = a.y;       A[r][8] = 1.0;
    }
    double f[9];
    null_vector_8x9_one_lane(A, f);
    rank2_project(f, F);
}

// draw (or read) the 8 indices of iteration it0 + h.  SAMPLER: -1 = given in idx_in, else SLAM_SAMPLER_*.
// words/pool: LDS with `lanes` columns, this lane using column `lane`.
template <int SAMPLER>
__device__ __forceinline__ bool indices_for(int h, int N, const int32_t* __restrict__ idx_in, uint64_t seed, int64_t it0,
                                            const uint32_t* __restrict__ raw, int64_t raw_it0, uint32_t* words,
                                            uint8_t* pool, int lane, int lanes, int32_t (&sel)[8]) {
    if (SAMPLER < 0) {
        const int4 lo = ((const int4*)idx_in)[2 * (size_t)h], hi = ((const int4*)idx_in)[2 * (size_t)h + 1];
        sel[0] = lo.x; sel[1] = lo.y; sel[2] = lo.z; sel[3] = lo.w;
        sel[4] = hi.x; sel[5] = hi.y; sel[6] = hi.z; sel[7] = hi.w;
        return true;
    } else if (SAMPLER == SLAM_SAMPLER_CPYTHON) {
        return sample_cpython(raw + (size_t)(it0 + h - raw_it0) * SLAM_RAW_WORDS, N, 8, sel, words, pool, lane, lanes);
    } else {
        return sample_counter8(seed, (uint64_t)(it0 + h), N, 8, sel);
    }
}

// K1 standalone: one lane per hypothesis.  Used for single solves and for large H, where the number of instructions
// per hypothesis counts (the eight-lane form of the fused kernel issues five times as many) and 64 hypotheses per
// wave keep the vector units busy.
template <int SAMPLER>
__global__ __launch_bounds__(64) void k_solve(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                              const int32_t* __restrict__ idx_in, uint64_t seed, int64_t it0, int H,
                                              const uint32_t* __restrict__ raw, int64_t raw_it0,
                                              double* __restrict__ Fout, int32_t* __restrict__ flags) {
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
}

// ------------------------------------------------------------------------------------------------
// Scoring   (reference: src/epipolar.py:73-77)
//   r = (x1^T F) x0 with x = (u, v, 1); err = r^2; inlier iff err < thr (strict); count and sum of inlier err.
// The test fl(r*r) < thr is evaluated as |r| < rcut, which is the same predicate for every double r because the
// rounded square is monotone in |r| (rcut is found on the host: square_cutoff()); the sum adds r*r by fma.
//
// Order of the inlier-error sum of one hypothesis (it fixes the bits, so every kernel below follows it and a
// hypothesis gets the same sum whichever kernel, workgroup, launch or GPU scores it).  PPL (matches per lane) is a
// function of N alone (1, 4 or 8); the matches are cut into chunks of 512*PPL; inside a chunk, match
// i = k*512 + w*64 + l belongs to slice w (0..7), lane position l (0..63), turn k (0..PPL-1):
//   p(w,l) = sum over k in order of r^2            (a chain of fma starting from 0, inliers only)
//   t(l)   = sum over w in order of p(w,l)
//   chunk  = tree over l of t(l): lanes l^1, then l^2, then 7-l within 8, then 15-l within 16, then rows of 16 as
//            (r0 + r1) + (r2 + r3)                 (wave_sum_f64 and the multi-value forms below build exactly it)
//   errsum = sum over chunks in order.
// ------------------------------------------------------------------------------------------------
#define SLAM_SLICES 8  // slices of 64 lane positions per turn: the eight waves of the fused kernel's workgroup

template <int PPL, int WAVES>
struct MatchChunk {  // PPL matches per lane of a chunk of 64*WAVES*PPL, plus which of them are real
    double a0[PPL], a1[PPL], b0[PPL], b1[PPL];
    unsigned long long livemask[PPL];  // wave-uniform lane masks of the real matches
    __device__ __forceinline__ void load(const double2* __restrict__ p0, const double2* __restrict__ p1, int N, int base) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int i = base + k * 64 * WAVES + (int)threadIdx.x;
            const bool live = i < N;
            livemask[k] = __ballot(live);
            // one 32-bit byte offset serves both arrays (scalar base + vector offset addressing; the callers keep
            // N below 2^27 matches)
            const uint32_t off = (uint32_t)(live ? i : N - 1) * (uint32_t)sizeof(double2);
            const double2 a = *(const double2*)((const char*)p0 + off), b = *(const double2*)((const char*)p1 + off);
            a0[k] = a.x; a1[k] = a.y; b0[k] = b.x; b1[k] = b.y;
        }
    }
};

// Sampson distance of one match (optional scoring mode, not the reference's): r^2 / (l0^2 + l1^2 + m0^2 + m1^2) with
// l = F^T x1 and m = F x0.  The same expression serves the counts, the sums and the winner's mask.
__device__ __forceinline__ double sampson_error(double r, double l0, double l1, double a0, double a1, double f00,
                                                double f01, double f02, double f10, double f11, double f12) {
    const double m0 = fma(f00, a0, fma(f01, a1, f02));
    const double m1 = fma(f10, a0, fma(f11, a1, f12));
    const double den = fma(l0, l0, fma(l1, l1, fma(m0, m0, m1 * m1)));
    return (r * r) * fast_rcp(den);
}

// residual of one match against one F (and, in Sampson mode, the distance in its place)
template <bool SAMPSON>
__device__ __forceinline__ double residual(double a0, double a1, double b0, double b1, double f00, double f01, double f02,
                                           double f10, double f11, double f12, double f20, double f21, double f22) {
    const double l0 = fma(b0, f00, fma(b1, f10, f20));
    const double l1 = fma(b0, f01, fma(b1, f11, f21));
    const double l2 = fma(b0, f02, fma(b1, f12, f22));
    const double r = fma(l0, a0, fma(l1, a1, l2));
    if (SAMPSON) return sampson_error(r, l0, l1, a0, a1, f00, f01, f02, f10, f11, f12);
    return r;
}

// es += (|r| < rcut on the lanes of `live`) ? r*r : 0; returns the number of those lanes.
// fl(r*r) < thr  <=>  |r| < rcut (rcut: the smallest double whose rounded square reaches thr).  The compare runs on
// the real matches only and writes the execution mask itself (and the same bits to a scalar pair), so the inlier lanes
// alone accumulate r*r: no select instructions, and no scalar instruction between the compare and the accumulation.
// The population count of the scalar copy is the inlier count (one scalar instruction, kept here so that the mask
// does not have to stay alive).
// COUNT_IN_ASM: the running count is added inside the statement, on the scalar unit.  The fused kernel wants that:
// left to the compiler, the 64 population counts of a chunk stay in scalar registers and are summed by vector
// instructions after the loop, on the way to LDS - 64 instructions between the end of the loop and the barrier.
template <bool COUNT_IN_ASM>
__device__ __forceinline__ void accumulate_inliers(double r, double rcut, unsigned long long live,
                                                   unsigned long long full_exec, double& es, int& count) {
    unsigned long long lt_mask;
    int n;
    if (COUNT_IN_ASM) {
        asm volatile(
            "s_mov_b64 exec, %[live]\n\t"
            "v_cmpx_lt_f64_e64 %[lt], |%[r]|, %[rc]\n\t"
            "v_fma_f64 %[es], %[r], %[r], %[es]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "s_bcnt1_i32_b64 %[n], %[lt]\n\t"
            "s_add_i32 %[c], %[c], %[n]"
            : [es] "+v"(es), [lt] "=&s"(lt_mask), [n] "=&s"(n), [c] "+s"(count)  // written before the inputs are all read
            : [r] "v"(r), [rc] "s"(rcut), [live] "s"(live), [full] "s"(full_exec)
            : "scc");
    } else {
        asm volatile(
            "s_mov_b64 exec, %[live]\n\t"
            "v_cmpx_lt_f64_e64 %[lt], |%[r]|, %[rc]\n\t"
            "v_fma_f64 %[es], %[r], %[r], %[es]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "s_bcnt1_i32_b64 %[n], %[lt]"
            : [es] "+v"(es), [lt] "=&s"(lt_mask), [n] "=&s"(n)  // written before the inputs are all read
            : [r] "v"(r), [rc] "s"(rcut), [live] "s"(live), [full] "s"(full_exec)
            : "scc");
        count += n;
    }
}

// Sampson mode: r is the distance itself, added as it is
__device__ __forceinline__ int accumulate_inliers_sampson(double d, double cut, unsigned long long live, double& es) {
    const bool in = d < cut;
    const unsigned long long m = __ballot(in) & live;
    es += (((m >> (threadIdx.x & 63)) & 1ull) != 0) ? d : 0.0;
    return __popcll(m);
}

// lane mask of the n lowest lanes (all for n >= 64, none for n <= 0); n is wave-uniform.  Scalar instructions only.
__device__ __forceinline__ unsigned long long low_lanes_mask(int n) {
    unsigned long long m;
    int t;
    asm("s_max_i32 %[t], %[n], 0\n\t"
        "s_bfm_b64 %[m], %[t], 0\n\t"      // ((1 << t[5:0]) - 1) << 0
        "s_cmp_ge_i32 %[n], 64\n\t"
        "s_cselect_b64 %[m], -1, %[m]"
        : [m] "=&s"(m), [t] "=&s"(t)
        : [n] "s"(n)
        : "scc");
    return m;
}

// Score phase of the fused kernel.  A workgroup of eight waves scores HB = 8 hypotheses (F in LDS) against ALL N
// matches: wave w owns slice w of every chunk, each lane keeps PPL matches of the chunk in registers and loops over
// the hypotheses.  The lanes' partial sums p(w,l) of the eight hypotheses meet in LDS; wave j then adds the slices of
// hypothesis j in order and reduces over the lanes.
template <int HB, int PPL, int WAVES, bool SAMPSON>
__device__ __forceinline__ void score_hypotheses(const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                 MatchChunk<PPL, WAVES>& m, const double (*s_F)[9], double rcut,
                                                 int h0, int H, int32_t* __restrict__ counts,
                                                 double* __restrict__ errsums, double (*s_part)[WAVES][64],
                                                 int (*s_cnt)[WAVES]) {
    static_assert((HB == 16 || HB == 8 || HB == 4) && WAVES == SLAM_SLICES, "wave j folds hypotheses j, j + 8; the slices are the eight waves");
    constexpr int THREADS = 64 * WAVES;
    constexpr int ROUNDS = HB > WAVES ? HB / WAVES : 1;  // hypotheses a wave folds
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // (the lane number is recomputed where it is needed instead of being kept in a register across the loop)
    const unsigned long long full_exec = __ballot(true);  // every lane of the workgroup is active here
    double total[ROUNDS];   // of hypotheses `wave`, `wave + 8`, over the chunks so far (the same on every lane)
    int ctotal[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) { total[r] = 0.0; ctotal[r] = 0; }
    // m holds the first chunk on entry and is refilled for the following ones
    for (int base = 0; base < N; base += THREADS * PPL) {
        if (base > 0) {
            __syncthreads();  // the previous chunk's partial sums have been read
            m.load(p0, p1, N, base);
        }
        int cnt[HB];  // wave-uniform: ballot popcounts
        SLAM_STAMP_W8(0);
        // F of the next hypothesis is fetched from LDS while the current one is scored (its latency would otherwise
        // be paid eight times per chunk by a wave that shares its SIMD with one other wave only)
        double fc[9], fn[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) fc[q] = s_F[0][q];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            if (j + 1 < HB) {
#pragma unroll
                for (int q = 0; q < 9; ++q) fn[q] = s_F[j + 1][q];
            }
            asm volatile("" ::: "memory");  // later hypotheses' F loads stay below: hoisting all of them costs 144 registers
            // The two waves of a SIMD issue oldest first; left alone, the older wave runs at the rate of a wave that
            // has the SIMD to itself and the younger one finishes its last third alone, at that same (slower) rate.
            // The wave that is behind gets the higher priority (set in the loop below, by progress).
            double es = 0.0;
            cnt[j] = 0;
            // two matches at a time: both residuals first (independent chains), then the masked accumulations in order
            constexpr int KB = PPL >= 2 ? 2 : 1;
#pragma unroll
            for (int k0 = 0; k0 < PPL; k0 += KB) {
                {   // Priority by progress q of STEPS.  A wave with priority gains on the other one slowly (it cannot
                    // use every issue slot), and what matters is how far apart the two are at the END, so the levels
                    // get shorter towards it: the first half of the loop, then 5/16, 1/8 and the last 1/16 of it
                    // (four equal levels left the younger wave 1 000 cycles behind at N=4096, these 500).
                    constexpr int STEPS = HB * PPL / KB;
                    constexpr int L0 = STEPS / 16 > 0 ? STEPS / 16 : 1, L1 = L0 + (STEPS / 8 > 0 ? STEPS / 8 : 1), L2 = STEPS / 2;
                    const int q = j * (PPL / KB) + k0 / KB;
                    auto level = [](int qq) { return qq >= STEPS - L0 ? 0 : (qq >= STEPS - L1 ? 1 : (qq >= STEPS - L2 ? 2 : 3)); };
                    if (q == 0 || level(q) != level(q - 1)) {
                        if (level(q) == 3) __builtin_amdgcn_s_setprio(3);
                        else if (level(q) == 2) __builtin_amdgcn_s_setprio(2);
                        else if (level(q) == 1) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                }
                double r[KB];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                    r[kk] = residual<SAMPSON>(m.a0[k0 + kk], m.a1[k0 + kk], m.b0[k0 + kk], m.b1[k0 + kk], fc[0], fc[1], fc[2],
                                              fc[3], fc[4], fc[5], fc[6], fc[7], fc[8]);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
                    if (SAMPSON) cnt[j] += accumulate_inliers_sampson(r[kk], rcut, m.livemask[k0 + kk], es);
                    else accumulate_inliers<true>(r[kk], rcut, m.livemask[k0 + kk], full_exec, es, cnt[j]);
            }
            {   // this lane's partial sum p(w,l) of hypothesis j
                const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                s_part[j][wave][lane] = es;
            }
            if (j + 1 < HB) {
#pragma unroll
                for (int q = 0; q < 9; ++q) fc[q] = fn[q];
            }
        }
        SLAM_STAMP(4);
        SLAM_STAMP_W8(1);
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < HB; ++j) s_cnt[j][wave] = cnt[j];
        }
        SLAM_STAMP(6);
        __syncthreads();
        SLAM_STAMP(5);
        SLAM_STAMP_W8(2);
        // hypothesis `wave` (and `wave + 8` of sixteen): the slices in order on every lane position, then the tree over
        // the lane positions (with fewer hypotheses than waves the surplus waves fold the last one again; their result
        // is dropped)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int hw = HB < WAVES ? min(wave, HB - 1) : wave + r * WAVES;
            double pw[WAVES];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) pw[w] = s_part[hw][w][lane];
            int c = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) c += s_cnt[hw][w];
            double t = pw[0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += pw[w];
            total[r] += wave_sum_f64(t);
            ctotal[r] += c;
        }
        SLAM_STAMP_W8(3);
    }
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int hw = wave + r * WAVES;
        if (lane == 0 && hw < HB && h0 + hw < H) {
            counts[h0 + hw] = ctotal[r];
            errsums[h0 + hw] = total[r];
        }
    }
}

// K2 standalone (large H, after k_solve).  The matches of one chunk live in LDS for the whole life of the workgroup;
// every WAVE scores hypotheses of its own, HG at a time with their F in registers, against all matches of the chunk:
// no workgroup barrier and no exchange between waves in the loop, one wave reduction per HG hypotheses and chunk.
// Work unit u = chunk * nslots + slot: the workgroup stages the chunk and its waves take the groups of HG hypotheses
// g = slot * WAVES + wave, then + nslots * WAVES, ...  With one chunk (N <= 512*PPL) the results are final; with more,
// out_counts / out_errsums are [chunk][H] partial tables that k_combine_chunks folds in chunk order.
template <int HG>
__device__ __forceinline__ double wave_sum_hg(const double (&v)[HG]);

template <int HG, int PPL, int WAVES, bool SAMPSON>
__global__ __launch_bounds__(64 * WAVES) void k_score_lds(const double2* __restrict__ p0, const double2* __restrict__ p1,
                                                          int N, const double* __restrict__ Fall, int H, double rcut,
                                                          int nchunks, int nslots, int32_t* __restrict__ out_counts,
                                                          double* __restrict__ out_errsums) {
    constexpr int CH = 64 * SLAM_SLICES * PPL, THREADS = 64 * WAVES;
    static_assert(CH % THREADS == 0 || THREADS % CH == 0, "staging loop");
    // LDS image of the chunk, in the order the loop reads it: [slice w][turn k][image 0 / image 1][lane position], so
    // that one base address per slice serves all 2 * PPL reads of a slice through immediate offsets
    __shared__ double2 s_m[SLAM_SLICES][PPL][2][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long full_exec = __ballot(true);
    const int ngroups = (H + HG - 1) / HG;
    int vzero;  // a zero the compiler cannot see through: keeps the (wave-uniform) F loads on the vector side, so that
                // the F values are VGPR operands and an FMA never needs two scalar operands
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    SLAM_STAMP_WAVE(0, clock64());
    SLAM_STAMP_WAVE(6, wall_clock64());
    int stamp_group = 0;
    (void)stamp_group;
    for (int u = blockIdx.x; u < nchunks * nslots; u += gridDim.x) {
        const int chunk = __builtin_amdgcn_readfirstlane(u / nslots);  // (the division runs on the vector unit)
        const int slot = u - chunk * nslots;
        const int base = chunk * CH;
        if (u != (int)blockIdx.x) __syncthreads();  // everybody has finished with the previous chunk
        for (int q = tid; q < CH; q += THREADS) {  // local index q = k * 512 + w * 64 + l
            const int i = min(base + q, N - 1);
            const int k = q / (64 * SLAM_SLICES), w = (q / 64) % SLAM_SLICES, l = q % 64;
            s_m[w][k][0][l] = p0[i];
            s_m[w][k][1][l] = p1[i];
        }
        __syncthreads();
        const int nlive = __builtin_amdgcn_readfirstlane(min(N - base, CH));  // real matches of this chunk
        SLAM_STAMP_WAVE(1, clock64());
        for (int g = slot * WAVES + wave; g < ngroups; g += nslots * WAVES) {
            const int h0 = g * HG;
            double f[HG][9];
#pragma unroll
            for (int h = 0; h < HG; ++h) {
                // surplus slots rescore the last hypothesis.  The six entries that multiply come through scalar loads
                // (wave-uniform address) and stay scalar operands; the three that start an FMA chain go to vector
                // registers, because an FMA takes one scalar operand only.
                const double* __restrict__ Fs = Fall + (size_t)min(h0 + h, H - 1) * 9;
                const double* __restrict__ Fp = Fs + vzero;
#pragma unroll
                for (int q = 0; q < 6; ++q) f[h][q] = Fs[q];
#pragma unroll
                for (int q = 6; q < 9; ++q) f[h][q] = Fp[q];
            }
            double t[HG];
            int cnt[HG];
#pragma unroll
            for (int h = 0; h < HG; ++h) { t[h] = 0.0; cnt[h] = 0; }
            // The four waves of a SIMD issue oldest first: left alone, the oldest wave would run through all of its
            // groups and leave the youngest to finish by itself, at the rate one wave can issue.  A wave with more
            // work still to do gets the higher priority.  What counts is how far apart the waves are at the END, and a
            // wave with priority gains on the others slowly, so the levels get shorter towards the end: the last two
            // slices of a wave's last group, the four before them, the ten before those, everything earlier.
            const int left = (ngroups - 1 - g) / (nslots * WAVES);  // groups after this one
#pragma unroll 1
            for (int w = 0; w < SLAM_SLICES; ++w) {
                {
                    const int rem = left * SLAM_SLICES + (SLAM_SLICES - 1 - w);  // slices after this one (wave-uniform)
                    if (rem >= 16) __builtin_amdgcn_s_setprio(3);
                    else if (rem >= 6) __builtin_amdgcn_s_setprio(2);
                    else if (rem >= 2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                double p[HG];
#pragma unroll
                for (int h = 0; h < HG; ++h) p[h] = 0.0;
                // two turns at a time: the residuals of both matches against the HG hypotheses first (2 * HG independent
                // chains, so that a wave issues back to back even when it has its SIMD to itself), then the masked
                // accumulations in the order of the sum (turn k before turn k + 1)
                constexpr int KB = PPL >= 2 ? 2 : 1;
#pragma unroll
                for (int k0 = 0; k0 < PPL; k0 += KB) {
                    double r[KB][HG];
                    unsigned long long live[KB];
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk) {
                        const int first = (k0 + kk) * 64 * SLAM_SLICES + w * 64;  // local index of lane position 0
                        const double2 a = s_m[w][k0 + kk][0][lane], b = s_m[w][k0 + kk][1][lane];
                        live[kk] = low_lanes_mask(nlive - first);  // lanes whose match exists
#pragma unroll
                        for (int h = 0; h < HG; ++h)
                            r[kk][h] = residual<SAMPSON>(a.x, a.y, b.x, b.y, f[h][0], f[h][1], f[h][2], f[h][3], f[h][4],
                                                         f[h][5], f[h][6], f[h][7], f[h][8]);
                    }
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                        for (int h = 0; h < HG; ++h)
                            if (SAMPSON) cnt[h] += accumulate_inliers_sampson(r[kk][h], rcut, live[kk], p[h]);
                            else accumulate_inliers<false>(r[kk][h], rcut, live[kk], full_exec, p[h], cnt[h]);
                }
#pragma unroll
                for (int h = 0; h < HG; ++h) t[h] += p[h];
            }
            const double sum = wave_sum_hg<HG>(t);  // lane l < HG holds the total of hypothesis wave_sum_hg_slot(l)
            if (lane < HG) {
                const int slot_h = HG == 4 ? wave_sum4_slot(lane) : 0;
                int c = cnt[0];
#pragma unroll
                for (int h = 1; h < HG; ++h) c = slot_h == h ? cnt[h] : c;
                if (h0 + slot_h < H) {
                    out_counts[(size_t)chunk * H + h0 + slot_h] = c;
                    out_errsums[(size_t)chunk * H + h0 + slot_h] = sum;
                }
            }
#ifdef SLAM_STAMPS
            if (stamp_group < 3) SLAM_STAMP_WAVE(2 + stamp_group, clock64());
            ++stamp_group;
#endif
        }
    }
    __builtin_amdgcn_s_setprio(0);
    SLAM_STAMP_WAVE(5, clock64());
    SLAM_STAMP_WAVE(7, wall_clock64());
}

template <>
__device__ __forceinline__ double wave_sum_hg<4>(const double (&v)[4]) { return wave_sum4_f64(v); }
template <>
__device__ __forceinline__ double wave_sum_hg<1>(const double (&v)[1]) { return wave_sum_f64(v[0]); }

// chunks of a hypothesis folded in order (N > 512*PPL): counts[h] = sum_c pc[c][h], errsums[h] = sum_c pe[c][h]
__global__ __launch_bounds__(256) void k_combine_chunks(const int32_t* __restrict__ pc, const double* __restrict__ pe,
                                                        int nchunks, int H, int32_t* __restrict__ counts,
                                                        double* __restrict__ errsums) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    int c = 0;
    double e = 0.0;
    for (int k = 0; k < nchunks; ++k) { c += pc[(size_t)k * H + h]; e += pe[(size_t)k * H + h]; }
    counts[h] = c;
    errsums[h] = e;
}

// ------------------------------------------------------------------------------------------------
// K3: selection   (reference: src/epipolar.py:78-81 applied in iteration order)
//   a later hypothesis replaces the incumbent iff it has more inliers, or as many and a strictly smaller errsum;
//   hence the winner is the lexicographic best of (count desc, errsum asc, index asc).
// Every workgroup of the grid finds the winner for itself; workgroup 0 emits its record, all of them share the
// winner's inlier mask.
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

// best of the 16 lanes of a row, on every lane of the row: four DPP exchanges (lane^1, lane^2, the mirrored halves
// of 8 and of 16 lanes), no LDS traffic.  The relation is a total order, so the pairing does not matter.
template <int CTRL>
__device__ __forceinline__ Best best_with_dpp_partner(const Best& b) {
    Best o;
    o.count = dpp_i32<CTRL>(b.count);
    o.index = dpp_i32<CTRL>(b.index);
    o.errsum = dpp_f64<CTRL>(b.errsum);
    return better(o, b) ? o : b;
}

__device__ __forceinline__ Best row_best(Best b) {
    b = best_with_dpp_partner<0xB1>(b);   // quad_perm [1,0,3,2]
    b = best_with_dpp_partner<0x4E>(b);   // quad_perm [2,3,0,1]
    b = best_with_dpp_partner<0x141>(b);  // row_half_mirror
    b = best_with_dpp_partner<0x140>(b);  // row_mirror
    return b;
}

__device__ __forceinline__ Best wave_best(Best b) {
    b = row_best(b);
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
        Best o;
        o.count = __shfl_xor(b.count, m, 64);
        o.index = __shfl_xor(b.index, m, 64);
        o.errsum = __shfl_xor(b.errsum, m, 64);
        if (better(o, b)) b = o;
    }
    return b;
}

// first stage for large H: every workgroup reduces a contiguous range of SLAM_SELECT_SPAN hypotheses to its best
// (same order relation, so the two stages together equal one pass); k_select then runs on the range winners
#define SLAM_SELECT_SPAN 8192
__global__ __launch_bounds__(1024) void k_select_partial(const int32_t* __restrict__ counts,
                                                         const double* __restrict__ errsums, int H,
                                                         int32_t* __restrict__ p_counts, double* __restrict__ p_errsums,
                                                         int32_t* __restrict__ p_index) {
    __shared__ Best s_best[16];
    const int tid = threadIdx.x;
    const int lo = blockIdx.x * SLAM_SELECT_SPAN, hi = min(H, lo + SLAM_SELECT_SPAN);
    Best b{-1, 0x7fffffff, 0.0};
    for (int h = lo + tid; h < hi; h += 1024) {
        const Best c{counts[h], h, errsums[h]};
        if (better(c, b)) b = c;
    }
    b = wave_best(b);
    if ((tid & 63) == 0) s_best[tid >> 6] = b;
    __syncthreads();
    if (tid == 0) {
        Best w = s_best[0];
        for (int q = 1; q < 16; ++q)
            if (better(s_best[q], w)) w = s_best[q];
        p_counts[blockIdx.x] = w.count;
        p_errsums[blockIdx.x] = w.errsum;
        p_index[blockIdx.x] = w.index;
    }
}

// threads of a selection workgroup (and matches it flags per pass)
#ifndef SLAM_SELECT_THREADS
#define SLAM_SELECT_THREADS 1024
#endif
// Selection by one workgroup of T threads, number `wg` of `n_wgs` that share the matches of the winner's mask.
// s_best: T / 64 entries of LDS.
template <int T, bool SAMPSON>
__device__ __forceinline__ void select_body(const int32_t* __restrict__ counts, const double* __restrict__ errsums,
                                            const int32_t* __restrict__ index_of,  // NULL: entry h is hypothesis h
                                            const double* __restrict__ Fall, int H, int64_t it0,
                                            const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                            double rcut, double* __restrict__ record, uint8_t* __restrict__ mask, int wg,
                                            int n_wgs, Best* s_best) {
    constexpr int W = T / 64;
    static_assert(W == 4 || W == 8 || W == 16, "the fold below handles 4, 8 or 16 wave winners");
    const int tid = threadIdx.x;
    SLAM_STAMP_SELECT(0);
    // The winner's mask needs every match: the grid's workgroups share the matches (workgroup x takes matches
    // x*T + tid, then strides by the grid), each finds the winner for itself - the reduction reads 12 bytes per
    // hypothesis, the mask 32 bytes per match through one CU's memory pipeline - and fetches its first match while
    // the reduction runs.  Workgroup 0 writes the record.
    const int i_first = wg * T + tid;
    const int stride = n_wgs * T;
    const int i_pre = i_first < N ? i_first : N - 1;
    const double2 a_pre = p0[i_pre], b_pre = p1[i_pre];
    // scan: four entries per thread and pass, their loads issued together (clamped, so that they need no branch)
    Best b{-1, 0x7fffffff, 0.0};
    for (int h0 = tid; h0 < H; h0 += 4 * T) {
        Best c[4];
        bool real[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = h0 + k * T;
            real[k] = h < H;
            const int hh = real[k] ? h : H - 1;
            c[k] = Best{counts[hh], index_of ? index_of[hh] : hh, errsums[hh]};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (real[k] && better(c[k], b)) b = c[k];
    }
    SLAM_STAMP_SELECT(1);
    b = wave_best(b);
    SLAM_STAMP_SELECT(2);
    if ((tid & 63) == 0) s_best[tid >> 6] = b;
    __syncthreads();
    SLAM_STAMP_SELECT(3);
    // every row of 16 lanes folds the wave winners the same way (lane l starts from winner l mod W): all lanes end with the same winner
    const Best w = row_best(s_best[tid & (W - 1)]);
    SLAM_STAMP_SELECT(4);
    const double* F = Fall + (size_t)w.index * 9;
    const double f00 = F[0], f01 = F[1], f02 = F[2], f10 = F[3], f11 = F[4], f12 = F[5], f20 = F[6], f21 = F[7],
                 f22 = F[8];
    if (tid == 0 && wg == 0) {
        record[SLAM_REC_COUNT] = (double)w.count;
        record[SLAM_REC_ERRSUM] = w.errsum;
        record[SLAM_REC_INDEX] = (double)(it0 + w.index);
        record[SLAM_REC_F + 0] = f00; record[SLAM_REC_F + 1] = f01; record[SLAM_REC_F + 2] = f02;
        record[SLAM_REC_F + 3] = f10; record[SLAM_REC_F + 4] = f11; record[SLAM_REC_F + 5] = f12;
        record[SLAM_REC_F + 6] = f20; record[SLAM_REC_F + 7] = f21; record[SLAM_REC_F + 8] = f22;
    }
    if (mask) {
        auto flag = [&](int i, const double2 a, const double2 bb) {
            const double l0 = fma(bb.x, f00, fma(bb.y, f10, f20));
            const double l1 = fma(bb.x, f01, fma(bb.y, f11, f21));
            const double l2 = fma(bb.x, f02, fma(bb.y, f12, f22));
            const double r = fma(l0, a.x, fma(l1, a.y, l2));
            if (SAMPSON)
                mask[i] = (sampson_error(r, l0, l1, a.x, a.y, f00, f01, f02, f10, f11, f12) < rcut) ? 1 : 0;
            else
                mask[i] = (fabs(r) < rcut) ? 1 : 0;  // same predicate as the scoring kernels
        };
        if (i_first < N) flag(i_first, a_pre, b_pre);
        for (int i = i_first + stride; i < N; i += stride) flag(i, p0[i], p1[i]);
    }
    SLAM_STAMP_SELECT(5);
}

template <bool SAMPSON>
__global__ __launch_bounds__(SLAM_SELECT_THREADS) void k_select(const int32_t* __restrict__ counts, const double* __restrict__ errsums,
                                                 const int32_t* __restrict__ index_of,  // NULL: entry h is hypothesis h
                                                 const double* __restrict__ Fall, int H, int64_t it0,
                                                 const double2* __restrict__ p0, const double2* __restrict__ p1, int N,
                                                 double rcut, double* __restrict__ record, uint8_t* __restrict__ mask,
                                                 const PairArgs* __restrict__ batch) {
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        counts = a.counts; errsums = a.errsums; Fall = a.F_all; H = a.H; p0 = a.n0; p1 = a.n1; N = a.N; record = a.record;
    }
    __shared__ Best s_best[SLAM_SELECT_THREADS / 64];
    select_body<SLAM_SELECT_THREADS, SAMPSON>(counts, errsums, index_of, Fall, H, it0, p0, p1, N, rcut, record, mask,
                                              (int)blockIdx.x, (int)gridDim.x, s_best);
}

// K1+K2 fused for small H.  A workgroup of eight waves owns eight hypotheses:
//   1. wave w draws the index set of hypothesis w (wave-parallel draw);
//   2. wave 0 solves the eight hypotheses, eight lanes each (the arithmetic of k_solve's one-lane solve, spread over
//      the lanes that own the columns, so F is bit-identical to the split path); the other waves request their
//      matches of the first chunk meanwhile, wave 0 after its solve;
//   3. the workgroup scores its hypotheses.
// One launch, no round trip of F through memory between solve and score, and at most 128 registers, so that two
// workgroups fit a CU and one of them can score while the other one solves.
// workgroup barrier that waits for this wave's LDS traffic only: global loads stay in flight across it
__device__ __forceinline__ void barrier_lds_only() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The selection of an EARLIER call, carried by a later call's fused kernel (the first n_wgs workgroups of the grid do
// it, with the fused kernel's 512 threads each): with several calls in flight on the context's internal streams a call's
// selection kernel is one launch more for the host to issue and one more link in its stream's chain of dependent
// launches, and it keeps four to eight workgroups busy for as long as a whole scoring kernel.  n_wgs = 0: nothing carried.
using CarriedSelect = slam_carried_select;  // ctx.h

// CARRY: the kernel can carry a selection (compiled for the large shape only: the small shapes are compiled for five
// waves per SIMD, 96 registers, and the extra role tips their register allocation into scratch).
template <int HB, int PPL, int WAVES, int SAMPLER, bool SAMPSON, bool CARRY = (PPL > 4)>
__global__ __launch_bounds__(64 * WAVES, (PPL <= 4 ? SLAM_SMALL_SHAPE_WAVES : 32) / WAVES) void k_ransac(
    // (what the first load of a wave needs comes first: the leading arguments are preloaded into scalar registers)
    const uint32_t* __restrict__ raw, int64_t raw_it0, int64_t it0, int H, int N, const double2* __restrict__ p0,
    const double2* __restrict__ p1, const int32_t* __restrict__ idx_in, uint64_t seed, double rcut,
    double* __restrict__ Fout, int32_t* __restrict__ counts, double* __restrict__ errsums, int32_t* __restrict__ flags,
    const PairArgs* __restrict__ batch, const CarriedSelect carried) {
    const int n_carried = CARRY ? __builtin_amdgcn_readfirstlane(carried.n_wgs) : 0;
    if (CARRY && (int)blockIdx.x < n_carried) {  // (uniform per workgroup)
        static_assert(WAVES == 8, "the carried selection runs with the fused kernel's 512 threads");
        __shared__ Best s_best[WAVES];
        if (carried.sampson)
            select_body<64 * WAVES, true>(carried.counts, carried.errsums, nullptr, carried.Fall, carried.H, carried.it0,
                                          carried.p0, carried.p1, carried.N, carried.rcut, carried.record, carried.mask,
                                          (int)blockIdx.x, n_carried, s_best);
        else
            select_body<64 * WAVES, false>(carried.counts, carried.errsums, nullptr, carried.Fall, carried.H, carried.it0,
                                           carried.p0, carried.p1, carried.N, carried.rcut, carried.record, carried.mask,
                                           (int)blockIdx.x, n_carried, s_best);
        return;
    }
    const int wg_index = (int)blockIdx.x - n_carried;
    if (batch) {
        const PairArgs& a = batch[blockIdx.y];
        p0 = a.n0; p1 = a.n1; N = a.N; H = a.H; Fout = a.F_all; counts = a.counts; errsums = a.errsums;
        if (wg_index * HB >= H) return;  // the grid is sized for the pair with most hypotheses
    }
    static_assert(HB == 16 || HB == 8 || HB == 4, "a wave draws hypotheses w, w + 8; a solving wave takes eight hypotheses, eight lanes each");
    constexpr int ROUNDS = HB > WAVES ? HB / WAVES : 1;  // hypotheses a wave draws
    constexpr int SOLVERS = HB > 8 ? HB / 8 : 1;         // waves that solve
    __shared__ double s_F[HB][9];
    __shared__ double s_part[HB][WAVES][64];  // partial sums p(w,l) of the eight hypotheses
    __shared__ int s_cnt[HB][WAVES];
    __shared__ int32_t s_sel[HB][8];
    __shared__ int32_t s_scratch[HB][16];
    __shared__ uint8_t s_pool[SAMPLER == SLAM_SAMPLER_CPYTHON ? SLAM_POOL_MAX * HB : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: the branches on it below are scalar branches
    const int h0 = wg_index * HB;
    SLAM_WALL_STAMP(0);
    SLAM_STAMP(0);
    SLAM_WG_LOG_BEGIN();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int hl = wave + r * WAVES;                  // hypothesis of the workgroup this wave draws now
        const int hs = min(h0 + min(hl, HB - 1), H - 1);  // surplus waves redo a hypothesis; their results are dropped
        uint32_t word = 0;
        if (SAMPLER == SLAM_SAMPLER_CPYTHON) word = raw[(size_t)(it0 + hs - raw_it0) * SLAM_RAW_WORDS + lane];
        if (hl < HB) {
            const bool ok = indices_for_wave<SAMPLER>(hs, N, word, idx_in, seed, it0, raw, raw_it0,
                                                      s_pool + (SAMPLER == SLAM_SAMPLER_CPYTHON ? SLAM_POOL_MAX * hl : 0),
                                                      s_scratch[hl], s_sel[hl]);
            if (!ok && lane == 0) raise_flag(flags);
        }
    }
    barrier_lds_only();
    SLAM_STAMP(1);
    // Wave 0 solves the workgroup's eight hypotheses with all of its lanes, eight lanes per hypothesis, one column
    // of the 9 x 8 system per lane (bit-identical to the one-lane solve of k_solve).  The matches of the first
    // chunk are requested by the other waves now and by wave 0 after its solve: inside the solve they would only
    // hold 64 registers, and with at most 128 two workgroups share a CU - one scores while the other one solves.
    MatchChunk<PPL, WAVES> first;
    if (wave < SOLVERS) {
        // (with four hypotheses the upper half of the wave repeats them; with sixteen, waves 0 and 1 take eight each)
        const int g = HB > 8 ? wave * 8 + (lane >> 3) : (lane >> 3) & (HB - 1), c = lane & 7;
        const int i = min(max(s_sel[g][c], 0), N - 1);  // never read outside the arrays, whatever idx holds
        double F[9];
        solve8_by_eight_lanes(p0[i], p1[i], F);
        if (c == 0 && (HB >= 8 || lane < 8 * HB)) {
#pragma unroll
            for (int q = 0; q < 9; ++q) s_F[g][q] = F[q];
            if (h0 + g < H) {
#pragma unroll
                for (int q = 0; q < 9; ++q) Fout[(size_t)(h0 + g) * 9 + q] = F[q];
            }
        }
        SLAM_STAMP(2);
        first.load(p0, p1, N, 0);
    } else {
        first.load(p0, p1, N, 0);  // in flight while wave 0 solves
    }
    barrier_lds_only();
    SLAM_STAMP(3);
    score_hypotheses<HB, PPL, WAVES, SAMPSON>(p0, p1, N, first, s_F, rcut, h0, H, counts, errsums, s_part, s_cnt);
    SLAM_WALL_STAMP(1);
    SLAM_WG_LOG_END((unsigned long long)(uintptr_t)Fout);
}

}  // namespace slamk
