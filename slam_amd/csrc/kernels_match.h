// Descriptor matching: brute-force Hamming 2-NN over 256-bit descriptors, ratio test, ordered gathering, gfx950.
//   reference: custom_slam.py:14-26 (match_knn_brute_force), :62-73 (ratio test and point lists)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamk {

// A candidate is one 32-bit key, (distance << 22) | train index: unsigned order on keys is the order the reference's
// stable sort produces (distance ascending, then train index ascending).  distance <= 256 and index < 2^22.
#define SLAM_MATCH_INDEX_BITS 22
#define SLAM_MATCH_EMPTY 0xFFFFFFFFu
#define SLAM_MATCH_MAX_TRAIN (1 << SLAM_MATCH_INDEX_BITS)

__device__ __forceinline__ void top2_insert(uint32_t key, uint32_t& k1, uint32_t& k2) {
    // {k1 <= k2} and key -> the two smallest of the three
    const uint32_t lo = min(k1, key);
    const uint32_t hi = max(k1, key);
    k2 = min(k2, hi);
    k1 = lo;
}

// ------------------------------------------------------------------------------------------------
// K8: partial 2-NN.  A workgroup of 4 waves owns 64 query descriptors (one per lane, 8 VGPRs) and one contiguous
// slice of the train descriptors (blockIdx.y); each wave scans a quarter of the slice.  A train descriptor is
// wave-uniform, so it arrives through scalar loads and is an SGPR operand of the XORs: 8 v_xor + 8 v_bcnt
// (accumulating) + 3 integer ops per (query, train) pair, no LDS traffic in the loop.
// part[slice * n0 + q] = the slice's two smallest keys of query q (slice-major: both kernels touch it coalesced).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_knn2(const uint32_t* __restrict__ des0, const uint32_t* __restrict__ des1,
                                                    int n0, int n1, int per_slice, uint2* __restrict__ part) {
    __shared__ uint32_t s_k[4][64][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform, so the train loop runs on the scalar unit
    const int q = blockIdx.x * 64 + lane;
    const int qq = q < n0 ? q : n0 - 1;
    uint32_t a[8];
    {
        const uint4 lo = *reinterpret_cast<const uint4*>(des0 + (size_t)qq * 8);
        const uint4 hi = *reinterpret_cast<const uint4*>(des0 + (size_t)qq * 8 + 4);
        a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w;
        a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
    }
    const int s_lo = blockIdx.y * per_slice;
    const int s_hi = min(n1, s_lo + per_slice);
    const int quarter = (per_slice + 3) >> 2;
    const int j_lo = min(s_hi, s_lo + wave * quarter);
    const int j_hi = min(s_hi, j_lo + quarter);
    uint32_t k1 = SLAM_MATCH_EMPTY, k2 = SLAM_MATCH_EMPTY;
#pragma unroll 4
    for (int j = j_lo; j < j_hi; ++j) {
        const uint32_t* __restrict__ t = des1 + (size_t)j * 8;  // uniform address: scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += __popc(a[w] ^ t[w]);
        top2_insert((d << SLAM_MATCH_INDEX_BITS) | (uint32_t)j, k1, k2);
    }
    s_k[wave][lane][0] = k1;
    s_k[wave][lane][1] = k2;
    __syncthreads();
    if (wave == 0 && q < n0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            top2_insert(s_k[w][lane][0], k1, k2);
            top2_insert(s_k[w][lane][1], k1, k2);
        }
        part[(size_t)blockIdx.y * n0 + q] = make_uint2(k1, k2);
    }
}

// ------------------------------------------------------------------------------------------------
// K9: fold the slices, emit (index, distance) pairs, apply the ratio test (custom_slam.py:67: best < 0.5 * second,
// i.e. 2 * best < second on integers) and gather the kept matches in query order (custom_slam.py:65-73).
// One workgroup of 1024 threads: an ordered compaction needs a running prefix over the queries.
// index/distance (n0 x 2, may be NULL): -1 where fewer than two train descriptors exist.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_match_finish(const uint2* __restrict__ part, int slices, int n0,
                                                       const double2* __restrict__ pts0, const double2* __restrict__ pts1,
                                                       int32_t* __restrict__ index, int32_t* __restrict__ distance,
                                                       double2* __restrict__ m0, double2* __restrict__ m1,
                                                       int32_t* __restrict__ n_matched) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // 4096 queries at a time, four per thread (q0 + j * 1024 + tid): the slice loop runs over all four at once, so
    // that sixteen loads are in flight per thread instead of four (26.4 -> 25.4 us per 4096 x 4096 call; what remains
    // is the 512 KB of partial keys passing through one CU: 8 k cycles at 64 bytes per cycle).
    constexpr int QPT = 4;
    for (int q0 = 0; q0 < n0; q0 += QPT * 1024) {
        uint32_t k1[QPT], k2[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) { k1[j] = SLAM_MATCH_EMPTY; k2[j] = SLAM_MATCH_EMPTY; }
#pragma unroll 4
        for (int s = 0; s < slices; ++s) {
            uint2 k[QPT];
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int q = q0 + j * 1024 + tid;
                k[j] = part[(size_t)s * n0 + (q < n0 ? q : n0 - 1)];
            }
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                top2_insert(k[j].x, k1[j], k2[j]);
                top2_insert(k[j].y, k1[j], k2[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int q = q0 + j * 1024 + tid;
            if (q0 + j * 1024 >= n0) break;  // uniform
            const int i1 = k1[j] == SLAM_MATCH_EMPTY ? -1 : (int)(k1[j] & (SLAM_MATCH_MAX_TRAIN - 1));
            const int d1 = k1[j] == SLAM_MATCH_EMPTY ? -1 : (int)(k1[j] >> SLAM_MATCH_INDEX_BITS);
            const int i2 = k2[j] == SLAM_MATCH_EMPTY ? -1 : (int)(k2[j] & (SLAM_MATCH_MAX_TRAIN - 1));
            const int d2 = k2[j] == SLAM_MATCH_EMPTY ? -1 : (int)(k2[j] >> SLAM_MATCH_INDEX_BITS);
            if (q < n0 && index) { index[2 * (size_t)q] = i1; index[2 * (size_t)q + 1] = i2; }
            if (q < n0 && distance) { distance[2 * (size_t)q] = d1; distance[2 * (size_t)q + 1] = d2; }
            const bool keep = q < n0 && i2 >= 0 && 2 * d1 < d2;
            const unsigned long long b = __ballot(keep);
            const int before = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave[wave] = __popcll(b);
            __syncthreads();
            int off = s_base;
            for (int w = 0; w < wave; ++w) off += s_wave[w];
            if (keep && m0) {
                m0[off + before] = pts0[q];
                m1[off + before] = pts1[i1];
            }
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < 16; ++w) tot += s_wave[w];
                s_base += tot;
            }
            __syncthreads();
        }
    }
    if (tid == 0) *n_matched = s_base;
}

// ------------------------------------------------------------------------------------------------
// The K nearest for any K up to SLAM_MATCH_MAX_K (custom_slam.py:24 slices the sorted list at any num_matches; the
// pipeline itself asks for two).  Same layout as the 2-NN pair above - a query per lane, a wave-uniform train
// descriptor through scalar loads, one 32-bit key per candidate - with the running best kept as a sorted register
// array: a new key sinks through it with one min and one max per position.
// part[(slice * n0 + q) * K + j]; out index / distance are (n0, k) with k <= K, -1 where fewer than k train
// descriptors exist.
// ------------------------------------------------------------------------------------------------
#define SLAM_MATCH_MAX_K 8

template <int K>
__device__ __forceinline__ void topk_insert(uint32_t key, uint32_t (&best)[K]) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const uint32_t lo = min(best[j], key);
        key = max(best[j], key);
        best[j] = lo;
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_match_knnk(const uint32_t* __restrict__ des0, const uint32_t* __restrict__ des1,
                                                    int n0, int n1, int per_slice, uint32_t* __restrict__ part) {
    __shared__ uint32_t s_k[4][K][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    const int qq = q < n0 ? q : n0 - 1;
    uint32_t a[8];
    {
        const uint4 lo = *reinterpret_cast<const uint4*>(des0 + (size_t)qq * 8);
        const uint4 hi = *reinterpret_cast<const uint4*>(des0 + (size_t)qq * 8 + 4);
        a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w;
        a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
    }
    const int s_lo = blockIdx.y * per_slice;
    const int s_hi = min(n1, s_lo + per_slice);
    const int quarter = (per_slice + 3) >> 2;
    const int j_lo = min(s_hi, s_lo + wave * quarter);
    const int j_hi = min(s_hi, j_lo + quarter);
    uint32_t best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = SLAM_MATCH_EMPTY;
#pragma unroll 2
    for (int j = j_lo; j < j_hi; ++j) {
        const uint32_t* __restrict__ t = des1 + (size_t)j * 8;  // uniform address: scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += __popc(a[w] ^ t[w]);
        topk_insert<K>((d << SLAM_MATCH_INDEX_BITS) | (uint32_t)j, best);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) s_k[wave][j][lane] = best[j];
    __syncthreads();
    if (wave == 0 && q < n0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int j = 0; j < K; ++j) topk_insert<K>(s_k[w][j][lane], best);
#pragma unroll
        for (int j = 0; j < K; ++j) part[((size_t)blockIdx.y * n0 + q) * K + j] = best[j];
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_match_finish_k(const uint32_t* __restrict__ part, int slices, int n0, int k,
                                                        int32_t* __restrict__ index, int32_t* __restrict__ distance) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n0) return;
    uint32_t best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = SLAM_MATCH_EMPTY;
    for (int s = 0; s < slices; ++s)
#pragma unroll
        for (int j = 0; j < K; ++j) topk_insert<K>(part[((size_t)s * n0 + q) * K + j], best);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            const bool none = best[j] == SLAM_MATCH_EMPTY;
            index[(size_t)q * k + j] = none ? -1 : (int)(best[j] & (SLAM_MATCH_MAX_TRAIN - 1));
            distance[(size_t)q * k + j] = none ? -1 : (int)(best[j] >> SLAM_MATCH_INDEX_BITS);
        }
    }
}

}  // namespace slamk
