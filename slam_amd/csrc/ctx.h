// Context, error handling and scratch management shared by the translation units of libslam_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/slam_amd.h"

#define SLAM_RAW_WORDS 96  // MT19937 output words kept per iteration for the CPython-compatible sampler

struct slam_buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct slam_prof {
    static const int kMax = 4096;
    hipEvent_t start[kMax];
    hipEvent_t stop[kMax];
    int kernel[kMax];
    int used = 0;
    bool created = false;
    double total_ms[SLAM_K_COUNT] = {};
    int64_t launches[SLAM_K_COUNT] = {};
};

#ifndef SLAM_LANES
#define SLAM_LANES 4  // internal streams that independent RANSAC calls rotate over
#endif

// A selection that has not been launched yet: what k_select would be given.  With several calls in flight
// (slam_set_concurrency) the fused kernel of the NEXT call on the same internal stream carries it (its first n_wgs
// workgroups), or slam_synchronize and its like launch it on its own.
struct slam_carried_select {
    const int32_t* counts;
    const double* errsums;
    const double* Fall;
    const double2 *p0, *p1;
    double* record;
    uint8_t* mask;
    int64_t it0;
    double rcut;
    int H, N;
    int n_wgs;   // workgroups of 512 threads that share the winner's mask (0: nothing pending)
    int sampson;
};

struct slam_ctx {
    int device = 0;
    int num_cus = 256;                  // compute units of the device
    int score_workgroups = 0;           // SLAM_AMD_SCORE_WORKGROUPS: workgroups of the large-H scoring kernel (0: one per CU)
    hipStream_t own_stream = nullptr;   // lane 0
    hipStream_t stream = nullptr;       // the context's ordered stream: own_stream, or the caller's (slam_set_stream)
    hipStream_t cur = nullptr;          // stream the launch helper uses right now (stream, or a lane)
    hipStream_t lane_stream[SLAM_LANES] = {};
    unsigned next_lane = 0;
    int lanes_in_use = 1;               // slam_set_concurrency
    bool sampson = false;               // slam_set_scoring
    bool inject_sampler_fault = false;  // SLAM_AMD_FAULT_INJECT=sampler (tests of the SLAM_ESAMPLER path)
    char err[512] = {0};

    // device scratch (grown on demand, never shrunk)
    slam_buf pts;       // staged (N,2)+(N,2) points of the host entry points
    slam_buf norm;      // normalised copies of a pair's matches (pair pipeline)
    slam_buf idx;       // H*8 int32
    slam_buf F;         // H*9 f64        } lane 0's hypothesis scratch
    slam_buf counts;    // H int32        }
    slam_buf errsums;   // H f64          }
    slam_buf lane_F[SLAM_LANES], lane_counts[SLAM_LANES], lane_errsums[SLAM_LANES];  // lanes 1.. (index 0 unused)
    // Calls whose selection is carried by the next call's kernel write their hypotheses into two scratch sets per
    // internal stream in turn (one being read by the carried selection while the other is written), in one buffer:
    // set s at carry_scratch[lane].p + s * carry_set_bytes[lane].
    slam_buf carry_scratch[SLAM_LANES];
    size_t carry_set_bytes[SLAM_LANES] = {};
    unsigned carry_turn[SLAM_LANES] = {};
    bool f32_counts = true;                        // SLAM_AMD_F32_COUNTS=0 keeps long calls on the float64 scoring kernel
    bool report_f32_band = false;                  // SLAM_AMD_REPORT_F32_BAND=1: the float32 counting path prints its band statistics
    bool wide_workgroups = true;                   // SLAM_AMD_WIDE_WORKGROUPS=0: eight hypotheses per workgroup whatever is in flight
    bool carry_selection = true;                   // SLAM_AMD_CARRY_SELECTION=0 turns the carrying off
    slam_carried_select pending[SLAM_LANES] = {};  // n_wgs > 0: a selection waiting to be carried or launched
    slam_buf mask;      // N u8
    slam_buf record;    // SLAM_REC_LEN f64 + flags
    slam_buf raw;       // MT19937 words [raw_n][SLAM_RAW_WORDS]
    slam_buf X4;        // 2 candidate planes of N*3 f64 (candidates 0 and 2; 1 and 3 are their negatives)
    slam_buf X;         // N*3 f64 staging
    slam_buf small;     // small matrices: candidates, poses, T, E ...
    slam_buf flags;     // int32[4] error flags written by kernels: .p is the device address of h_flags (mapped host memory), never freed as a buffer
    slam_buf track;     // tracking step: work block + per-hypothesis blocks
    slam_buf batch;     // slam_init_pairs: all pairs' points, planes and per-pair blocks
    hipEvent_t lane_event[SLAM_LANES + 1] = {};  // created on first use
    hipEvent_t sync_event[SLAM_LANES] = {};      // slam_sync_all: one per internal stream, created on first use
    unsigned lanes_dirty = 0;                    // bit l: lane l (l >= 1) has had launches since the context last waited for it
    slam_buf score_parts[SLAM_LANES];   // per-chunk partial counts and sums of the scoring kernel (N beyond one chunk)
    slam_buf select_parts[SLAM_LANES];  // range winners of the two-stage selection (large H), per internal stream
    slam_buf match;     // descriptor matching: per-slice candidate keys, staged descriptors and points
    uint64_t raw_seed = 0;
    int64_t raw_it0 = 0, raw_n = 0;

    // pinned host staging for small results
    double* h_small = nullptr;  // 1024 doubles
    int32_t* h_flags = nullptr;
    // pinned staging area of the host entry points (uploads of the match lists, downloads of mask and record): one
    // copy through it replaces several copies from and to pageable memory, each of which the runtime stages by itself
    char* h_stage = nullptr;
    size_t h_stage_cap = 0;

    void* comm = nullptr;               // the context's own RCCL communicator (slam_comm_init_rank), or none
    int comm_ranks = 0;
    slam_buf xchg;                      // gathered winner records of the exchange step

    unsigned char* noalloc_scratch = nullptr;  // this device's address of the code object's scratch array

    bool profiling = false;
    slam_prof prof;
};

#define SLAM_HIP(ctx, call)                                                                               \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                                 \
            return SLAM_EHIP;                                                                             \
        }                                                                                                 \
    } while (0)

#define SLAM_REQUIRE(ctx, cond, msg)                                                    \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            snprintf((ctx)->err, sizeof((ctx)->err), "invalid argument: %s", msg);      \
            return SLAM_EINVAL;                                                         \
        }                                                                               \
    } while (0)

#define SLAM_TRY(call)            \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != SLAM_OK) return rc_; \
    } while (0)

// launch the selections that wait to be carried (slam_carried_select), each on its internal stream
int slam_flush_pending(slam_ctx* ctx);

// Wait for everything the context has queued, on its ordered stream and on every lane.
// A hipStreamSynchronize of a stream that has run anything since it was last waited for is a round trip between host
// and device (the runtime queues a marker and waits for it: 8 us), and one per stream makes them consecutive.  With more
// than one stream to wait for, an event is recorded behind the last launch of each - the markers travel together - and
// the host then waits for the events: 27 instead of 46 us for four streams that ran a 9 us kernel each
// (tools/activation_probe.hip).  Lanes that have launched nothing since the last wait are left alone.
static inline int slam_sync_all(slam_ctx* ctx) {
    SLAM_TRY(slam_flush_pending(ctx));
    hipStream_t waited[SLAM_LANES];
    hipEvent_t* events[SLAM_LANES];
    int n = 0;
    for (int l = 1; l < SLAM_LANES; ++l)
        if ((ctx->lanes_dirty >> l & 1u) && ctx->lane_stream[l] && ctx->lane_stream[l] != ctx->stream) {
            waited[n] = ctx->lane_stream[l];
            events[n++] = &ctx->sync_event[l];
        }
    if (n == 0) {
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SLAM_OK;
    }
    for (int i = 0; i < n; ++i) {
        if (!*events[i]) SLAM_HIP(ctx, hipEventCreateWithFlags(events[i], hipEventDisableTiming));
        SLAM_HIP(ctx, hipEventRecord(*events[i], waited[i]));
    }
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (its own round trip overlaps the markers')
    for (int i = 0; i < n; ++i) SLAM_HIP(ctx, hipEventSynchronize(*events[i]));
    ctx->lanes_dirty = 0;
    return SLAM_OK;
}

// make sure b holds at least `bytes`; contents are NOT preserved
static inline int slam_reserve(slam_ctx* ctx, slam_buf& b, size_t bytes) {
    if (bytes <= b.cap) return SLAM_OK;
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    // a buffer may still be in use by work queued on any of the context's streams
    SLAM_TRY(slam_sync_all(ctx));
    if (b.p) SLAM_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return SLAM_ENOMEM;
    }
    b.cap = want;
    return SLAM_OK;
}

// Launch on the context's current stream (ctx->cur).  With profiling on, the launch carries a start/stop event pair whose
// timestamps are those of the dispatch itself, so the elapsed time is the kernel's own duration.
int slam_prof_slot(slam_ctx* ctx, int kernel);  // next free event pair, or -1

template <typename... KArgs, typename... Args>
static inline int slam_launch(slam_ctx* ctx, int kernel_id, void (*kernel)(KArgs...), dim3 grid, dim3 block,
                              unsigned shmem, Args... args) {
    int slot = (ctx->profiling && kernel_id >= 0) ? slam_prof_slot(ctx, kernel_id) : -1;
    if (slot >= 0)
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, ctx->cur, ctx->prof.start[slot], ctx->prof.stop[slot], 0,
                              static_cast<KArgs>(args)...);
    else
        hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->cur, static_cast<KArgs>(args)...);
    SLAM_HIP(ctx, hipGetLastError());
    return SLAM_OK;
}
