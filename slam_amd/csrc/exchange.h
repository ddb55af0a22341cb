// The exchange step of the sharded runs, inside the library: RCCL collectives on the context's stream followed by the
// merge on the device, so that no result returns to the host between a rank's kernels and the global winner.
//   hypothesis ranges: all-gather of the 96-byte winner records, then the selection rule on every rank
//   point shards:      all-reduce (sum) of the four cheirality counts
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, single-GPU users
// never load it, and a process that already holds a copy (PyTorch ships one under the same soname) shares that copy.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "ctx.h"

namespace slamx {

// the few declarations of rccl.h that are used (kept local so that building needs no RCCL headers)
typedef int ncclResult_t;  // ncclSuccess = 0
typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { kNcclSum = 0, kNcclInt64 = 4, kNcclFloat64 = 8 };

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
    bool ok = false;
};

inline RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("SLAM_AMD_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            snprintf(api.why, sizeof api.why, "librccl.so.1 could not be loaded (%s)", dlerror());
            return;
        }
        auto sym = [&](const char* s) { return dlsym(api.handle, s); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.AllGather && api.AllReduce;
        if (!api.ok) snprintf(api.why, sizeof api.why, "the RCCL library lacks an expected entry point");
    });
    return api;
}

// Winner among R records of SLAM_REC_LEN doubles under the reference's rule extended across ranks (more inliers, then
// the smaller inlier-error sum, then the earlier iteration: src/epipolar.py:78-81), written to out (which may be one of
// the inputs' rank slots or a separate buffer: neither pointer is restrict-qualified, and the winner's twelve values
// are loaded before the barrier and stored after it).  One wave; R is the number of ranks, a handful.  A record with a
// NaN error sum never wins a comparison and never loses one it is the incumbent of: rank order decides.
__global__ __launch_bounds__(64) void k_merge_records(const double* recs, int R, double* out) {
    int best = 0;
    for (int r = 1; r < R; ++r) {
        const double* x = recs + (size_t)r * SLAM_REC_LEN;
        const double* y = recs + (size_t)best * SLAM_REC_LEN;
        bool wins;
        if (x[SLAM_REC_COUNT] != y[SLAM_REC_COUNT]) wins = x[SLAM_REC_COUNT] > y[SLAM_REC_COUNT];
        else if (x[SLAM_REC_ERRSUM] != y[SLAM_REC_ERRSUM]) wins = x[SLAM_REC_ERRSUM] < y[SLAM_REC_ERRSUM];
        else wins = x[SLAM_REC_INDEX] < y[SLAM_REC_INDEX];
        if (wins) best = r;
    }
    const double v = threadIdx.x < SLAM_REC_LEN ? recs[(size_t)best * SLAM_REC_LEN + threadIdx.x] : 0.0;
    __syncthreads();  // every load of the wave has completed before the first store (out may overlap recs)
    if (threadIdx.x < SLAM_REC_LEN) out[threadIdx.x] = v;
}

}  // namespace slamx

#define SLAM_RCCL(ctx, call)                                                                               \
    do {                                                                                                   \
        slamx::ncclResult_t r_ = (call);                                                                   \
        if (r_ != 0) {                                                                                     \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s", #call,                               \
                     slamx::rccl().GetErrorString ? slamx::rccl().GetErrorString(r_) : "RCCL error");      \
            return SLAM_EHIP;                                                                              \
        }                                                                                                  \
    } while (0)

static inline int slam_need_rccl(slam_ctx* ctx) {
    slamx::RcclApi& api = slamx::rccl();
    if (!api.ok) {
        snprintf(ctx->err, sizeof ctx->err, "RCCL is not available: %s", api.why);
        return SLAM_EHIP;
    }
    return SLAM_OK;
}
