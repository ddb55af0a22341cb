// libslam_amd.so — host side of the C ABI declared in include/slam_amd.h (gfx950 only).
#include <algorithm>
#include <cstdlib>
#include <limits>
#include <cmath>
#include <new>
#include <thread>
#include <vector>

#include "ctx.h"
#include "exchange.h"
#include "kernels_ransac.h"
#include "kernels_count32.h"
#include "kernels_tri.h"
#include "kernels_track.h"
#include "kernels_match.h"

using namespace slamk;

static char g_create_err[512] = "";

// Hypothesis scratch of slam_ransac_f_dev_noalloc: F (H x 9 f64), inlier-error sums (H f64) and inlier counts (H i32)
// for at most SLAM_NOALLOC_MAX_H hypotheses, in device memory that belongs to the code object, so that the entry
// point needs no allocation.  Every GPU the library runs on holds its own copy (1.3 MiB) until the process ends.
// In host code the array's NAME is not a device address: slam_create asks hipGetSymbolAddress for this device's
// copy and keeps the pointer in the context; host code uses that pointer only.
#define SLAM_NOALLOC_F_BYTES ((size_t)SLAM_NOALLOC_MAX_H * 9 * sizeof(double))
#define SLAM_NOALLOC_E_BYTES ((size_t)SLAM_NOALLOC_MAX_H * sizeof(double))
#define SLAM_NOALLOC_C_BYTES ((size_t)SLAM_NOALLOC_MAX_H * sizeof(int32_t))
static __device__ __align__(256) unsigned char scratch[SLAM_NOALLOC_F_BYTES + SLAM_NOALLOC_E_BYTES + SLAM_NOALLOC_C_BYTES];

// ================================================================================================
// lifecycle
// ================================================================================================
extern "C" int slam_create(int device, slam_ctx** out) {
    if (!out) return SLAM_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_create_err, sizeof g_create_err, "no HIP device available (%s)", hipGetErrorString(e));
        return SLAM_ENODEVICE;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "device %d out of range (0..%d)", device, ndev - 1);
        return SLAM_EINVAL;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        snprintf(g_create_err, sizeof g_create_err, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return SLAM_EHIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_create_err, sizeof g_create_err, "device %d is %s; this library carries gfx950 code only", device,
                 prop.gcnArchName);
        return SLAM_ENODEVICE;
    }
    slam_ctx* ctx = new (std::nothrow) slam_ctx();
    if (!ctx) return SLAM_ENOMEM;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* g = getenv("SLAM_AMD_SCORE_WORKGROUPS")) {  // tuning aid: workgroups of the large-H scoring kernel only
        const int v = atoi(g);
        if (v > 0) ctx->score_workgroups = v;
    }
    auto fail = [&](const char* what, hipError_t err) {
        snprintf(g_create_err, sizeof g_create_err, "%s: %s", what, hipGetErrorString(err));
        slam_destroy(ctx);  // releases the streams and pinned buffers created so far
        return SLAM_EHIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    // The context's streams are created at the device's greatest stream priority.  The HIP runtime keeps one pool of
    // hardware queues (4 by default) per priority level and shares a pool's queues among all streams of that level:
    // at the default level the context's four streams end up sharing queues with each other as soon as the process
    // has other streams (its null stream, a copy queue, a framework's or a collective library's streams), and steps
    // meant to overlap run one after the other (measured: 9.5 us per step instead of 6.9).  A level of their own
    // gives them one hardware queue each whatever else the process did before.  SLAM_AMD_STREAM_PRIORITY=normal|low
    // chooses another level.
    int prio_least = 0, prio_greatest = 0;
    if ((e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest)) != hipSuccess)
        return fail("hipDeviceGetStreamPriorityRange", e);
    int prio = prio_greatest;
    if (const char* want = getenv("SLAM_AMD_STREAM_PRIORITY")) {
        if (!strcmp(want, "normal")) prio = 0;
        else if (!strcmp(want, "low")) prio = prio_least;
        else if (strcmp(want, "high")) {
            snprintf(g_create_err, sizeof g_create_err, "SLAM_AMD_STREAM_PRIORITY=%s: expected high, normal or low", want);
            slam_destroy(ctx);
            return SLAM_EINVAL;
        }
    }
    // test hook: SLAM_AMD_FAULT_INJECT=sampler makes the MT19937 table useless (all words rejected), so that the
    // SLAM_ESAMPLER path of every entry point can be exercised
    if (const char* inject = getenv("SLAM_AMD_FAULT_INJECT")) ctx->inject_sampler_fault = !strcmp(inject, "sampler");
    // SLAM_AMD_F32_COUNTS=0: long calls count in float64 as well (kernels_count32.h is not used)
    if (const char* c = getenv("SLAM_AMD_F32_COUNTS")) ctx->f32_counts = strcmp(c, "0") != 0;
    if (const char* c = getenv("SLAM_AMD_REPORT_F32_BAND")) ctx->report_f32_band = strcmp(c, "0") != 0;
    // SLAM_AMD_WIDE_WORKGROUPS=0: eight hypotheses per workgroup of the fused kernel also with several calls in flight
    if (const char* c = getenv("SLAM_AMD_WIDE_WORKGROUPS")) ctx->wide_workgroups = strcmp(c, "0") != 0;
    // SLAM_AMD_CARRY_SELECTION=0: every call launches its own selection kernel also with several calls in flight
    if (const char* c = getenv("SLAM_AMD_CARRY_SELECTION")) ctx->carry_selection = strcmp(c, "0") != 0;
    if ((e = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, prio)) != hipSuccess)
        return fail("hipStreamCreate", e);
    ctx->stream = ctx->cur = ctx->own_stream;
    ctx->lane_stream[0] = ctx->own_stream;
    for (int l = 1; l < SLAM_LANES; ++l)
        if ((e = hipStreamCreateWithPriority(&ctx->lane_stream[l], hipStreamNonBlocking, prio)) != hipSuccess)
            return fail("hipStreamCreate", e);
    if ((e = hipHostMalloc((void**)&ctx->h_small, 1024 * sizeof(double), hipHostMallocDefault)) != hipSuccess)
        return fail("hipHostMalloc", e);
    // error flags: host memory the kernels write directly (linalg.h: raise_flag); flags.p is its device address
    if ((e = hipHostMalloc((void**)&ctx->h_flags, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
        return fail("hipHostMalloc", e);
    memset(ctx->h_flags, 0, 16 * sizeof(int32_t));
    if ((e = hipHostGetDevicePointer(&ctx->flags.p, ctx->h_flags, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
    // this device's copy of the code object's scratch array (the device is current here)
    if ((e = hipGetSymbolAddress((void**)&ctx->noalloc_scratch, HIP_SYMBOL(scratch))) != hipSuccess)
        return fail("hipGetSymbolAddress(scratch)", e);
    int rc = slam_reserve(ctx, ctx->record, 64 * sizeof(double));
    if (rc == SLAM_OK) rc = slam_reserve(ctx, ctx->small, (1024 + 1024) * sizeof(double));  // small matrices, then the count sets of the bulk triangulation
    if (rc != SLAM_OK) {
        snprintf(g_create_err, sizeof g_create_err, "%s", ctx->err);
        slam_destroy(ctx);
        return rc;
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    *out = ctx;
    return SLAM_OK;
}

extern "C" void slam_destroy(slam_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // selections that were never waited for are dropped, not launched: their output buffers are the caller's and may be gone
    for (int l = 0; l < SLAM_LANES; ++l) ctx->pending[l].n_wgs = 0;
    if (ctx->stream) (void)slam_sync_all(ctx);  // a context that failed half-way through slam_create has no stream yet
    if (ctx->comm && slamx::rccl().ok) (void)slamx::rccl().CommDestroy((slamx::ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    if (ctx->xchg.p) (void)hipFree(ctx->xchg.p);
    slam_buf* bufs[] = {&ctx->pts, &ctx->norm, &ctx->idx, &ctx->F, &ctx->counts, &ctx->errsums, &ctx->mask, &ctx->record,
                        &ctx->raw, &ctx->X4, &ctx->X, &ctx->small, &ctx->track, &ctx->match, &ctx->batch};
    for (slam_buf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int l = 0; l < SLAM_LANES; ++l) {
        if (ctx->lane_F[l].p) (void)hipFree(ctx->lane_F[l].p);
        if (ctx->select_parts[l].p) (void)hipFree(ctx->select_parts[l].p);
        if (ctx->score_parts[l].p) (void)hipFree(ctx->score_parts[l].p);
        if (ctx->lane_counts[l].p) (void)hipFree(ctx->lane_counts[l].p);
        if (ctx->lane_errsums[l].p) (void)hipFree(ctx->lane_errsums[l].p);
        if (ctx->carry_scratch[l].p) (void)hipFree(ctx->carry_scratch[l].p);
        if (l > 0 && ctx->lane_stream[l]) (void)hipStreamDestroy(ctx->lane_stream[l]);
    }
    for (hipEvent_t ev : ctx->lane_event)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->sync_event)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->prof.created)
        for (int i = 0; i < slam_prof::kMax; ++i) {
            (void)hipEventDestroy(ctx->prof.start[i]);
            (void)hipEventDestroy(ctx->prof.stop[i]);
        }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

#ifndef SLAM_BUILD_HASH
#define SLAM_BUILD_HASH "unknown"
#endif
extern "C" const char* slam_build_hash(void) { return SLAM_BUILD_HASH; }

extern "C" const char* slam_last_error(const slam_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int slam_set_stream(slam_ctx* ctx, void* hip_stream) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_TRY(slam_sync_all(ctx));
    ctx->stream = ctx->cur = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SLAM_OK;
}

extern "C" int slam_set_concurrency(slam_ctx* ctx, int lanes) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, lanes >= 1 && lanes <= SLAM_LANES, "lanes out of range (1..4)");
    SLAM_TRY(slam_sync_all(ctx));
    ctx->lanes_in_use = lanes;
    ctx->next_lane = 0;
    return SLAM_OK;
}

extern "C" int slam_host_alloc(slam_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, out != nullptr && bytes > 0, "out is NULL or bytes is 0");
    *out = nullptr;
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        *out = nullptr;
        snprintf(ctx->err, sizeof ctx->err, "page-locked allocation of %llu bytes failed", (unsigned long long)bytes);
        return SLAM_ENOMEM;
    }
    return SLAM_OK;
}

extern "C" int slam_host_free(slam_ctx* ctx, void* p) {  // ctx may be NULL (memory may outlive the context it came from)
    if (!p) return SLAM_OK;
    if (!ctx) return hipHostFree(p) == hipSuccess ? SLAM_OK : SLAM_EHIP;
    SLAM_HIP(ctx, hipHostFree(p));
    return SLAM_OK;
}

extern "C" int slam_set_scoring(slam_ctx* ctx, int mode) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, mode == SLAM_SCORE_ALGEBRAIC || mode == SLAM_SCORE_SAMPSON, "unknown scoring mode");
    ctx->sampson = mode == SLAM_SCORE_SAMPSON;
    return SLAM_OK;
}

static int check_flags(slam_ctx* ctx);

extern "C" int slam_synchronize(slam_ctx* ctx) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_TRY(slam_sync_all(ctx));
    // the asynchronous entry points cannot report a sampler that ran out of words when they return: it shows here
    return check_flags(ctx);
}

// ================================================================================================
// kernel timing
// ================================================================================================
static int prof_flush(slam_ctx* ctx) {
    slam_prof& p = ctx->prof;
    if (p.used == 0) return SLAM_OK;
    SLAM_TRY(slam_sync_all(ctx));
    for (int i = 0; i < p.used; ++i) {
        float ms = 0.f;
        SLAM_HIP(ctx, hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        p.total_ms[p.kernel[i]] += ms;
        p.launches[p.kernel[i]] += 1;
    }
    p.used = 0;
    return SLAM_OK;
}

int slam_prof_slot(slam_ctx* ctx, int kernel) {
    slam_prof& p = ctx->prof;
    if (p.used == slam_prof::kMax && prof_flush(ctx) != SLAM_OK) return -1;
    p.kernel[p.used] = kernel;
    return p.used++;
}

extern "C" int slam_profile_enable(slam_ctx* ctx, int on) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->prof.created) {
        for (int i = 0; i < slam_prof::kMax; ++i) {
            SLAM_HIP(ctx, hipEventCreate(&ctx->prof.start[i]));
            SLAM_HIP(ctx, hipEventCreate(&ctx->prof.stop[i]));
        }
        ctx->prof.created = true;
    }
    if (!on) SLAM_TRY(prof_flush(ctx));
    ctx->profiling = on != 0;
    return SLAM_OK;
}

extern "C" int slam_profile_read(slam_ctx* ctx, int kernel, double* total_ms, int64_t* launches) {
    if (!ctx || kernel < 0 || kernel >= SLAM_K_COUNT) return SLAM_EINVAL;
    SLAM_TRY(prof_flush(ctx));
    if (total_ms) *total_ms = ctx->prof.total_ms[kernel];
    if (launches) *launches = ctx->prof.launches[kernel];
    return SLAM_OK;
}

extern "C" int slam_profile_reset(slam_ctx* ctx) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_TRY(prof_flush(ctx));
    for (int k = 0; k < SLAM_K_COUNT; ++k) { ctx->prof.total_ms[k] = 0; ctx->prof.launches[k] = 0; }
    return SLAM_OK;
}

// ================================================================================================
// sampler support: MT19937 output words after CPython's random.seed(int)
//   (CPython Modules/_randommodule.c: init_genrand(19650218) then init_by_array(key), key = 32-bit words of the
//    seed, least significant first; outputs are the tempered words of the first state regeneration)
// ================================================================================================
namespace {

struct Mt19937 {
    static const int N = 624, M = 397;
    uint32_t mt[N];
    void seed_by_array(const uint32_t* key, int klen) {
        mt[0] = 19650218u;
        for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        int i = 1, j = 0;
        for (int k = std::max(N, klen); k > 0; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = N - 1; k > 0; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    // first `count` (<= N - M) outputs: word k of the regenerated state depends on old words k, k+1, k+M only
    void first_outputs(uint32_t* out, int count) const {
        for (int k = 0; k < count; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            y = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[k] = y;
        }
    }
};
static_assert(SLAM_RAW_WORDS <= 624 - 397, "table row must come from the first regeneration block");

}  // namespace

// make ctx->raw cover iterations [it0, it0+n) of `seed`
int slam_fill_raw_table(slam_ctx* ctx, uint64_t seed, int64_t it0, int64_t n) {
    if (ctx->raw_n > 0 && ctx->raw_seed == seed && it0 >= ctx->raw_it0 && it0 + n <= ctx->raw_it0 + ctx->raw_n)
        return SLAM_OK;
    // regenerate, keeping the start when the request extends an existing table and rounding the size up
    int64_t new_it0 = it0, new_n = n;
    if (ctx->raw_n > 0 && ctx->raw_seed == seed && it0 >= ctx->raw_it0 && it0 <= ctx->raw_it0 + ctx->raw_n) {
        new_it0 = ctx->raw_it0;
        new_n = it0 + n - new_it0;
    }
    new_n = std::max<int64_t>(new_n + new_n / 2, 1024);
    std::vector<uint32_t> host((size_t)new_n * SLAM_RAW_WORDS);
    unsigned nthreads = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (new_n < 4096) nthreads = 1;
    auto work = [&](int64_t lo, int64_t hi) {
        Mt19937 g;
        for (int64_t r = lo; r < hi; ++r) {
            const uint64_t s = seed + (uint64_t)(new_it0 + r);
            const uint32_t key[2] = {(uint32_t)s, (uint32_t)(s >> 32)};
            g.seed_by_array(key, key[1] ? 2 : 1);
            g.first_outputs(host.data() + (size_t)r * SLAM_RAW_WORDS, SLAM_RAW_WORDS);
        }
    };
    if (nthreads == 1) {
        work(0, new_n);
    } else {
        std::vector<std::thread> pool;
        const int64_t per = (new_n + nthreads - 1) / nthreads;
        for (unsigned t = 0; t < nthreads; ++t) {
            const int64_t lo = t * per, hi = std::min<int64_t>(new_n, lo + per);
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    if (ctx->inject_sampler_fault) std::fill(host.begin(), host.end(), 0xFFFFFFFFu);  // every word rejected by every N
    ctx->raw_n = 0;
    // Kernels queued earlier on ANY of the context's streams index the table with the old first iteration: they must
    // have finished before the rows are replaced in place (growing the buffer synchronises too, a refill that fits
    // the capacity would not).
    SLAM_TRY(slam_sync_all(ctx));
    SLAM_TRY(slam_reserve(ctx, ctx->raw, host.size() * sizeof(uint32_t)));
    SLAM_HIP(ctx, hipMemcpyAsync(ctx->raw.p, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                                 ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->raw_seed = seed;
    ctx->raw_it0 = new_it0;
    ctx->raw_n = new_n;
    return SLAM_OK;
}

extern "C" int64_t slam_num_hypotheses(int64_t n_points, int64_t iterations) {
    if (n_points < 8 || iterations < 0) return 0;
    // C(n, 8), saturating: stop as soon as it exceeds `iterations`
    unsigned __int128 c = 1;
    for (int i = 1; i <= 8; ++i) {
        c = c * (unsigned __int128)(n_points - 8 + i) / (unsigned)i;  // exact: c is C(n-8+i, i) after each step
        if (c > (unsigned __int128)iterations) return iterations;     // C(n-8+i, i) only grows with i
    }
    return (int64_t)c;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int check_flags(slam_ctx* ctx) {
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (callers count on this: their downloads are complete afterwards)
    if (__atomic_load_n(&ctx->h_flags[0], __ATOMIC_ACQUIRE) & 1) {
        __atomic_store_n(&ctx->h_flags[0], 0, __ATOMIC_RELEASE);
        snprintf(ctx->err, sizeof ctx->err, "sampler ran out of random words for at least one iteration");
        return SLAM_ESAMPLER;
    }
    return SLAM_OK;
}

extern "C" int slam_sample_indices(slam_ctx* ctx, int kind, uint64_t seed, int64_t it0, int H, int N, int k,
                                   int32_t* idx) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, idx != nullptr, "idx is NULL");
    SLAM_REQUIRE(ctx, H >= 1 && it0 >= 0, "H < 1 or it0 < 0");
    SLAM_REQUIRE(ctx, k >= 1 && k <= SLAM_MAX_K && N >= k, "need 1 <= k <= 16 and N >= k");
    SLAM_REQUIRE(ctx, kind == SLAM_SAMPLER_CPYTHON || kind == SLAM_SAMPLER_COUNTER, "unknown sampler");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    if (kind == SLAM_SAMPLER_CPYTHON) SLAM_TRY(slam_fill_raw_table(ctx, seed, it0, H));
    SLAM_TRY(slam_reserve(ctx, ctx->idx, (size_t)H * k * sizeof(int32_t)));
    SLAM_TRY(slam_launch(ctx, -1, k_sample, dim3((H + 63) / 64), dim3(64), 0, kind, seed, it0, H, N, k,
                         (const uint32_t*)ctx->raw.p, ctx->raw_it0, (int32_t*)ctx->idx.p, (int32_t*)ctx->flags.p));
    SLAM_HIP(ctx, hipMemcpyAsync(idx, ctx->idx.p, (size_t)H * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return check_flags(ctx);
}

// ================================================================================================
// RANSAC
// ================================================================================================
// Smallest double c >= 0 with fl(c*c) >= thr: then fl(r*r) < thr  <=>  |r| < c for every double r, because the
// correctly rounded square is a non-decreasing function of |r|.  thr <= 0 gives 0 (nothing is below it).
static double square_cutoff(double thr) {
    if (!(thr > 0.0)) return 0.0;
    if (std::isinf(thr)) return thr;
    double c = std::sqrt(thr);
    while (c * c >= thr && c > 0.0) c = std::nextafter(c, 0.0);              // now fl(c*c) < thr
    while (c * c < thr) c = std::nextafter(c, std::numeric_limits<double>::infinity());
    return c;
}

// Workgroup shape (waves, matches per lane and chunk) is a function of N only, so that a hypothesis' errsum never
// depends on H or on how the hypotheses are spread over kernels, workgroups and GPUs.
//   8 waves; 1, 4 or 8 matches per lane and chunk for N <= 512, <= 2048, larger
#define SLAM_HB 8
#define SLAM_DISPATCH_SHAPE(N, CALL)             \
    do {                                         \
        if ((N) <= 512) { CALL(1, 8); }          \
        else if ((N) <= 2048) { CALL(4, 8); }    \
        else { CALL(8, 8); }                     \
    } while (0)

// small H: one fused launch
template <int SAMPLER, bool SAMPSON>
static int launch_fused(slam_ctx* ctx, const double2* p0, const double2* p1, int N, const int32_t* d_idx, uint64_t seed,
                        int64_t it0, int H, double thr, double* F, int32_t* counts, double* errsums,
                        const CarriedSelect& carried = CarriedSelect{}) {
    const uint32_t* raw = (const uint32_t*)ctx->raw.p;
    int32_t* flags = (int32_t*)ctx->flags.p;
    // Eight hypotheses per workgroup; four when eight would leave half of the CUs without a workgroup (the reference's
    // default of 1000 iterations: 250 workgroups instead of 125, 6.4 instead of 7.5 us at N=2048).  A hypothesis'
    // numbers do not depend on the choice.
    if ((H + SLAM_HB - 1) / SLAM_HB <= ctx->num_cus / 2) {
        const dim3 grid((H + 3) / 4 + carried.n_wgs);
#define SLAM_CALL(PPL, WAVES)                                                                                         \
    return slam_launch(ctx, SLAM_K_SCORE, k_ransac<4, PPL, WAVES, SAMPLER, SAMPSON>, grid, dim3(64 * WAVES), 0, raw, ctx->raw_it0, \
                       it0, H, N, p0, p1, d_idx, seed, thr, F, counts, errsums, flags, (const PairArgs*)nullptr, carried)
        SLAM_DISPATCH_SHAPE(N, SLAM_CALL);
#undef SLAM_CALL
    }
    // Sixteen per workgroup with several calls in flight (large shape): a workgroup then spends two thirds of its life
    // scoring instead of half of it, and of the two workgroups a CU holds (of two different calls) one is scoring more
    // often.  A call on its own is slower that way (half the CUs, twice the scoring per workgroup).
    // (the reference's residual only: with the Sampson distance this shape does not fit 128 registers)
    if constexpr (!SAMPSON)
    if (ctx->wide_workgroups && ctx->lanes_in_use > 1 && N > 2048 && (H + 15) / 16 >= ctx->num_cus / 2) {
        const dim3 grid((H + 15) / 16 + carried.n_wgs);
        return slam_launch(ctx, SLAM_K_SCORE, k_ransac<16, 8, 8, SAMPLER, SAMPSON>, grid, dim3(512), 0, raw, ctx->raw_it0,
                           it0, H, N, p0, p1, d_idx, seed, thr, F, counts, errsums, flags, (const PairArgs*)nullptr, carried);
    }
    const dim3 grid((H + SLAM_HB - 1) / SLAM_HB + carried.n_wgs);
#define SLAM_CALL(PPL, WAVES)                                                                                         \
    return slam_launch(ctx, SLAM_K_SCORE, k_ransac<SLAM_HB, PPL, WAVES, SAMPLER, SAMPSON>, grid, dim3(64 * WAVES), 0, raw, ctx->raw_it0, \
                       it0, H, N, p0, p1, d_idx, seed, thr, F, counts, errsums, flags, (const PairArgs*)nullptr, carried)
    SLAM_DISPATCH_SHAPE(N, SLAM_CALL);
#undef SLAM_CALL
}

// large H: a lane per hypothesis for the solve, then the scoring kernel
// The scoring kernel keeps one chunk of matches in LDS per workgroup (one workgroup per CU) and gives every wave
// hypotheses of its own.  With several chunks (N > 512 * matches-per-lane) it writes per-chunk partial results, which
// a small kernel folds in chunk order; the partial tables are bounded by scoring the hypotheses in tiles.
#define SLAM_SCORE_HG 4       // hypotheses a wave scores at a time
#define SLAM_SCORE_WAVES 16   // waves per scoring workgroup
#define SLAM_PARTIAL_BYTES ((size_t)96 << 20)  // upper bound of the per-chunk partial tables of one tile

template <int SAMPLER, bool SAMPSON>
static int launch_split(slam_ctx* ctx, const double2* p0, const double2* p1, int N, const int32_t* d_idx, uint64_t seed,
                        int64_t it0, int H, double thr, double* F, int32_t* counts, double* errsums) {
    const int ppl = N <= 512 ? 1 : (N <= 2048 ? 4 : 8);  // the same limits as SLAM_DISPATCH_SHAPE
    const int chunk = 512 * ppl;
    const int nchunks = (N + chunk - 1) / chunk;
    int lane = 0;  // calls in flight on different internal streams must not share the partial tables
    for (int l = 1; l < SLAM_LANES; ++l)
        if (ctx->cur == ctx->lane_stream[l]) lane = l;
    int tile = H;
    if (nchunks > 1) {
        const size_t per_h = (size_t)nchunks * (sizeof(double) + sizeof(int32_t));
        tile = (int)std::min<size_t>((size_t)H, std::max<size_t>(SLAM_SCORE_HG, SLAM_PARTIAL_BYTES / per_h / SLAM_SCORE_HG * SLAM_SCORE_HG));
        SLAM_TRY(slam_reserve(ctx, ctx->score_parts[lane], (size_t)tile * per_h + 256));
    }
    SLAM_TRY(slam_launch(ctx, SLAM_K_SOLVE, k_solve<SAMPLER>, dim3((H + 63) / 64), dim3(64), 0, p0, p1, N, d_idx, seed, it0,
                         H, (const uint32_t*)ctx->raw.p, ctx->raw_it0, F, (int32_t*)ctx->flags.p));
    for (int t0 = 0; t0 < H; t0 += tile) {
        const int Ht = std::min(tile, H - t0);
        const int ngroups = (Ht + SLAM_SCORE_HG - 1) / SLAM_SCORE_HG;
        const int want_slots = (ngroups + SLAM_SCORE_WAVES - 1) / SLAM_SCORE_WAVES;
        const int wgs = ctx->score_workgroups > 0 ? ctx->score_workgroups : ctx->num_cus;
        const int nslots = std::max(1, std::min(wgs / std::max(1, std::min(nchunks, wgs)), want_slots));
        const dim3 grid((unsigned)std::min<int64_t>((int64_t)nchunks * nslots, wgs));
        double* pe = errsums + t0;
        int32_t* pc = counts + t0;
        if (nchunks > 1) {
            pe = (double*)ctx->score_parts[lane].p;
            pc = (int32_t*)((char*)ctx->score_parts[lane].p + (size_t)nchunks * Ht * sizeof(double));
        }
#define SLAM_CALL(PPL, WAVES)                                                                                          \
    SLAM_TRY(slam_launch(ctx, SLAM_K_SCORE, k_score_lds<SLAM_SCORE_HG, PPL, SLAM_SCORE_WAVES, SAMPSON>, grid,          \
                         dim3(64 * SLAM_SCORE_WAVES), 0, p0, p1, N, (const double*)(F + (size_t)t0 * 9), Ht, thr, nchunks,   \
                         nslots, pc, pe))
        SLAM_DISPATCH_SHAPE(N, SLAM_CALL);
#undef SLAM_CALL
        if (nchunks > 1)
            SLAM_TRY(slam_launch(ctx, -1, k_combine_chunks, dim3((Ht + 255) / 256), dim3(256), 0, (const int32_t*)pc,
                                 (const double*)pe, nchunks, Ht, counts + t0, errsums + t0));
    }
    return SLAM_OK;
}

#define SLAM_FUSED_MAX_H 4096
static constexpr int SLAM_MAX_MATCHES = 1 << 27;

// workgroups of the selection kernel: one per SLAM_SELECT_THREADS matches when it also writes the winner's mask (at most 64)
static dim3 select_grid(int N, const void* mask) {
    return dim3(mask ? (unsigned)std::min((N + SLAM_SELECT_THREADS - 1) / SLAM_SELECT_THREADS, 64) : 1u);
}

// Large H without per-hypothesis error sums: the counts in float32 with a float64 re-check band (kernels_count32.h),
// then the float64 error sums of the hypotheses that tie on the largest count, then the selection among those.
// Layout of the call's work area (ctx->score_parts[lane]): float32 table, state words, tie list and tie tables.
struct F32Area {
    float* F32;
    CountState* state;
    int32_t* n_out;
    int32_t* tie_index;
    int32_t* tie_counts;
    double* tie_errsums;
    double* partial;
    int cap;
};
#define SLAM_F32_MIN_EVALS ((int64_t)1 << 27)  // below this many (hypothesis, match) pairs the extra launches cost more than they save

static int f32_area(slam_ctx* ctx, int lane, int H, int nchunks, F32Area* a) {
    // ties one rescoring launch handles (its per-chunk sums take cap * nchunks doubles: 16 MiB at most)
    const int cap = (int)std::min<int64_t>(H, std::max<int64_t>(64, ((int64_t)1 << 21) / nchunks));
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align256(bytes); return at; };
    const size_t o_f32 = take((size_t)H * SLAM_F32_ROW * sizeof(float)), o_state = take(sizeof(CountState)), o_n = take(sizeof(int32_t)),
                 o_ti = take((size_t)H * 4), o_tc = take((size_t)H * 4), o_te = take((size_t)H * 8),
                 o_part = take((size_t)cap * nchunks * 8);
    SLAM_TRY(slam_reserve(ctx, ctx->score_parts[lane], off));
    char* b = (char*)ctx->score_parts[lane].p;
    a->F32 = (float*)(b + o_f32); a->state = (CountState*)(b + o_state); a->n_out = (int32_t*)(b + o_n);
    a->tie_index = (int32_t*)(b + o_ti); a->tie_counts = (int32_t*)(b + o_tc); a->tie_errsums = (double*)(b + o_te);
    a->partial = (double*)(b + o_part); a->cap = cap;
    return SLAM_OK;
}

#ifndef SLAM_COUNT_HG
#define SLAM_COUNT_HG 4   // hypotheses a wave counts at a time
#endif
template <int SAMPLER>
static int launch_count_f32(slam_ctx* ctx, const double2* p0, const double2* p1, int N, const int32_t* d_idx, uint64_t seed,
                            int64_t it0, int H, double rcut, double* F, int32_t* counts, double* d_record, uint8_t* d_mask) {
    const int ppl = N <= 2048 ? 4 : 8;  // the same limits as SLAM_DISPATCH_SHAPE (N <= 512 does not come here)
    const int chunk = 512 * ppl;
    const int nchunks = (N + chunk - 1) / chunk;
    int lane = 0;
    for (int l = 1; l < SLAM_LANES; ++l)
        if (ctx->cur == ctx->lane_stream[l]) lane = l;
    F32Area a;
    SLAM_TRY(f32_area(ctx, lane, H, nchunks, &a));
    SLAM_HIP(ctx, hipMemsetAsync(a.state, 0, sizeof(CountState), ctx->cur));
    if (nchunks > 1) SLAM_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)H * sizeof(int32_t), ctx->cur));
    SLAM_TRY(slam_launch(ctx, SLAM_K_SOLVE, k_solve_f32<SAMPLER>, dim3((H + 63) / 64), dim3(64), 0, p0, p1, N, d_idx, seed, it0,
                         H, (const uint32_t*)ctx->raw.p, ctx->raw_it0, F, a.F32, (int32_t*)ctx->flags.p));
    const int ngroups = (H + SLAM_COUNT_HG - 1) / SLAM_COUNT_HG;
    const int want_slots = (ngroups + SLAM_SCORE_WAVES - 1) / SLAM_SCORE_WAVES;
    const int wgs = ctx->score_workgroups > 0 ? ctx->score_workgroups : ctx->num_cus;
    const int nslots = std::max(1, std::min(wgs / std::max(1, std::min(nchunks, wgs)), want_slots));
    const dim3 grid((unsigned)std::min<int64_t>((int64_t)nchunks * nslots, wgs));
    if (ppl == 4)
        SLAM_TRY(slam_launch(ctx, SLAM_K_SCORE, k_count_f32<SLAM_COUNT_HG, 4, SLAM_SCORE_WAVES>, grid, dim3(64 * SLAM_SCORE_WAVES), 0,
                             p0, p1, N, (const float*)a.F32, (const double*)F, H, rcut, nchunks, nslots, counts, nchunks > 1 ? 1 : 0, a.state));
    else
        SLAM_TRY(slam_launch(ctx, SLAM_K_SCORE, k_count_f32<SLAM_COUNT_HG, 8, SLAM_SCORE_WAVES>, grid, dim3(64 * SLAM_SCORE_WAVES), 0,
                             p0, p1, N, (const float*)a.F32, (const double*)F, H, rcut, nchunks, nslots, counts, nchunks > 1 ? 1 : 0, a.state));
    if (ctx->report_f32_band) {  // SLAM_AMD_REPORT_F32_BAND=1: statistics of the re-check band on stderr (diagnostic; synchronises)
        CountState hs;
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->cur));
        SLAM_HIP(ctx, hipMemcpy(&hs, a.state, sizeof hs, hipMemcpyDeviceToHost));
        const double blocks = (double)H / SLAM_COUNT_HG * (N / 128.0);
        fprintf(stderr, "f32 counts: N=%d H=%d max_count=%u band blocks=%u of %.0f (%.3f %%)\n", N, H, hs.max_count, hs.band_evals,
                blocks, 100.0 * hs.band_evals / blocks);
    }
    if (nchunks > 1)
        SLAM_TRY(slam_launch(ctx, -1, k_max_count, dim3((H + 8191) / 8192), dim3(256), 0, (const int32_t*)counts, H, a.state));
    SLAM_TRY(slam_launch(ctx, -1, k_collect_ties, dim3((H + 255) / 256), dim3(256), 0, (const int32_t*)counts, H, a.state,
                         a.tie_index));
    // Usually one hypothesis has the largest count and the first launch pair does all there is; every hypothesis may
    // tie, though (degenerate data), so the launches cover all H list positions, a.cap at a time.
    const dim3 rgrid((unsigned)nchunks, (unsigned)std::min(a.cap, 64));
    for (int t_base = 0; t_base < H; t_base += a.cap) {
        if (ppl == 4)
            SLAM_TRY(slam_launch(ctx, -1, k_rescore_ties<4>, rgrid, dim3(512), 0, p0, p1, N, (const double*)F, rcut,
                                 (const CountState*)a.state, (const int32_t*)a.tie_index, t_base, a.cap, a.partial));
        else
            SLAM_TRY(slam_launch(ctx, -1, k_rescore_ties<8>, rgrid, dim3(512), 0, p0, p1, N, (const double*)F, rcut,
                                 (const CountState*)a.state, (const int32_t*)a.tie_index, t_base, a.cap, a.partial));
        SLAM_TRY(slam_launch(ctx, -1, k_fold_ties, dim3((a.cap + 255) / 256), dim3(256), 0, (const CountState*)a.state, t_base, a.cap,
                             nchunks, (const double*)a.partial, a.tie_counts, a.tie_errsums, a.n_out));
    }
    SLAM_TRY(slam_launch(ctx, SLAM_K_SELECT, k_select_ties, select_grid(N, d_mask), dim3(SLAM_SELECT_THREADS), 0,
                         (const int32_t*)a.tie_counts, (const double*)a.tie_errsums, (const int32_t*)a.tie_index,
                         (const int32_t*)a.n_out, (const double*)F, it0, p0, p1, N, rcut, d_record, d_mask));
    return SLAM_OK;
}

template <int SAMPLER>
static int launch_ransac(slam_ctx* ctx, const double2* p0, const double2* p1, int N, const int32_t* d_idx, uint64_t seed,
                         int64_t it0, int H, double thr, double* F, int32_t* counts, double* errsums,
                         bool no_partial_tables) {
    const bool fused = H <= SLAM_FUSED_MAX_H;
    if (!fused && no_partial_tables && N > 512 * 8) {
        // The split path would need per-chunk partial tables (a context buffer that grows on demand).  The fused
        // kernel walks the chunks itself, in the same order: ranges of SLAM_FUSED_MAX_H hypotheses, same bits.
        for (int t0 = 0; t0 < H; t0 += SLAM_FUSED_MAX_H) {
            const int Ht = std::min(SLAM_FUSED_MAX_H, H - t0);
            const int32_t* idx_t = d_idx ? d_idx + (size_t)t0 * 8 : nullptr;
            if (ctx->sampson)
                SLAM_TRY((launch_fused<SAMPLER, true>(ctx, p0, p1, N, idx_t, seed, it0 + t0, Ht, thr, F + (size_t)t0 * 9,
                                                      counts + t0, errsums + t0)));
            else
                SLAM_TRY((launch_fused<SAMPLER, false>(ctx, p0, p1, N, idx_t, seed, it0 + t0, Ht, thr, F + (size_t)t0 * 9,
                                                       counts + t0, errsums + t0)));
        }
        return SLAM_OK;
    }
    if (ctx->sampson)
        return fused ? launch_fused<SAMPLER, true>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, F, counts, errsums)
                     : launch_split<SAMPLER, true>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, F, counts, errsums);
    return fused ? launch_fused<SAMPLER, false>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, F, counts, errsums)
                 : launch_split<SAMPLER, false>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, F, counts, errsums);
}


static int ransac_enqueue(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                          int sampler, uint64_t seed, int64_t it0, int H, double thr, double* d_record,
                          uint8_t* d_mask, int32_t* d_counts, double* d_errsums, double* d_F_all,
                          bool no_alloc = false, int errsums_requested = -1 /* -1: exactly when d_errsums is given */) {
    SLAM_REQUIRE(ctx, d_p0 && d_p1 && d_record, "points or record pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");          // reference: src/epipolar.py:60
    SLAM_REQUIRE(ctx, N <= SLAM_MAX_MATCHES, "more than 2^27 matches");  // the kernels address matches by 32-bit byte offsets
    SLAM_REQUIRE(ctx, H >= 1 && it0 >= 0, "H < 1 or it0 < 0");
    SLAM_REQUIRE(ctx, d_idx || sampler == SLAM_SAMPLER_CPYTHON || sampler == SLAM_SAMPLER_COUNTER, "unknown sampler");
    SLAM_REQUIRE(ctx, !d_idx || ((uintptr_t)d_idx & 15) == 0, "idx must be 16-byte aligned");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    // Nobody reads the per-hypothesis error sums, the call is long and scores the reference's residual: counts in
    // float32 with a float64 re-check band, error sums for the hypotheses that tie on the largest count only.
    const bool want_errsums = errsums_requested < 0 ? d_errsums != nullptr : errsums_requested != 0;
    const bool f32_counts = ctx->f32_counts && !want_errsums && !no_alloc && !ctx->sampson && H > SLAM_FUSED_MAX_H && N > 512 &&
                            (int64_t)H * N >= SLAM_F32_MIN_EVALS;
    if (!d_F_all) { SLAM_TRY(slam_reserve(ctx, ctx->F, (size_t)H * 9 * sizeof(double))); d_F_all = (double*)ctx->F.p; }
    if (!d_counts) { SLAM_TRY(slam_reserve(ctx, ctx->counts, (size_t)H * sizeof(int32_t))); d_counts = (int32_t*)ctx->counts.p; }
    if (!d_errsums && !f32_counts) { SLAM_TRY(slam_reserve(ctx, ctx->errsums, (size_t)H * sizeof(double))); d_errsums = (double*)ctx->errsums.p; }
    const double2* p0 = (const double2*)d_p0;
    const double2* p1 = (const double2*)d_p1;
    if (!ctx->sampson) thr = square_cutoff(thr);  // algebraic mode: from here on the threshold is the cutoff on |r|
    if (f32_counts) {
        if (d_idx) return launch_count_f32<-1>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, d_F_all, d_counts, d_record, d_mask);
        if (sampler == SLAM_SAMPLER_CPYTHON) {
            SLAM_TRY(slam_fill_raw_table(ctx, seed, it0, H));
            return launch_count_f32<SLAM_SAMPLER_CPYTHON>(ctx, p0, p1, N, nullptr, seed, it0, H, thr, d_F_all, d_counts, d_record, d_mask);
        }
        return launch_count_f32<SLAM_SAMPLER_COUNTER>(ctx, p0, p1, N, nullptr, seed, it0, H, thr, d_F_all, d_counts, d_record, d_mask);
    }
    if (d_idx) {
        SLAM_TRY(launch_ransac<-1>(ctx, p0, p1, N, d_idx, seed, it0, H, thr, d_F_all, d_counts, d_errsums, no_alloc));
    } else if (sampler == SLAM_SAMPLER_CPYTHON) {
        if (!no_alloc) SLAM_TRY(slam_fill_raw_table(ctx, seed, it0, H));  // (the caller has checked that the table covers the range)
        SLAM_TRY(launch_ransac<SLAM_SAMPLER_CPYTHON>(ctx, p0, p1, N, nullptr, seed, it0, H, thr, d_F_all, d_counts, d_errsums, no_alloc));
    } else {
        SLAM_TRY(launch_ransac<SLAM_SAMPLER_COUNTER>(ctx, p0, p1, N, nullptr, seed, it0, H, thr, d_F_all, d_counts, d_errsums, no_alloc));
    }
    // large H: a first stage reduces ranges of SLAM_SELECT_SPAN hypotheses in parallel
    const int32_t* sel_counts = d_counts;
    const double* sel_errsums = d_errsums;
    const int32_t* sel_index = nullptr;
    int sel_H = H;
    if (H > 4 * SLAM_SELECT_SPAN) {
        const int parts = (H + SLAM_SELECT_SPAN - 1) / SLAM_SELECT_SPAN;
        int lane = 0;  // calls in flight on different internal streams must not share the table
        for (int l = 1; l < SLAM_LANES; ++l)
            if (ctx->cur == ctx->lane_stream[l]) lane = l;
        slam_buf& table = ctx->select_parts[lane];
        SLAM_TRY(slam_reserve(ctx, table, (size_t)parts * 16));
        int32_t* pc = (int32_t*)table.p;
        int32_t* pi = pc + parts;
        double* pe = (double*)((char*)table.p + (size_t)parts * 8);
        SLAM_TRY(slam_launch(ctx, SLAM_K_SELECT, k_select_partial, dim3(parts), dim3(1024), 0, (const int32_t*)d_counts,
                             (const double*)d_errsums, H, pc, pe, pi));
        sel_counts = pc; sel_errsums = pe; sel_index = pi; sel_H = parts;
    }
    if (ctx->sampson)
        SLAM_TRY(slam_launch(ctx, SLAM_K_SELECT, k_select<true>, select_grid(N, d_mask), dim3(SLAM_SELECT_THREADS), 0, sel_counts, sel_errsums, sel_index,
                             (const double*)d_F_all, sel_H, it0, p0, p1, N, thr, d_record, d_mask, (const PairArgs*)nullptr));
    else
        SLAM_TRY(slam_launch(ctx, SLAM_K_SELECT, k_select<false>, select_grid(N, d_mask), dim3(SLAM_SELECT_THREADS), 0, sel_counts, sel_errsums, sel_index,
                             (const double*)d_F_all, sel_H, it0, p0, p1, N, thr, d_record, d_mask, (const PairArgs*)nullptr));
    return SLAM_OK;
}

// ---- selections carried by the next call's kernel (several calls in flight) ------------------------------------------
// the selection of lane l that waits to be carried, launched on its own (on that lane's stream, behind the fused kernel
// that produced its hypotheses)
static int flush_lane(slam_ctx* ctx, int l) {
    slam_carried_select& c = ctx->pending[l];
    if (c.n_wgs == 0) return SLAM_OK;
    hipStream_t keep = ctx->cur;
    ctx->cur = ctx->lane_stream[l];
    ctx->lanes_dirty |= 1u << l;
    int rc;
    if (c.sampson)
        rc = slam_launch(ctx, SLAM_K_SELECT, k_select<true>, select_grid(c.N, c.mask), dim3(SLAM_SELECT_THREADS), 0, c.counts,
                         c.errsums, (const int32_t*)nullptr, c.Fall, c.H, c.it0, c.p0, c.p1, c.N, c.rcut, c.record, c.mask,
                         (const PairArgs*)nullptr);
    else
        rc = slam_launch(ctx, SLAM_K_SELECT, k_select<false>, select_grid(c.N, c.mask), dim3(SLAM_SELECT_THREADS), 0, c.counts,
                         c.errsums, (const int32_t*)nullptr, c.Fall, c.H, c.it0, c.p0, c.p1, c.N, c.rcut, c.record, c.mask,
                         (const PairArgs*)nullptr);
    ctx->cur = keep;
    c.n_wgs = 0;  // (also after a failed launch: it is not tried again)
    return rc;
}

int slam_flush_pending(slam_ctx* ctx) {
    int rc = SLAM_OK;
    for (int l = 0; l < SLAM_LANES; ++l) {
        const int r = flush_lane(ctx, l);
        if (rc == SLAM_OK) rc = r;
    }
    return rc;
}

// A call in concurrency mode whose hypothesis scratch is the library's own: its fused kernel carries the selection of
// the previous call on this lane, and its own selection waits for the next call (or for slam_synchronize).
static int ransac_carry(slam_ctx* ctx, int lane, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                        int sampler, uint64_t seed, int64_t it0, int H, double thr, double* d_record, uint8_t* d_mask) {
    SLAM_REQUIRE(ctx, d_p0 && d_p1 && d_record, "points or record pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");
    SLAM_REQUIRE(ctx, N <= SLAM_MAX_MATCHES, "more than 2^27 matches");
    SLAM_REQUIRE(ctx, H >= 1 && H <= SLAM_FUSED_MAX_H && it0 >= 0, "H < 1 or it0 < 0");
    SLAM_REQUIRE(ctx, d_idx || sampler == SLAM_SAMPLER_CPYTHON || sampler == SLAM_SAMPLER_COUNTER, "unknown sampler");
    SLAM_REQUIRE(ctx, !d_idx || ((uintptr_t)d_idx & 15) == 0, "idx must be 16-byte aligned");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t f_bytes = align256((size_t)H * 9 * sizeof(double)), e_bytes = align256((size_t)H * sizeof(double)),
                 c_bytes = align256((size_t)H * sizeof(int32_t));
    const size_t need = f_bytes + e_bytes + c_bytes;
    if (need > ctx->carry_set_bytes[lane]) {
        // the two sets move: the selection that still reads the old ones goes first (the stream orders the rest)
        SLAM_TRY(flush_lane(ctx, lane));
        SLAM_TRY(slam_reserve(ctx, ctx->carry_scratch[lane], 2 * need));
        ctx->carry_set_bytes[lane] = need;
    }
    if (!d_idx && sampler == SLAM_SAMPLER_CPYTHON) SLAM_TRY(slam_fill_raw_table(ctx, seed, it0, H));
    char* set = (char*)ctx->carry_scratch[lane].p + (size_t)(ctx->carry_turn[lane]++ & 1u) * ctx->carry_set_bytes[lane];
    double* F = (double*)set;
    double* errsums = (double*)(set + f_bytes);
    int32_t* counts = (int32_t*)(set + f_bytes + e_bytes);
    const double2* p0 = (const double2*)d_p0;
    const double2* p1 = (const double2*)d_p1;
    const double cut = ctx->sampson ? thr : square_cutoff(thr);
    const CarriedSelect carried = ctx->pending[lane];  // n_wgs == 0: nothing to carry
    ctx->cur = ctx->lane_stream[lane];
    ctx->lanes_dirty |= 1u << lane;
    int rc;
#define SLAM_CARRY(SAMPLER_ID, IDX)                                                                                        \
    rc = ctx->sampson ? launch_fused<SAMPLER_ID, true>(ctx, p0, p1, N, IDX, seed, it0, H, cut, F, counts, errsums, carried) \
                      : launch_fused<SAMPLER_ID, false>(ctx, p0, p1, N, IDX, seed, it0, H, cut, F, counts, errsums, carried)
    if (d_idx) { SLAM_CARRY(-1, d_idx); }
    else if (sampler == SLAM_SAMPLER_CPYTHON) { SLAM_CARRY(SLAM_SAMPLER_CPYTHON, nullptr); }
    else { SLAM_CARRY(SLAM_SAMPLER_COUNTER, nullptr); }
#undef SLAM_CARRY
    ctx->cur = ctx->stream;
    if (rc != SLAM_OK) return rc;
    slam_carried_select& next = ctx->pending[lane];
    next.counts = counts; next.errsums = errsums; next.Fall = F; next.p0 = p0; next.p1 = p1; next.record = d_record;
    next.mask = d_mask; next.it0 = it0; next.rcut = cut; next.H = H; next.N = N; next.sampson = ctx->sampson ? 1 : 0;
    next.n_wgs = d_mask ? std::min((N + 511) / 512, 64) : 1;
    return SLAM_OK;
}

extern "C" int slam_ransac_f_dev(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                                 int sampler, uint64_t seed, int64_t it0, int H, double threshold, double* d_record,
                                 uint8_t* d_mask, int32_t* d_counts, double* d_errsums, double* d_F_all) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, H >= 1, "H < 1");
    // With slam_set_concurrency(n > 1) independent calls rotate over the context's internal streams (each with its
    // own hypothesis scratch) so that one call's selection kernel and the next call's solve phase overlap.  With a
    // caller-supplied stream, or n = 1 (the default), every call stays on one stream, in order.
    int lane = 0;
    if (ctx->stream == ctx->own_stream && ctx->lanes_in_use > 1) lane = (int)(ctx->next_lane++ % (unsigned)ctx->lanes_in_use);
    // (the fused kernel of the large shape, N > 2048, is the one compiled with the carrying role)
    if (ctx->stream == ctx->own_stream && ctx->lanes_in_use > 1 && H <= SLAM_FUSED_MAX_H && N > 2048 && !d_counts && !d_errsums &&
        !d_F_all && !ctx->profiling && ctx->carry_selection)
        return ransac_carry(ctx, lane, d_p0, d_p1, N, d_idx, sampler, seed, it0, H, threshold, d_record, d_mask);
    SLAM_TRY(flush_lane(ctx, lane));  // a selection that waits on this lane goes before this call's kernels
    if (lane > 0) {
        if (sampler == SLAM_SAMPLER_CPYTHON && !d_idx) SLAM_TRY(slam_fill_raw_table(ctx, seed, it0, H));  // on the ordered stream
        if (!d_F_all) { SLAM_TRY(slam_reserve(ctx, ctx->lane_F[lane], (size_t)H * 9 * sizeof(double))); d_F_all = (double*)ctx->lane_F[lane].p; }
        if (!d_counts) { SLAM_TRY(slam_reserve(ctx, ctx->lane_counts[lane], (size_t)H * sizeof(int32_t))); d_counts = (int32_t*)ctx->lane_counts[lane].p; }
    }
    double* const caller_errsums = d_errsums;
    if (lane > 0 && !d_errsums) { SLAM_TRY(slam_reserve(ctx, ctx->lane_errsums[lane], (size_t)H * sizeof(double))); d_errsums = (double*)ctx->lane_errsums[lane].p; }
    ctx->cur = ctx->lane_stream[lane] && ctx->stream == ctx->own_stream ? ctx->lane_stream[lane] : ctx->stream;
    if (ctx->cur != ctx->stream) ctx->lanes_dirty |= 1u << lane;
    const int rc = ransac_enqueue(ctx, d_p0, d_p1, N, d_idx, sampler, seed, it0, H, threshold, d_record, d_mask, d_counts,
                                  d_errsums, d_F_all, false, caller_errsums != nullptr);
    ctx->cur = ctx->stream;
    return rc;
}

// The same call for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid it): the
// per-hypothesis scratch lives in the code object's `scratch` array instead of context buffers that grow on demand.
// No hipMalloc / hipFree, no host-side table generation, nothing synchronises; always on the context's ordered stream.
extern "C" int slam_ransac_f_dev_noalloc(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N,
                                         const int32_t* d_idx, int sampler, uint64_t seed, int64_t it0, int H,
                                         double threshold, double* d_record, uint8_t* d_mask) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, H >= 1, "H < 1");
    if (H > SLAM_NOALLOC_MAX_H) {
        snprintf(ctx->err, sizeof ctx->err,
                 "invalid argument: slam_ransac_f_dev_noalloc scores at most %d hypotheses per call (H = %d); split the "
                 "range into several calls or use slam_ransac_f_dev",
                 SLAM_NOALLOC_MAX_H, H);
        return SLAM_EINVAL;
    }
    SLAM_REQUIRE(ctx, ctx->noalloc_scratch != nullptr, "the context has no scratch pointer");
    if (!d_idx && sampler == SLAM_SAMPLER_CPYTHON) {
        // Under stream capture the launches would bake the word table's address and first iteration into the graph,
        // and any later call with another seed or range rewrites that table in place or frees and reallocates it: a
        // replay would then draw other samples, or read freed memory.  Refused; pre-drawn indices and the
        // counter-based sampler depend on nothing that a later call can change.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        SLAM_HIP(ctx, hipStreamIsCapturing(ctx->stream, &cap));
        SLAM_REQUIRE(ctx, cap == hipStreamCaptureStatusNone,
                     "slam_ransac_f_dev_noalloc under stream capture: the CPython-compatible sampler reads a word table "
                     "that later calls may rewrite or move, so a captured launch cannot be replayed safely; pass "
                     "pre-drawn indices (slam_sample_indices) or use the counter-based sampler");
        // the MT19937 word table is generated on the host and uploaded into a buffer that may have to grow: not here
        const bool covered = ctx->raw_n > 0 && ctx->raw_seed == seed && it0 >= ctx->raw_it0 &&
                             it0 + H <= ctx->raw_it0 + ctx->raw_n;
        SLAM_REQUIRE(ctx, covered,
                     "slam_ransac_f_dev_noalloc: the CPython-compatible sampler needs its word table for this seed and "
                     "iteration range to exist already (made by any earlier call with the same seed that covered the "
                     "range); pass pre-drawn indices or use the counter-based sampler otherwise");
    }
    unsigned char* base = ctx->noalloc_scratch;  // device address from hipGetSymbolAddress, never the array's name
    double* d_F_all = (double*)base;
    double* d_errsums = (double*)(base + SLAM_NOALLOC_F_BYTES);
    int32_t* d_counts = (int32_t*)(base + SLAM_NOALLOC_F_BYTES + SLAM_NOALLOC_E_BYTES);
    // Nothing in the scratch has to start at zero: every launch writes F, count and sum of each of its H hypotheses
    // before the selection kernel reads them, and reads no entry beyond H.
    const bool prof = ctx->profiling;
    ctx->profiling = false;  // the timing helper may synchronise when its event pool wraps
    ctx->cur = ctx->stream;
    const int rc = ransac_enqueue(ctx, d_p0, d_p1, N, d_idx, sampler, seed, it0, H, threshold, d_record, d_mask, d_counts,
                                  d_errsums, d_F_all, /*no_alloc=*/true);
    ctx->profiling = prof;
    return rc;
}

struct slam_job {
    slam_ctx* ctx;
    const double *d_p0, *d_p1;
    int N;
    const int32_t* d_idx;
    int sampler;
    uint64_t seed;
    int64_t it0;
    int H;
    double threshold;
    double* d_record;
    uint8_t* d_mask;
    int32_t* d_counts;
    double* d_errsums;
    double* d_F_all;
};

extern "C" int slam_ransac_job_create(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                                      int sampler, uint64_t seed, int64_t it0, int H, double threshold, double* d_record,
                                      uint8_t* d_mask, int32_t* d_counts, double* d_errsums, double* d_F_all,
                                      slam_job** out) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, out != nullptr, "out is NULL");
    *out = nullptr;
    SLAM_REQUIRE(ctx, d_p0 && d_p1 && d_record, "points or record pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");
    SLAM_REQUIRE(ctx, H >= 1 && it0 >= 0, "H < 1 or it0 < 0");
    SLAM_REQUIRE(ctx, d_idx || sampler == SLAM_SAMPLER_CPYTHON || sampler == SLAM_SAMPLER_COUNTER, "unknown sampler");
    slam_job* j = new (std::nothrow) slam_job{ctx, d_p0, d_p1, N, d_idx, sampler, seed, it0, H, threshold, d_record,
                                              d_mask, d_counts, d_errsums, d_F_all};
    if (!j) return SLAM_ENOMEM;
    *out = j;
    return SLAM_OK;
}

extern "C" int slam_job_run(slam_job* j) {
    if (!j) return SLAM_EINVAL;
    return slam_ransac_f_dev(j->ctx, j->d_p0, j->d_p1, j->N, j->d_idx, j->sampler, j->seed, j->it0, j->H, j->threshold,
                             j->d_record, j->d_mask, j->d_counts, j->d_errsums, j->d_F_all);
}

extern "C" void slam_job_destroy(slam_job* j) { delete j; }

// pinned staging area of at least `bytes` (nullptr above SLAM_STAGE_MAX: such transfers go directly)
#define SLAM_STAGE_MAX ((size_t)32 << 20)
static char* stage_area(slam_ctx* ctx, size_t bytes) {
    if (bytes > SLAM_STAGE_MAX) return nullptr;
    if (bytes > ctx->h_stage_cap) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;  // nothing may still read the old area
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr;
        ctx->h_stage_cap = 0;
        size_t cap = (size_t)1 << 20;
        while (cap < bytes) cap <<= 1;
        if (hipHostMalloc((void**)&ctx->h_stage, cap, hipHostMallocDefault) != hipSuccess) { ctx->h_stage = nullptr; return nullptr; }
        ctx->h_stage_cap = cap;
    }
    return ctx->h_stage;
}

// stage two (N,2) point arrays contiguously in ctx->pts; returns device pointers.  The host entry points that call
// this synchronise before they return, so the staging area is free again at the next call.
static int stage_points(slam_ctx* ctx, const double* p0, const double* p1, int64_t N, double** d0, double** d1) {
    const size_t bytes = (size_t)N * 2 * sizeof(double);
    const size_t second = (bytes + 63) & ~(size_t)63;
    SLAM_TRY(slam_reserve(ctx, ctx->pts, 2 * bytes + 64));
    *d0 = (double*)ctx->pts.p;
    *d1 = (double*)((char*)ctx->pts.p + second);
    if (char* h = stage_area(ctx, second + bytes)) {
        memcpy(h, p0, bytes);
        memcpy(h + second, p1, bytes);
        SLAM_HIP(ctx, hipMemcpyAsync(*d0, h, second + bytes, hipMemcpyHostToDevice, ctx->stream));
        return SLAM_OK;
    }
    SLAM_HIP(ctx, hipMemcpyAsync(*d0, p0, bytes, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(*d1, p1, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SLAM_OK;
}

// ctx->X (N x 3) to host memory through the pinned area, then synchronise; everything enqueued before is done after it
static int download_points(slam_ctx* ctx, double* X, int64_t N) {
    const size_t bytes = (size_t)N * 3 * sizeof(double);
    char* h = stage_area(ctx, bytes);
    SLAM_HIP(ctx, hipMemcpyAsync(h ? (void*)h : (void*)X, ctx->X.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) memcpy(X, h, bytes);
    return SLAM_OK;
}

extern "C" int slam_ransac_f(slam_ctx* ctx, const double* p0, const double* p1, int N, const int32_t* idx, int sampler,
                             uint64_t seed, int64_t it0, int H, double threshold, double* record, uint8_t* mask,
                             int32_t* counts, double* errsums, double* F_all) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, p0 && p1 && record, "points or record pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");
    SLAM_REQUIRE(ctx, H >= 1, "H < 1");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pts_bytes = (((size_t)N * 16 + 63) & ~(size_t)63) + (size_t)N * 16;
    const size_t out_at = (pts_bytes + 255) & ~(size_t)255;
    (void)stage_area(ctx, out_at + 256 + (size_t)N);  // sized once for the uploads and the downloads of this call
    double *d0, *d1;
    SLAM_TRY(stage_points(ctx, p0, p1, N, &d0, &d1));
    const int32_t* d_idx = nullptr;
    if (idx) {
        SLAM_TRY(slam_reserve(ctx, ctx->idx, (size_t)H * 8 * sizeof(int32_t)));
        SLAM_HIP(ctx, hipMemcpyAsync(ctx->idx.p, idx, (size_t)H * 8 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        d_idx = (const int32_t*)ctx->idx.p;
    }
    // The record and the mask share one device buffer (record first, mask 256 bytes in), so that ONE copy brings both
    // back: a second small download costs 3 us of a 40 us call.
    SLAM_TRY(slam_reserve(ctx, ctx->mask, 256 + (size_t)N));
    double* d_rec = (double*)ctx->mask.p;
    uint8_t* d_mask = mask ? (uint8_t*)ctx->mask.p + 256 : nullptr;
    SLAM_TRY(slam_reserve(ctx, ctx->F, (size_t)H * 9 * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->counts, (size_t)H * sizeof(int32_t)));
    SLAM_TRY(slam_reserve(ctx, ctx->errsums, (size_t)H * sizeof(double)));
    SLAM_TRY(ransac_enqueue(ctx, d0, d1, N, d_idx, sampler, seed, it0, H, threshold, d_rec, d_mask,
                            (int32_t*)ctx->counts.p, (double*)ctx->errsums.p, (double*)ctx->F.p, false, errsums != nullptr));
    // record and mask come back through the pinned area (behind the staged points) and are handed over after the
    // synchronisation of check_flags(); the optional per-hypothesis arrays go directly
    char* h_out = stage_area(ctx, out_at + 256 + (size_t)N);
    if (h_out) {
        h_out += out_at;
        SLAM_HIP(ctx, hipMemcpyAsync(h_out, d_rec, mask ? 256 + (size_t)N : SLAM_REC_LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        SLAM_HIP(ctx, hipMemcpyAsync(record, d_rec, SLAM_REC_LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (mask) SLAM_HIP(ctx, hipMemcpyAsync(mask, d_mask, (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (counts) SLAM_HIP(ctx, hipMemcpyAsync(counts, ctx->counts.p, (size_t)H * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (errsums) SLAM_HIP(ctx, hipMemcpyAsync(errsums, ctx->errsums.p, (size_t)H * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (F_all) SLAM_HIP(ctx, hipMemcpyAsync(F_all, ctx->F.p, (size_t)H * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    const int rc = check_flags(ctx);
    if (h_out && rc != SLAM_EHIP) {
        memcpy(record, h_out, SLAM_REC_LEN * sizeof(double));
        if (mask) memcpy(mask, h_out + 256, (size_t)N);
    }
    return rc;
}

extern "C" int slam_ransac_f_adaptive(slam_ctx* ctx, const double* p0, const double* p1, int N, int sampler,
                                      uint64_t seed, int64_t max_iterations, int chunk, double confidence,
                                      double threshold, double* record, uint8_t* mask, int64_t* iterations_run) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, p0 && p1 && record, "points or record pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");
    SLAM_REQUIRE(ctx, max_iterations >= 1 && chunk >= 1, "max_iterations < 1 or chunk < 1");
    SLAM_REQUIRE(ctx, confidence > 0.0 && confidence < 1.0, "confidence must be inside (0, 1)");
    SLAM_REQUIRE(ctx, sampler == SLAM_SAMPLER_CPYTHON || sampler == SLAM_SAMPLER_COUNTER, "unknown sampler");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t limit = slam_num_hypotheses(N, max_iterations);
    double *d0, *d1;
    SLAM_TRY(stage_points(ctx, p0, p1, N, &d0, &d1));
    SLAM_TRY(slam_reserve(ctx, ctx->mask, (size_t)N));
    const int hmax = (int)std::min<int64_t>(chunk, limit);
    SLAM_TRY(slam_reserve(ctx, ctx->F, (size_t)hmax * 9 * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->counts, (size_t)hmax * sizeof(int32_t)));
    SLAM_TRY(slam_reserve(ctx, ctx->errsums, (size_t)hmax * sizeof(double)));
    double best[SLAM_REC_LEN];
    bool have = false;
    int64_t done = 0, need = limit;
    double* hrec = ctx->h_small + 160;
    while (done < need) {
        const int H = (int)std::min<int64_t>(chunk, limit - done);
        SLAM_TRY(ransac_enqueue(ctx, d0, d1, N, nullptr, sampler, seed, done, H, threshold, (double*)ctx->record.p, nullptr,
                                (int32_t*)ctx->counts.p, (double*)ctx->errsums.p, (double*)ctx->F.p));
        SLAM_HIP(ctx, hipMemcpyAsync(hrec, ctx->record.p, SLAM_REC_LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SLAM_TRY(check_flags(ctx));
        if (have) {
            double two[2 * SLAM_REC_LEN];
            memcpy(two, best, sizeof best);
            memcpy(two + SLAM_REC_LEN, hrec, sizeof best);
            slam_merge_records(two, 2, best);
        } else {
            memcpy(best, hrec, sizeof best);
            have = true;
        }
        done += H;
        const double w = best[SLAM_REC_COUNT] / (double)N;
        const double w8 = std::pow(w, 8.0);
        int64_t est;
        if (w8 >= 1.0) est = 1;
        else if (w8 <= 0.0) est = limit;
        else {
            const double k = std::ceil(std::log(1.0 - confidence) / std::log1p(-w8));
            est = k >= (double)limit ? limit : std::max<int64_t>(1, (int64_t)k);
        }
        need = std::min(need, std::max(est, (int64_t)1));  // the estimate only ever shortens the run
        if (done >= limit) break;
    }
    memcpy(record, best, sizeof best);
    if (iterations_run) *iterations_run = done;
    if (mask) {
        // the winner's mask: the selection kernel on a one-entry table holding the winner
        double* dF = (double*)ctx->F.p;
        int32_t one_count = (int32_t)best[SLAM_REC_COUNT];
        SLAM_HIP(ctx, hipMemcpyAsync(dF, best + SLAM_REC_F, 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(ctx->counts.p, &one_count, sizeof one_count, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(ctx->errsums.p, best + SLAM_REC_ERRSUM, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const double cut = ctx->sampson ? threshold : square_cutoff(threshold);
        const double2 *q0 = (const double2*)d0, *q1 = (const double2*)d1;
        if (ctx->sampson)
            SLAM_TRY(slam_launch(ctx, -1, k_select<true>, select_grid(N, ctx->mask.p), dim3(SLAM_SELECT_THREADS), 0, (const int32_t*)ctx->counts.p,
                                 (const double*)ctx->errsums.p, (const int32_t*)nullptr, (const double*)dF, 1,
                                 (int64_t)best[SLAM_REC_INDEX], q0, q1, N,
                                 cut, (double*)ctx->record.p, (uint8_t*)ctx->mask.p, (const PairArgs*)nullptr));
        else
            SLAM_TRY(slam_launch(ctx, -1, k_select<false>, select_grid(N, ctx->mask.p), dim3(SLAM_SELECT_THREADS), 0, (const int32_t*)ctx->counts.p,
                                 (const double*)ctx->errsums.p, (const int32_t*)nullptr, (const double*)dF, 1,
                                 (int64_t)best[SLAM_REC_INDEX], q0, q1, N,
                                 cut, (double*)ctx->record.p, (uint8_t*)ctx->mask.p, (const PairArgs*)nullptr));
        SLAM_HIP(ctx, hipMemcpyAsync(mask, ctx->mask.p, (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SLAM_OK;
}

extern "C" int slam_merge_records(const double* records, int n_records, double* out) {
    if (!records || !out || n_records < 1) return SLAM_EINVAL;
    int best = 0;
    for (int r = 1; r < n_records; ++r) {
        const double* x = records + (size_t)r * SLAM_REC_LEN;
        const double* y = records + (size_t)best * SLAM_REC_LEN;
        bool wins;
        if (x[SLAM_REC_COUNT] != y[SLAM_REC_COUNT]) wins = x[SLAM_REC_COUNT] > y[SLAM_REC_COUNT];
        else if (x[SLAM_REC_ERRSUM] != y[SLAM_REC_ERRSUM]) wins = x[SLAM_REC_ERRSUM] < y[SLAM_REC_ERRSUM];
        else wins = x[SLAM_REC_INDEX] < y[SLAM_REC_INDEX];
        if (wins) best = r;
    }
    memcpy(out, records + (size_t)best * SLAM_REC_LEN, SLAM_REC_LEN * sizeof(double));
    return best;
}

// ================================================================================================
// exchange step of the sharded runs (RCCL)
// ================================================================================================
// The exchange runs on the context's ordered stream.  With slam_set_concurrency(n > 1) the RANSAC calls before it ran
// on the internal streams: the ordered stream waits (on the device, not on the host) for what they have queued so far.
static int join_lanes(slam_ctx* ctx) {
    SLAM_TRY(slam_flush_pending(ctx));  // selections that wait to be carried are part of what the exchange waits for
    if (ctx->stream != ctx->own_stream || ctx->lanes_in_use <= 1) return SLAM_OK;
    for (int l = 1; l < ctx->lanes_in_use; ++l) {
        if (!ctx->lane_stream[l] || ctx->lane_stream[l] == ctx->stream) continue;
        if (!ctx->lane_event[l]) SLAM_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_event[l], hipEventDisableTiming));
        SLAM_HIP(ctx, hipEventRecord(ctx->lane_event[l], ctx->lane_stream[l]));
        SLAM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_event[l], 0));
    }
    return SLAM_OK;
}

extern "C" int slam_merge_records_dev(slam_ctx* ctx, const double* d_records, int n_records, double* d_out) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, d_records && d_out && n_records >= 1, "NULL pointer or no records");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    SLAM_TRY(join_lanes(ctx));
    ctx->cur = ctx->stream;
    return slam_launch(ctx, -1, slamx::k_merge_records, dim3(1), dim3(64), 0, d_records, n_records, d_out);
}

extern "C" int slam_comm_unique_id(slam_ctx* ctx, void* id128) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, id128 != nullptr, "NULL pointer");
    SLAM_TRY(slam_need_rccl(ctx));
    static_assert(sizeof(slamx::ncclUniqueId) == SLAM_COMM_ID_BYTES, "unique id size");
    SLAM_RCCL(ctx, slamx::rccl().GetUniqueId((slamx::ncclUniqueId*)id128));
    return SLAM_OK;
}

extern "C" int slam_comm_init_rank(slam_ctx* ctx, const void* id128, int n_ranks, int rank) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, id128 != nullptr, "NULL pointer");
    SLAM_REQUIRE(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank outside 0..n_ranks-1");
    SLAM_REQUIRE(ctx, ctx->comm == nullptr, "the context already has a communicator");
    SLAM_TRY(slam_need_rccl(ctx));
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    SLAM_TRY(slam_reserve(ctx, ctx->xchg, (size_t)n_ranks * SLAM_REC_LEN * sizeof(double)));
    slamx::ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    slamx::ncclComm_t comm = nullptr;
    SLAM_RCCL(ctx, slamx::rccl().CommInitRank(&comm, n_ranks, id, rank));
    ctx->comm = comm;
    ctx->comm_ranks = n_ranks;
    return SLAM_OK;
}

// communicator and its size for an exchange call: the caller's (any RCCL communicator whose device is the context's),
// or the context's own
static int exchange_comm(slam_ctx* ctx, void* rccl_comm, slamx::ncclComm_t* comm, int* n_ranks) {
    SLAM_TRY(slam_need_rccl(ctx));
    if (rccl_comm) {
        *comm = (slamx::ncclComm_t)rccl_comm;
        SLAM_RCCL(ctx, slamx::rccl().CommCount(*comm, n_ranks));
    } else {
        SLAM_REQUIRE(ctx, ctx->comm != nullptr, "no communicator: call slam_comm_init_rank first or pass one");
        *comm = (slamx::ncclComm_t)ctx->comm;
        *n_ranks = ctx->comm_ranks;
    }
    return SLAM_OK;
}

extern "C" int slam_comm_count(slam_ctx* ctx, void* rccl_comm, int* n_ranks) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, n_ranks != nullptr, "NULL pointer");
    SLAM_TRY(slam_need_rccl(ctx));
    slamx::ncclComm_t comm = rccl_comm ? (slamx::ncclComm_t)rccl_comm : (slamx::ncclComm_t)ctx->comm;
    SLAM_REQUIRE(ctx, comm != nullptr, "no communicator: call slam_comm_init_rank first or pass one");
    SLAM_RCCL(ctx, slamx::rccl().CommCount(comm, n_ranks));
    return SLAM_OK;
}

extern "C" int slam_exchange_winner_dev(slam_ctx* ctx, void* rccl_comm, double* d_record) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, d_record != nullptr, "NULL pointer");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    slamx::ncclComm_t comm;
    int n_ranks = 0;
    SLAM_TRY(exchange_comm(ctx, rccl_comm, &comm, &n_ranks));
    SLAM_TRY(slam_reserve(ctx, ctx->xchg, (size_t)n_ranks * SLAM_REC_LEN * sizeof(double)));  // grows for a caller's communicator only
    SLAM_TRY(join_lanes(ctx));
    SLAM_RCCL(ctx, slamx::rccl().AllGather(d_record, ctx->xchg.p, SLAM_REC_LEN, slamx::kNcclFloat64, comm, ctx->stream));
    ctx->cur = ctx->stream;
    return slam_launch(ctx, -1, slamx::k_merge_records, dim3(1), dim3(64), 0, (const double*)ctx->xchg.p, n_ranks, d_record);
}

extern "C" int slam_allreduce_counts4_dev(slam_ctx* ctx, void* rccl_comm, int64_t* d_counts4) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, d_counts4 != nullptr, "NULL pointer");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    slamx::ncclComm_t comm;
    int n_ranks = 0;
    SLAM_TRY(exchange_comm(ctx, rccl_comm, &comm, &n_ranks));
    SLAM_TRY(join_lanes(ctx));
    SLAM_RCCL(ctx, slamx::rccl().AllReduce(d_counts4, d_counts4, 4, slamx::kNcclInt64, slamx::kNcclSum, comm, ctx->stream));
    return SLAM_OK;
}

// ================================================================================================
// eight-point solve
// ================================================================================================
extern "C" int slam_eightpoint(slam_ctx* ctx, const double* p0, const double* p1, int M, double* F) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, p0 && p1 && F, "NULL pointer");
    SLAM_REQUIRE(ctx, M >= 8, "fewer than 8 matches");  // reference: src/epipolar.py:23
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double *d0, *d1;
    SLAM_TRY(stage_points(ctx, p0, p1, M, &d0, &d1));
    double* dF = (double*)ctx->small.p + 512;
    if (M == 8) {
        static const int32_t iota[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        SLAM_TRY(slam_reserve(ctx, ctx->idx, sizeof iota));
        SLAM_HIP(ctx, hipMemcpyAsync(ctx->idx.p, iota, sizeof iota, hipMemcpyHostToDevice, ctx->stream));
        SLAM_TRY(slam_launch(ctx, SLAM_K_SOLVE, k_solve<-1>, dim3(1), dim3(64), 0, (const double2*)d0, (const double2*)d1, 8,
                             (const int32_t*)ctx->idx.p, (uint64_t)0, (int64_t)0, 1, (const uint32_t*)nullptr, (int64_t)0,
                             dF, (int32_t*)ctx->flags.p));
    } else {
        SLAM_TRY(slam_launch(ctx, -1, k_eightpoint_general, dim3(1), dim3(64), 0, (const double2*)d0, (const double2*)d1,
                             M, dF));
    }
    SLAM_HIP(ctx, hipMemcpyAsync(F, dF, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ================================================================================================
// normalisation, F -> E
// ================================================================================================
extern "C" int slam_normalize_points(slam_ctx* ctx, const double* p, int N, double* out, double* T) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, p && out && T, "NULL pointer");
    SLAM_REQUIRE(ctx, N >= 1, "no points");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)N * 2 * sizeof(double);
    SLAM_TRY(slam_reserve(ctx, ctx->pts, 2 * bytes + 64));
    double* din = (double*)ctx->pts.p;
    double* dout = (double*)((char*)ctx->pts.p + ((bytes + 63) & ~(size_t)63));
    double* dT = (double*)ctx->small.p + 512;
    SLAM_HIP(ctx, hipMemcpyAsync(din, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    SLAM_TRY(slam_launch(ctx, -1, k_normalize, dim3(1), dim3(1024), 0, (const double2*)din, N, (double2*)dout, dT));
    SLAM_HIP(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(T, dT, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_fundamental_to_essential(slam_ctx* ctx, const double* F, const double* K, double* E) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, F && K && E, "NULL pointer");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double* dE = (double*)ctx->small.p + 512;
    Mat3x2 fk;
    memcpy(fk.A, F, sizeof fk.A);
    memcpy(fk.B, K, sizeof fk.B);
    SLAM_TRY(slam_launch(ctx, -1, k_f_to_e, dim3(1), dim3(64), 0, fk, dE));
    SLAM_HIP(ctx, hipMemcpyAsync(E, dE, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ================================================================================================
// triangulation
// ================================================================================================
extern "C" int slam_triangulate_dev(slam_ctx* ctx, const double* P0_host, const double* P1_host, const double* d_p0,
                                    const double* d_p1, int64_t N, double* d_X) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, P0_host && P1_host, "camera matrix pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 0 && N < ((int64_t)1 << 31) * 256, "N out of range");
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(ctx, d_p0 && d_p1 && d_X, "NULL pointer");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    Cams cams;  // the two 3x4 matrices travel as kernel arguments: nothing to stage, the call stays asynchronous
    memcpy(cams.P0, P0_host, sizeof cams.P0);
    memcpy(cams.P1, P1_host, sizeof cams.P1);
    SLAM_TRY(slam_launch(ctx, SLAM_K_TRIANGULATE, k_triangulate, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, cams,
                         (const double2*)d_p0, (const double2*)d_p1, N, d_X));
    return SLAM_OK;
}

extern "C" int slam_triangulate(slam_ctx* ctx, const double* P0, const double* P1, const double* p0, const double* p1,
                                int64_t N, double* X) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, P0 && P1, "camera matrix pointer is NULL");
    SLAM_REQUIRE(ctx, N >= 0, "N < 0");
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(ctx, p0 && p1 && X, "NULL pointer");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double *d0, *d1;
    SLAM_TRY(stage_points(ctx, p0, p1, N, &d0, &d1));
    SLAM_TRY(slam_reserve(ctx, ctx->X, (size_t)N * 3 * sizeof(double)));
    SLAM_TRY(slam_triangulate_dev(ctx, P0, P1, d0, d1, N, (double*)ctx->X.p));
    SLAM_TRY(download_points(ctx, X, N));
    return SLAM_OK;
}

// ================================================================================================
// pose recovery
// ================================================================================================
// sets of counters for launches beyond a million matches, or NULL (they live behind the small-matrix block)
#define SLAM_SMALL_WORDS 1024
#define SLAM_TRI4_PER_WG_SETS 256
static unsigned long long* tri4_count_sets(slam_ctx* ctx, int64_t N) {
    return N > ((int64_t)1 << 20) ? (unsigned long long*)((double*)ctx->small.p + SLAM_SMALL_WORDS) : nullptr;
}

static int tri4_enqueue(slam_ctx* ctx, const double* d_p0, const double* d_p1, int64_t N, unsigned long long* d_counts4) {
    SLAM_TRY(slam_reserve(ctx, ctx->X4, (size_t)2 * (size_t)N * 3 * sizeof(double)));
    unsigned long long* sets = tri4_count_sets(ctx, N);
    int per_wg = sets ? SLAM_TRI4_PER_WG_SETS : tri4_per_wg(N);
    if (sets)
        if (const char* g = getenv("SLAM_AMD_TRI4_PER_WG")) {  // tuning aid: matches per workgroup of the bulk launch
            const int v = atoi(g);
            if (v >= 64 && v % 64 == 0) per_wg = v;
        }
    if (sets) SLAM_HIP(ctx, hipMemsetAsync(sets, 0, SLAM_TRI4_COUNT_SETS * 4 * sizeof(unsigned long long), ctx->cur));
    SLAM_TRY(slam_launch(ctx, SLAM_K_TRIANGULATE, k_triangulate4, dim3((unsigned)((N + per_wg - 1) / per_wg)), dim3(64), 0,
                         (const double*)ctx->small.p, (const double2*)d_p0, (const double2*)d_p1, N, (double*)ctx->X4.p,
                         d_counts4, (const PairArgs*)nullptr, per_wg, sets));
    if (sets) SLAM_TRY(slam_launch(ctx, -1, k_fold_counts4, dim3(1), dim3(SLAM_TRI4_COUNT_SETS * 4), 0, (const unsigned long long*)sets, d_counts4));
    return SLAM_OK;
}

extern "C" int slam_pose_candidates_dev(slam_ctx* ctx, const double* E_host, const double* K_host, const double* d_p0,
                                        const double* d_p1, int64_t N, int64_t* d_counts4) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, E_host && K_host && d_p0 && d_p1 && d_counts4, "NULL pointer");
    SLAM_REQUIRE(ctx, N >= 1 && N < ((int64_t)1 << 31) * 256, "N out of range");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double* pw = (double*)ctx->small.p;
    SLAM_TRY(slam_reserve(ctx, ctx->X4, (size_t)2 * (size_t)N * 3 * sizeof(double)));
    Mat3x2 ek;
    memcpy(ek.A, E_host, sizeof ek.A);
    memcpy(ek.B, K_host, sizeof ek.B);
    SLAM_TRY(slam_launch(ctx, -1, k_pose_prepare, dim3(1), dim3(64), 0, ek, pw, (unsigned long long*)d_counts4));
    return tri4_enqueue(ctx, d_p0, d_p1, N, (unsigned long long*)d_counts4);
}

extern "C" int slam_pose_select_dev(slam_ctx* ctx, const int64_t* d_counts4, int64_t N, double* d_pose, double* d_X,
                                    int32_t* d_winner) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, d_counts4 && d_pose && d_winner, "NULL pointer");
    SLAM_REQUIRE(ctx, N >= 1 && ctx->X4.cap >= (size_t)2 * (size_t)N * 3 * sizeof(double),
                 "slam_pose_candidates_dev must run first with the same N");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    SLAM_TRY(slam_launch(ctx, -1, k_pose_select, dim3(1), dim3(64), 0, (const double*)ctx->small.p,
                         (const unsigned long long*)d_counts4, d_pose, d_winner));
    if (d_X) {
        const unsigned blocks = (unsigned)std::min<int64_t>((N * 3 + 255) / 256, 4096);
        SLAM_TRY(slam_launch(ctx, -1, k_emit_points, dim3(blocks), dim3(256), 0, (const double*)ctx->X4.p,
                             (const int32_t*)d_winner, N, d_X, (uint8_t*)nullptr));
    }
    return SLAM_OK;
}

extern "C" int slam_pose_from_E(slam_ctx* ctx, const double* E, const double* K, const double* p0, const double* p1,
                                int64_t N, double* pose, double* X, int32_t* counts4, int32_t* winner) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, E && K && p0 && p1 && pose && X, "NULL pointer");
    SLAM_REQUIRE(ctx, N >= 1, "no points");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double *d0, *d1;
    SLAM_TRY(stage_points(ctx, p0, p1, N, &d0, &d1));
    SLAM_TRY(slam_reserve(ctx, ctx->X, (size_t)N * 3 * sizeof(double)));
    double* small = (double*)ctx->small.p;
    int64_t* d_counts4 = (int64_t*)(small + 700);
    double* d_pose = small + 710;
    int32_t* d_winner = (int32_t*)(small + 730);
    SLAM_TRY(slam_pose_candidates_dev(ctx, E, K, d0, d1, N, d_counts4));
    SLAM_TRY(slam_pose_select_dev(ctx, d_counts4, N, d_pose, (double*)ctx->X.p, d_winner));
    SLAM_HIP(ctx, hipMemcpyAsync(ctx->h_small + 32, d_counts4, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(ctx->h_small + 40, d_pose, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(ctx->h_small + 60, d_winner, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_TRY(download_points(ctx, X, N));
    memcpy(pose, ctx->h_small + 40, 16 * sizeof(double));
    const int64_t* c = (const int64_t*)(ctx->h_small + 32);
    if (counts4) for (int i = 0; i < 4; ++i) counts4[i] = (int32_t)c[i];
    if (winner) *winner = *(const int32_t*)(ctx->h_small + 60);
    return SLAM_OK;
}

// ================================================================================================
// whole pair
// ================================================================================================
// the pair pipeline always scores with the reference's residual, whatever slam_set_scoring says
namespace {
struct KeepScoring {
    slam_ctx* c;
    bool was;
    explicit KeepScoring(slam_ctx* c_) : c(c_), was(c_->sampson) { c->sampson = false; }
    ~KeepScoring() { c->sampson = was; }
};
}  // namespace

// Whole-pair pipeline on matches that are already on the device (r0, r1: pixel coordinates, (N,2) each): normalise,
// RANSAC, denormalise, F -> E, four candidate triangulations, selection, z > 0 flags; then the small results and the
// requested arrays come back and the stream is synchronised.
static int init_pair_on_device(slam_ctx* ctx, const double* r0, const double* r1, int N, const double* K, int iterations,
                               double* F, double* E, double* pose, double* X, uint8_t* good, double* record,
                               int32_t* counts4, int32_t* winner) {
    KeepScoring keep_scoring(ctx);
    const int H = (int)slam_num_hypotheses(N, iterations);
    const size_t bytes = (size_t)N * 2 * sizeof(double), stride = (bytes + 63) & ~(size_t)63;
    SLAM_TRY(slam_reserve(ctx, ctx->norm, 2 * stride + 64));
    double *n0 = (double*)ctx->norm.p, *n1 = (double*)((char*)ctx->norm.p + stride);
    SLAM_TRY(slam_reserve(ctx, ctx->X, (size_t)N * 3 * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->mask, (size_t)N));
    SLAM_TRY(slam_reserve(ctx, ctx->X4, (size_t)2 * (size_t)N * 3 * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->F, (size_t)H * 9 * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->counts, (size_t)H * sizeof(int32_t)));
    SLAM_TRY(slam_reserve(ctx, ctx->errsums, (size_t)H * sizeof(double)));
    SLAM_TRY(slam_fill_raw_table(ctx, SLAM_SEED_BASE, 0, H));
    double* small = (double*)ctx->small.p;
    double* pw = small;
    unsigned long long* d_counts4 = (unsigned long long*)(small + 700);
    double* d_pose = small + 710;
    int32_t* d_winner = (int32_t*)(small + 730);
    double* d_record = small + 740;  // with the other small results, so that one copy brings all of them back
    Mat3x2 kk;
    memcpy(kk.A, K, sizeof kk.A);
    memset(kk.B, 0, sizeof kk.B);
    SLAM_TRY(slam_launch(ctx, -1, k_normalize_pair, dim3(2), dim3(1024), 0, (const double2*)r0, (const double2*)r1, N,
                         (double2*)n0, (double2*)n1, pw + PW_T0, pw + PW_T1, (const PairArgs*)nullptr));
    SLAM_TRY(ransac_enqueue(ctx, n0, n1, N, nullptr, SLAM_SAMPLER_CPYTHON, SLAM_SEED_BASE, 0, H, SLAM_DEFAULT_THRESHOLD,
                            d_record, nullptr, nullptr, nullptr, nullptr));
    SLAM_TRY(slam_launch(ctx, -1, k_pair_mid, dim3(1), dim3(64), 0, kk, (const double*)d_record, pw, d_counts4, (const PairArgs*)nullptr));
    SLAM_TRY(tri4_enqueue(ctx, r0, r1, N, d_counts4));
    const unsigned blocks = (unsigned)std::min<int64_t>(((int64_t)N * 3 + 255) / 256, 4096);
    SLAM_TRY(slam_launch(ctx, -1, k_select_emit, dim3(blocks), dim3(256), 0, (const double*)pw,
                         (const unsigned long long*)d_counts4, (const double*)ctx->X4.p, (int64_t)N, d_pose, d_winner,
                         (double*)ctx->X.p, (uint8_t*)ctx->mask.p, (const PairArgs*)nullptr));
    // points and flags come back through the pinned area (the uploads that may still sit in it were enqueued earlier
    // on this stream, so they have been read by the time these copies write)
    const size_t x_bytes = (size_t)N * 3 * sizeof(double);
    char* h_out = (X || good) ? stage_area(ctx, x_bytes + (size_t)N) : nullptr;
    if (X) SLAM_HIP(ctx, hipMemcpyAsync(h_out ? (void*)h_out : (void*)X, ctx->X.p, x_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (good) SLAM_HIP(ctx, hipMemcpyAsync(h_out ? (void*)(h_out + x_bytes) : (void*)good, ctx->mask.p, (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    double* hs = ctx->h_small;  // mirrors the device block `small` (1024 words)
    SLAM_HIP(ctx, hipMemcpyAsync(hs, small, 760 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_TRY(check_flags(ctx));
    if (h_out && X) memcpy(X, h_out, x_bytes);
    if (h_out && good) memcpy(good, h_out + x_bytes, (size_t)N);
    if (E) memcpy(E, hs + PW_E, 9 * sizeof(double));
    if (F) memcpy(F, hs + PW_F, 9 * sizeof(double));
    if (pose) memcpy(pose, hs + 710, 16 * sizeof(double));
    if (record) memcpy(record, hs + 740, SLAM_REC_LEN * sizeof(double));
    const int64_t* c = (const int64_t*)(hs + 700);
    if (counts4) for (int i = 0; i < 4; ++i) counts4[i] = (int32_t)c[i];
    if (winner) *winner = *(const int32_t*)(hs + 730);
    return SLAM_OK;
}

extern "C" int slam_init_pair(slam_ctx* ctx, const double* pts0, const double* pts1, int N, const double* K,
                              int iterations, double* F, double* E, double* pose, double* X, uint8_t* good,
                              double* record, int32_t* counts4, int32_t* winner) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, pts0 && pts1 && K, "NULL pointer");
    SLAM_REQUIRE(ctx, N >= 8, "fewer than 8 matches");
    SLAM_REQUIRE(ctx, iterations >= 1, "iterations < 1");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    double *r0, *r1;
    SLAM_TRY(stage_points(ctx, pts0, pts1, N, &r0, &r1));
    return init_pair_on_device(ctx, r0, r1, N, K, iterations, F, E, pose, X, good, record, counts4, winner);
}

// ------------------------------------------------------------------------------------------------
// a batch of pairs
// ------------------------------------------------------------------------------------------------
namespace {
enum { PB_PW = 0, PB_COUNTS = 256, PB_POSE = 260, PB_WIN = 276, PB_REC = 280, PB_LEN = 320 };  // per-pair block, f64 words
static_assert(PW_LEN <= PB_COUNTS, "the pose work block must fit its slot");
}  // namespace

extern "C" int slam_init_pairs(slam_ctx* ctx, const double* pts0, const double* pts1, const int64_t* offsets, int P,
                               const double* K, int iterations, double* F, double* E, double* pose, double* X,
                               uint8_t* good, double* record, int32_t* counts4, int32_t* winner) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, pts0 && pts1 && offsets && K, "NULL pointer");
    SLAM_REQUIRE(ctx, P >= 1, "no pairs");
    SLAM_REQUIRE(ctx, iterations >= 1, "iterations < 1");
    SLAM_REQUIRE(ctx, offsets[0] == 0, "offsets must start at 0");
    int maxH = 0;
    for (int p = 0; p < P; ++p) {
        const int64_t n = offsets[p + 1] - offsets[p];
        SLAM_REQUIRE(ctx, n >= 8 && n <= SLAM_MAX_MATCHES, "a pair has fewer than 8 matches (or more than 2^27)");
        maxH = std::max(maxH, (int)slam_num_hypotheses(n, iterations));
    }
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const bool scoring_was = ctx->sampson;  // the pair pipeline always scores with the reference's residual
    ctx->sampson = false;
    struct Restore { slam_ctx* c; bool v; ~Restore() { c->sampson = v; } } restore_scoring{ctx, scoring_was};
    const int64_t total = offsets[P];
    int maxN = 0;
    for (int p = 0; p < P; ++p) maxN = std::max(maxN, (int)(offsets[p + 1] - offsets[p]));
    // every allocation first: growing a buffer synchronises the context
    const size_t b_pts = align256((size_t)total * 16), b_x = align256((size_t)total * 24), b_m = align256((size_t)total);
    const size_t b_blk = align256((size_t)P * PB_LEN * sizeof(double));
    const size_t b_F = align256((size_t)P * maxH * 9 * sizeof(double)), b_c = align256((size_t)P * maxH * sizeof(int32_t));
    const size_t b_e = align256((size_t)P * maxH * sizeof(double)), b_args = align256((size_t)P * sizeof(PairArgs));
    SLAM_TRY(slam_reserve(ctx, ctx->batch, 4 * b_pts + 3 * b_x + b_m + b_blk + b_F + b_c + b_e + 3 * b_args));
    SLAM_TRY(slam_fill_raw_table(ctx, SLAM_SEED_BASE, 0, maxH));
    char* at = (char*)ctx->batch.p;
    auto take = [&](size_t bytes) { char* q = at; at += bytes; return q; };
    double *r0 = (double*)take(b_pts), *r1 = (double*)take(b_pts), *n0 = (double*)take(b_pts), *n1 = (double*)take(b_pts);
    double* X4 = (double*)take(2 * b_x);
    double* dX = (double*)take(b_x);
    uint8_t* dmask = (uint8_t*)take(b_m);
    double* blocks = (double*)take(b_blk);
    double* dF = (double*)take(b_F);
    int32_t* dC = (int32_t*)take(b_c);
    double* dE = (double*)take(b_e);
    PairArgs* d_args = (PairArgs*)take(3 * b_args);  // all pairs, then the same pairs grouped by scoring shape
    // One launch per stage for the whole batch: blockIdx.y selects the pair, the grid is sized for the largest one.
    // The scoring kernel's shape (matches per lane) depends on N, so pairs are grouped by shape: at most three launches.
    std::vector<PairArgs> args((size_t)P), grouped;
    grouped.reserve((size_t)P);
    for (int p = 0; p < P; ++p) {
        const int64_t off = offsets[p];
        PairArgs& a = args[(size_t)p];
        double* blk = blocks + (size_t)p * PB_LEN;
        a.r0 = (const double2*)(r0 + 2 * off); a.r1 = (const double2*)(r1 + 2 * off);
        a.n0 = (double2*)(n0 + 2 * off); a.n1 = (double2*)(n1 + 2 * off);
        a.N = (int)(offsets[p + 1] - off);
        a.H = (int)slam_num_hypotheses(a.N, iterations);
        a.pw = blk + PB_PW;
        a.F_all = dF + (size_t)p * maxH * 9; a.counts = dC + (size_t)p * maxH; a.errsums = dE + (size_t)p * maxH;
        a.record = blk + PB_REC;
        a.counts4 = (unsigned long long*)(blk + PB_COUNTS);
        a.X4 = X4 + 2 * 3 * off; a.X = dX + 3 * off; a.good = dmask + off;
        a.pose = blk + PB_POSE; a.winner = (int32_t*)(blk + PB_WIN);
    }
    int class_begin[4] = {0, 0, 0, 0};
    for (int cls = 0; cls < 3; ++cls) {
        for (int p = 0; p < P; ++p) {
            const int N = args[(size_t)p].N;
            const int c = N <= 512 ? 0 : (N <= 2048 ? 1 : 2);  // the same limits as SLAM_DISPATCH_SHAPE
            if (c == cls) grouped.push_back(args[(size_t)p]);
        }
        class_begin[cls + 1] = (int)grouped.size();
    }
    // From here on copies from and to function-local vectors are in flight: whatever fails below, the stream is
    // drained before the vectors go out of scope.
    std::vector<double> hb((size_t)P * PB_LEN);
    auto enqueue_and_wait = [&]() -> int {
    SLAM_HIP(ctx, hipMemcpyAsync(r0, pts0, (size_t)total * 16, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(r1, pts1, (size_t)total * 16, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(d_args, args.data(), (size_t)P * sizeof(PairArgs), hipMemcpyHostToDevice, ctx->stream));
    PairArgs* d_grouped = (PairArgs*)((char*)d_args + b_args);
    SLAM_HIP(ctx, hipMemcpyAsync(d_grouped, grouped.data(), (size_t)P * sizeof(PairArgs), hipMemcpyHostToDevice, ctx->stream));
    Mat3x2 kk;
    memcpy(kk.A, K, sizeof kk.A);
    memset(kk.B, 0, sizeof kk.B);
    const double rcut = square_cutoff(SLAM_DEFAULT_THRESHOLD);
    const PairArgs* all = d_args;
    SLAM_TRY(slam_launch(ctx, -1, k_normalize_pair, dim3(2, P), dim3(1024), 0, (const double2*)nullptr, (const double2*)nullptr, 0,
                         (double2*)nullptr, (double2*)nullptr, (double*)nullptr, (double*)nullptr, all));
    for (int cls = 0; cls < 3; ++cls) {
        const int n_cls = class_begin[cls + 1] - class_begin[cls];
        if (n_cls == 0) continue;
        int clsH = 0;
        for (int q = class_begin[cls]; q < class_begin[cls + 1]; ++q) clsH = std::max(clsH, grouped[(size_t)q].H);
        const PairArgs* part = d_grouped + class_begin[cls];
        const dim3 grid((clsH + SLAM_HB - 1) / SLAM_HB, n_cls);
#define SLAM_BATCH_RANSAC(PPL)                                                                                            \
    SLAM_TRY(slam_launch(ctx, SLAM_K_SCORE, k_ransac<SLAM_HB, PPL, 8, SLAM_SAMPLER_CPYTHON, false>, grid, dim3(512), 0,     \
                         (const uint32_t*)ctx->raw.p, ctx->raw_it0, (int64_t)0, 0, 0, (const double2*)nullptr,            \
                         (const double2*)nullptr, (const int32_t*)nullptr, (uint64_t)SLAM_SEED_BASE, rcut,                \
                         (double*)nullptr, (int32_t*)nullptr, (double*)nullptr, (int32_t*)ctx->flags.p, part, CarriedSelect{}))
        if (cls == 0) SLAM_BATCH_RANSAC(1);
        else if (cls == 1) SLAM_BATCH_RANSAC(4);
        else SLAM_BATCH_RANSAC(8);
#undef SLAM_BATCH_RANSAC
    }
    SLAM_TRY(slam_launch(ctx, SLAM_K_SELECT, k_select<false>, dim3(1, P), dim3(SLAM_SELECT_THREADS), 0, (const int32_t*)nullptr,
                         (const double*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, 0, (int64_t)0,
                         (const double2*)nullptr, (const double2*)nullptr, 0, rcut, (double*)nullptr, (uint8_t*)nullptr, all));
    SLAM_TRY(slam_launch(ctx, -1, k_pair_mid, dim3(1, P), dim3(64), 0, kk, (const double*)nullptr, (double*)nullptr,
                         (unsigned long long*)nullptr, all));
    const int per_wg = tri4_per_wg(maxN);  // (each pair has counts of its own: the atomics of different pairs do not queue)
    SLAM_TRY(slam_launch(ctx, SLAM_K_TRIANGULATE, k_triangulate4, dim3((unsigned)((maxN + per_wg - 1) / per_wg), P), dim3(64), 0,
                         (const double*)nullptr, (const double2*)nullptr, (const double2*)nullptr, (int64_t)0,
                         (double*)nullptr, (unsigned long long*)nullptr, all, per_wg, (unsigned long long*)nullptr));
    const unsigned eb = (unsigned)std::min<int64_t>(((int64_t)maxN * 3 + 255) / 256, 64);
    SLAM_TRY(slam_launch(ctx, -1, k_select_emit, dim3(eb, P), dim3(256), 0, (const double*)nullptr,
                         (const unsigned long long*)nullptr, (const double*)nullptr, (int64_t)0, (double*)nullptr,
                         (int32_t*)nullptr, (double*)nullptr, (uint8_t*)nullptr, all));
    SLAM_HIP(ctx, hipMemcpyAsync(hb.data(), blocks, hb.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (X) SLAM_HIP(ctx, hipMemcpyAsync(X, dX, (size_t)total * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (good) SLAM_HIP(ctx, hipMemcpyAsync(good, dmask, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    return check_flags(ctx);  // synchronises
    };
    const int rc_batch = enqueue_and_wait();
    if (rc_batch != SLAM_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc_batch;
    }
    for (int p = 0; p < P; ++p) {
        const double* blk = hb.data() + (size_t)p * PB_LEN;
        if (E) memcpy(E + 9 * (size_t)p, blk + PB_PW + PW_E, 9 * sizeof(double));
        if (F) memcpy(F + 9 * (size_t)p, blk + PB_PW + PW_F, 9 * sizeof(double));
        if (pose) memcpy(pose + 16 * (size_t)p, blk + PB_POSE, 16 * sizeof(double));
        if (record) memcpy(record + SLAM_REC_LEN * (size_t)p, blk + PB_REC, SLAM_REC_LEN * sizeof(double));
        const int64_t* c = (const int64_t*)(blk + PB_COUNTS);
        if (counts4) for (int i = 0; i < 4; ++i) counts4[4 * (size_t)p + i] = (int32_t)c[i];
        if (winner) winner[p] = *(const int32_t*)(blk + PB_WIN);
    }
    return SLAM_OK;
}

// ================================================================================================
// tracking step
// ================================================================================================
static int64_t num_subsets_capped(int64_t n, int k, int64_t cap) {
    if (n < k || cap < 0) return 0;
    unsigned __int128 c = 1;
    for (int i = 1; i <= k; ++i) {
        c = c * (unsigned __int128)(n - k + i) / (unsigned)i;  // exact: C(n-k+i, i) after each step, growing with i
        if (c > (unsigned __int128)cap) return cap;
    }
    return (int64_t)c;
}

extern "C" int64_t slam_num_tracking_hypotheses(int64_t n_known, int64_t iterations) {
    return num_subsets_capped(n_known, 6, iterations);
}

extern "C" int slam_pose_from_points(slam_ctx* ctx, const double* image_points, const double* known_3d_points, int M,
                                     double* pose, double* extra) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, image_points && known_3d_points && pose, "NULL pointer");
    SLAM_REQUIRE(ctx, M >= 6, "fewer than 6 matches");  // reference: src/triangulation.py:82
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b2 = (size_t)M * 2 * sizeof(double), b3 = (size_t)M * 3 * sizeof(double);
    SLAM_TRY(slam_reserve(ctx, ctx->pts, b2 + b3 + 64));
    SLAM_TRY(slam_reserve(ctx, ctx->track, (size_t)(TW_LEN + HP_LEN) * sizeof(double)));
    double* duv = (double*)ctx->pts.p;
    double* dX = (double*)((char*)ctx->pts.p + ((b2 + 63) & ~(size_t)63));
    double* tw = (double*)ctx->track.p;
    double* hyp = tw + TW_LEN;
    SLAM_HIP(ctx, hipMemcpyAsync(duv, image_points, b2, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(dX, known_3d_points, b3, hipMemcpyHostToDevice, ctx->stream));
    TrackSetup ts;
    memset(&ts, 0, sizeof ts);
    ts.K[0] = ts.K[4] = ts.K[8] = 1.0;
    SLAM_TRY(slam_launch(ctx, -1, k_track_setup, dim3(1), dim3(64), 0, ts, tw));
    SLAM_TRY(slam_launch(ctx, SLAM_K_RESECT, k_resect, dim3(1), dim3(SLAM_RESECT_LANES), 0, (const double2*)duv,
                         (const double*)dX, M, (const int32_t*)nullptr, 1, (const double*)tw, hyp));
    SLAM_HIP(ctx, hipMemcpyAsync(ctx->h_small, hyp, HP_LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(pose, ctx->h_small + HP_POSE, 16 * sizeof(double));
    if (extra) memcpy(extra, ctx->h_small + HP_EXTRA, 24 * sizeof(double));
    return SLAM_OK;
}

// queues sample -> resect -> score -> select + final triangulation; everything stays on the device
// (with more hypotheses than this every workgroup of the final triangulation would repeat a long selection scan: two launches then)
#define SLAM_TRACK_FUSED_FINISH_MAX_H 2048
static int track_enqueue(slam_ctx* ctx, const double* previous_pose, const double* d_a, const double* d_b,
                         const double* d_X3, int M, const double* d_prev_all, const double* d_next_all, int64_t N,
                         const double* K, int H, double* d_pose, double* d_X) {
    SLAM_TRY(slam_fill_raw_table(ctx, SLAM_SEED_BASE, 0, H));
    SLAM_TRY(slam_reserve(ctx, ctx->track, ((size_t)TW_LEN + (size_t)H * HP_LEN) * sizeof(double)));
    SLAM_TRY(slam_reserve(ctx, ctx->counts, (size_t)H * sizeof(int32_t)));
    SLAM_TRY(slam_reserve(ctx, ctx->errsums, (size_t)H * sizeof(double)));
    double* tw = (double*)ctx->track.p;
    double* hyp = tw + TW_LEN;
    TrackSetup ts;
    memcpy(ts.K, K, sizeof ts.K);
    memcpy(ts.prev, previous_pose, sizeof ts.prev);  // the first three rows of the 4x4 pose
    SLAM_TRY(slam_launch(ctx, SLAM_K_RESECT, k_resect6, dim3((H + 3) / 4), dim3(64), 0, (const double2*)d_b, d_X3, M,
                         (const uint32_t*)ctx->raw.p, ctx->raw_it0, H, ts, tw, hyp, (int32_t*)ctx->flags.p));  // draws its samples, fills tw
    const int score_threads = std::min(SLAM_TRACK_SCORE_MAX_THREADS, std::max(64, (M + 63) / 64 * 64));
    SLAM_TRY(slam_launch(ctx, SLAM_K_TRACK_SCORE, k_track_score, dim3(H), dim3(score_threads), 0, (const double*)tw,
                         (const double*)hyp, (const double2*)d_a, (const double2*)d_b, d_X3, M,
                         (int32_t*)ctx->counts.p, (double*)ctx->errsums.p));
    if (H <= SLAM_TRACK_FUSED_FINISH_MAX_H) {  // selection and final triangulation in one launch
        SLAM_TRY(slam_launch(ctx, SLAM_K_TRIANGULATE, k_track_finish, dim3((unsigned)std::max<int64_t>(1, (N + 255) / 256)), dim3(256), 0,
                             (const int32_t*)ctx->counts.p, (const double*)ctx->errsums.p, H, (const double*)hyp, tw, d_pose,
                             (const double2*)d_prev_all, (const double2*)d_next_all, N, d_X));
        return SLAM_OK;
    }
    SLAM_TRY(slam_launch(ctx, -1, k_track_select, dim3(1), dim3(64), 0, (const int32_t*)ctx->counts.p,
                         (const double*)ctx->errsums.p, H, (const double*)hyp, tw, d_pose));
    if (N > 0)
        SLAM_TRY(slam_launch(ctx, SLAM_K_TRIANGULATE, k_triangulate_cams, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                             (const double*)(tw + TW_WIN), (const double2*)d_prev_all, (const double2*)d_next_all, N, d_X));
    return SLAM_OK;
}

#define SLAM_TRACK_ARGS_OK(ctx, M, N, iterations)                                           \
    SLAM_REQUIRE(ctx, (M) >= 6, "fewer than 6 matches with a known 3D position");           \
    SLAM_REQUIRE(ctx, (N) >= 0 && (N) < ((int64_t)1 << 31) * 256, "N out of range");        \
    SLAM_REQUIRE(ctx, (iterations) >= 1, "iterations < 1")

extern "C" int slam_track_pose_dev(slam_ctx* ctx, const double* previous_pose, const double* d_prev_known,
                                   const double* d_next_known, const double* d_known_3d_points, int M,
                                   const double* d_prev_all, const double* d_next_all, int64_t N, const double* K,
                                   int iterations, double* d_pose, double* d_X, double* d_best) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, previous_pose && d_prev_known && d_next_known && d_known_3d_points && K && d_pose, "NULL pointer");
    SLAM_TRACK_ARGS_OK(ctx, M, N, iterations);
    SLAM_REQUIRE(ctx, N == 0 || (d_prev_all && d_next_all && d_X), "NULL pointer for the full match list");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int H = (int)slam_num_tracking_hypotheses(M, iterations);
    SLAM_TRY(track_enqueue(ctx, previous_pose, d_prev_known, d_next_known, d_known_3d_points, M, d_prev_all, d_next_all,
                           N, K, H, d_pose, d_X));
    if (d_best)
        SLAM_HIP(ctx, hipMemcpyAsync(d_best, (double*)ctx->track.p + TW_BEST, 3 * sizeof(double), hipMemcpyDeviceToDevice,
                                     ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_track_pose(slam_ctx* ctx, const double* previous_pose, const double* prev_known,
                               const double* next_known, const double* known_3d_points, int M, const double* prev_all,
                               const double* next_all, int64_t N, const double* K, int iterations, double* pose,
                               double* X, double* best, int32_t* counts, double* errs, double* poses) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, previous_pose && prev_known && next_known && known_3d_points && K && pose, "NULL pointer");
    SLAM_TRACK_ARGS_OK(ctx, M, N, iterations);
    SLAM_REQUIRE(ctx, N == 0 || (prev_all && next_all && X), "NULL pointer for the full match list");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int H = (int)slam_num_tracking_hypotheses(M, iterations);
    // pts: prev_known | next_known | X_known | prev_all | next_all, each 64-byte aligned
    const size_t b2 = ((size_t)M * 2 * sizeof(double) + 63) & ~(size_t)63, b3 = ((size_t)M * 3 * sizeof(double) + 63) & ~(size_t)63;
    const size_t bn = ((size_t)N * 2 * sizeof(double) + 63) & ~(size_t)63;
    SLAM_TRY(slam_reserve(ctx, ctx->pts, 2 * b2 + b3 + 2 * bn + 64));
    SLAM_TRY(slam_reserve(ctx, ctx->X, (size_t)(N > 0 ? N : 1) * 3 * sizeof(double)));
    char* base = (char*)ctx->pts.p;
    double *da = (double*)base, *db = (double*)(base + b2), *dX3 = (double*)(base + 2 * b2);
    double *dpa = (double*)(base + 2 * b2 + b3), *dna = (double*)(base + 2 * b2 + b3 + bn);
    // the five arrays go up in one copy from the pinned area, laid out as on the device; the points come back through
    // it as well (behind the uploads, in stream order)
    const size_t up_bytes = 2 * b2 + b3 + 2 * bn, x_bytes = (size_t)N * 3 * sizeof(double);
    char* h = stage_area(ctx, up_bytes + x_bytes);
    if (h) {
        memcpy(h, prev_known, (size_t)M * 2 * sizeof(double));
        memcpy(h + b2, next_known, (size_t)M * 2 * sizeof(double));
        memcpy(h + 2 * b2, known_3d_points, (size_t)M * 3 * sizeof(double));
        if (N > 0) {
            memcpy(h + 2 * b2 + b3, prev_all, (size_t)N * 2 * sizeof(double));
            memcpy(h + 2 * b2 + b3 + bn, next_all, (size_t)N * 2 * sizeof(double));
        }
        SLAM_HIP(ctx, hipMemcpyAsync(base, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        SLAM_HIP(ctx, hipMemcpyAsync(da, prev_known, (size_t)M * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(db, next_known, (size_t)M * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(dX3, known_3d_points, (size_t)M * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (N > 0) {
            SLAM_HIP(ctx, hipMemcpyAsync(dpa, prev_all, (size_t)N * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            SLAM_HIP(ctx, hipMemcpyAsync(dna, next_all, (size_t)N * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    double* d_pose = (double*)ctx->small.p + 710;
    SLAM_TRY(track_enqueue(ctx, previous_pose, da, db, dX3, M, dpa, dna, N, K, H, d_pose, (double*)ctx->X.p));
    double* tw = (double*)ctx->track.p;
    double* hs = ctx->h_small;
    SLAM_HIP(ctx, hipMemcpyAsync(hs, d_pose, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(hs + 16, tw + TW_BEST, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (N > 0) SLAM_HIP(ctx, hipMemcpyAsync(h ? (void*)(h + up_bytes) : (void*)X, ctx->X.p, x_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) SLAM_HIP(ctx, hipMemcpyAsync(counts, ctx->counts.p, (size_t)H * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (errs) SLAM_HIP(ctx, hipMemcpyAsync(errs, ctx->errsums.p, (size_t)H * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (poses)
        SLAM_HIP(ctx, hipMemcpy2DAsync(poses, 16 * sizeof(double), tw + TW_LEN + HP_POSE, HP_LEN * sizeof(double),
                                       16 * sizeof(double), (size_t)H, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_TRY(check_flags(ctx));
    if (h && N > 0) memcpy(X, h + up_bytes, x_bytes);
    memcpy(pose, hs, 16 * sizeof(double));
    if (best) memcpy(best, hs + 16, 3 * sizeof(double));
    return SLAM_OK;
}

// ================================================================================================
// descriptor matching
// ================================================================================================

// number of train slices: about 1024 workgroups in all (measured best at 4096 x 4096), slices of at least 64 descriptors
static int match_slices(int n0, int n1) {
    const int qblocks = (n0 + 63) / 64;
    int s = (1024 + qblocks - 1) / qblocks;  // ~4 waves per SIMD hide the scalar-load latency of the train loop
    s = std::min(s, (n1 + 63) / 64);
    return std::max(1, std::min(s, 64));
}

// d_part: n0 * slices * 2 uint32
static int match_enqueue(slam_ctx* ctx, const uint8_t* d_des0, const double* d_pts0, int n0, const uint8_t* d_des1,
                         const double* d_pts1, int n1, int slices, uint32_t* d_part, double* d_m0, double* d_m1,
                         int32_t* d_n, int32_t* d_index, int32_t* d_distance) {
    const int per_slice = (n1 + slices - 1) / slices;
    SLAM_TRY(slam_launch(ctx, SLAM_K_MATCH, k_match_knn2, dim3((n0 + 63) / 64, slices), dim3(256), 0,
                         (const uint32_t*)d_des0, (const uint32_t*)d_des1, n0, n1, per_slice, (uint2*)d_part));
    SLAM_TRY(slam_launch(ctx, -1, k_match_finish, dim3(1), dim3(1024), 0, (const uint2*)d_part, slices, n0,
                         (const double2*)d_pts0, (const double2*)d_pts1, d_index, d_distance, (double2*)d_m0,
                         (double2*)d_m1, d_n));
    return SLAM_OK;
}

#define SLAM_MATCH_ARGS_OK(ctx, n0, n1)                                                              \
    SLAM_REQUIRE(ctx, (n0) >= 1 && (n1) >= 1, "empty descriptor set");                               \
    SLAM_REQUIRE(ctx, (n1) <= SLAM_MATCH_MAX_TRAIN, "more than 2^22 train descriptors")

extern "C" int slam_match_points_dev(slam_ctx* ctx, const uint8_t* d_des0, const double* d_pts0, int n0,
                                     const uint8_t* d_des1, const double* d_pts1, int n1, double* d_matched0,
                                     double* d_matched1, int32_t* d_n_matched, int32_t* d_index, int32_t* d_distance) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, d_des0 && d_des1 && d_pts0 && d_pts1 && d_matched0 && d_matched1 && d_n_matched, "NULL pointer");
    SLAM_MATCH_ARGS_OK(ctx, n0, n1);
    SLAM_REQUIRE(ctx, ((uintptr_t)d_des0 & 15) == 0 && ((uintptr_t)d_des1 & 15) == 0, "descriptor pointers must be 16-byte aligned");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int slices = match_slices(n0, n1);
    SLAM_TRY(slam_reserve(ctx, ctx->match, (size_t)n0 * slices * 2 * sizeof(uint32_t)));
    return match_enqueue(ctx, d_des0, d_pts0, n0, d_des1, d_pts1, n1, slices, (uint32_t*)ctx->match.p, d_matched0,
                         d_matched1, d_n_matched, d_index, d_distance);
}

// host form shared by slam_match_knn (no points) and slam_match_points
static int match_host(slam_ctx* ctx, const uint8_t* des0, const double* pts0, int n0, const uint8_t* des1,
                      const double* pts1, int n1, double* matched0, double* matched1, int32_t* n_matched, int32_t* index,
                      int32_t* distance) {
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int slices = match_slices(n0, n1);
    const bool gather = pts0 != nullptr;
    const size_t b_part = align256((size_t)n0 * slices * 2 * sizeof(uint32_t));
    const size_t b_d0 = align256((size_t)n0 * 32), b_d1 = align256((size_t)n1 * 32);
    const size_t b_p0 = gather ? align256((size_t)n0 * 16) : 0, b_p1 = gather ? align256((size_t)n1 * 16) : 0;
    const size_t b_m = gather ? align256((size_t)n0 * 16) : 0, b_i = align256((size_t)n0 * 2 * sizeof(int32_t));
    SLAM_TRY(slam_reserve(ctx, ctx->match, b_part + b_d0 + b_d1 + b_p0 + b_p1 + 2 * b_m + 2 * b_i + 256));
    char* at = (char*)ctx->match.p;
    auto take = [&](size_t bytes) { char* p = at; at += bytes; return p; };
    uint32_t* d_part = (uint32_t*)take(b_part);
    uint8_t* d_d0 = (uint8_t*)take(b_d0);
    uint8_t* d_d1 = (uint8_t*)take(b_d1);
    double* d_p0 = gather ? (double*)take(b_p0) : nullptr;
    double* d_p1 = gather ? (double*)take(b_p1) : nullptr;
    double* d_m0 = gather ? (double*)take(b_m) : nullptr;
    double* d_m1 = gather ? (double*)take(b_m) : nullptr;
    int32_t* d_index = (int32_t*)take(b_i);
    int32_t* d_distance = (int32_t*)take(b_i);
    int32_t* d_n = (int32_t*)take(256);
    // descriptors and keypoints go up in one copy from the pinned area (laid out as on the device: d0 | d1 | p0 | p1);
    // the matched points come back through it at full capacity together with their count: one synchronisation
    const size_t up_bytes = b_d0 + b_d1 + b_p0 + b_p1;
    char* h = stage_area(ctx, up_bytes + 2 * b_m);
    if (h) {
        memcpy(h, des0, (size_t)n0 * 32);
        memcpy(h + b_d0, des1, (size_t)n1 * 32);
        if (gather) {
            memcpy(h + b_d0 + b_d1, pts0, (size_t)n0 * 16);
            memcpy(h + b_d0 + b_d1 + b_p0, pts1, (size_t)n1 * 16);
        }
        SLAM_HIP(ctx, hipMemcpyAsync(d_d0, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        SLAM_HIP(ctx, hipMemcpyAsync(d_d0, des0, (size_t)n0 * 32, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(d_d1, des1, (size_t)n1 * 32, hipMemcpyHostToDevice, ctx->stream));
        if (gather) {
            SLAM_HIP(ctx, hipMemcpyAsync(d_p0, pts0, (size_t)n0 * 16, hipMemcpyHostToDevice, ctx->stream));
            SLAM_HIP(ctx, hipMemcpyAsync(d_p1, pts1, (size_t)n1 * 16, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    SLAM_TRY(match_enqueue(ctx, d_d0, d_p0, n0, d_d1, d_p1, n1, slices, d_part, d_m0, d_m1, d_n, d_index, d_distance));
    int32_t* h_n = (int32_t*)(ctx->h_small + 128);
    SLAM_HIP(ctx, hipMemcpyAsync(h_n, d_n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (index) SLAM_HIP(ctx, hipMemcpyAsync(index, d_index, (size_t)n0 * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (distance) SLAM_HIP(ctx, hipMemcpyAsync(distance, d_distance, (size_t)n0 * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (gather && h) SLAM_HIP(ctx, hipMemcpyAsync(h + up_bytes, d_m0, 2 * b_m, hipMemcpyDeviceToHost, ctx->stream));  // m0 | m1
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_matched) *n_matched = *h_n;
    if (gather && *h_n > 0) {
        if (h) {
            memcpy(matched0, h + up_bytes, (size_t)*h_n * 16);
            memcpy(matched1, h + up_bytes + b_m, (size_t)*h_n * 16);
        } else {
            SLAM_HIP(ctx, hipMemcpyAsync(matched0, d_m0, (size_t)*h_n * 16, hipMemcpyDeviceToHost, ctx->stream));
            SLAM_HIP(ctx, hipMemcpyAsync(matched1, d_m1, (size_t)*h_n * 16, hipMemcpyDeviceToHost, ctx->stream));
            SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return SLAM_OK;
}

extern "C" int slam_match_points(slam_ctx* ctx, const uint8_t* des0, const double* pts0, int n0, const uint8_t* des1,
                                 const double* pts1, int n1, double* matched0, double* matched1, int32_t* n_matched,
                                 int32_t* index, int32_t* distance) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, des0 && des1 && pts0 && pts1 && matched0 && matched1 && n_matched, "NULL pointer");
    SLAM_MATCH_ARGS_OK(ctx, n0, n1);
    return match_host(ctx, des0, pts0, n0, des1, pts1, n1, matched0, matched1, n_matched, index, distance);
}

// k nearest for 3 <= k <= SLAM_MATCH_MAX_K: sorted register arrays of four or eight keys
template <int K>
static int match_knnk_host(slam_ctx* ctx, const uint8_t* des0, int n0, const uint8_t* des1, int n1, int k, int32_t* index,
                           int32_t* distance) {
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    const int slices = match_slices(n0, n1);
    const int per_slice = (n1 + slices - 1) / slices;
    const size_t b_part = align256((size_t)n0 * slices * K * sizeof(uint32_t));
    const size_t b_d0 = align256((size_t)n0 * 32), b_d1 = align256((size_t)n1 * 32);
    const size_t b_i = align256((size_t)n0 * k * sizeof(int32_t));
    SLAM_TRY(slam_reserve(ctx, ctx->match, b_part + b_d0 + b_d1 + 2 * b_i));
    char* at = (char*)ctx->match.p;
    auto take = [&](size_t bytes) { char* p = at; at += bytes; return p; };
    uint32_t* d_part = (uint32_t*)take(b_part);
    uint8_t* d_d0 = (uint8_t*)take(b_d0);
    uint8_t* d_d1 = (uint8_t*)take(b_d1);
    int32_t* d_index = (int32_t*)take(b_i);
    int32_t* d_distance = (int32_t*)take(b_i);
    SLAM_HIP(ctx, hipMemcpyAsync(d_d0, des0, (size_t)n0 * 32, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(d_d1, des1, (size_t)n1 * 32, hipMemcpyHostToDevice, ctx->stream));
    ctx->cur = ctx->stream;
    SLAM_TRY(slam_launch(ctx, SLAM_K_MATCH, k_match_knnk<K>, dim3((n0 + 63) / 64, slices), dim3(256), 0,
                         (const uint32_t*)d_d0, (const uint32_t*)d_d1, n0, n1, per_slice, d_part));
    SLAM_TRY(slam_launch(ctx, -1, k_match_finish_k<K>, dim3((n0 + 255) / 256), dim3(256), 0, (const uint32_t*)d_part, slices,
                         n0, k, d_index, d_distance));
    SLAM_HIP(ctx, hipMemcpyAsync(index, d_index, (size_t)n0 * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipMemcpyAsync(distance, d_distance, (size_t)n0 * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_match_knn(slam_ctx* ctx, const uint8_t* des0, int n0, const uint8_t* des1, int n1, int k,
                              int32_t* index, int32_t* distance) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, des0 && des1 && index && distance, "NULL pointer");
    SLAM_REQUIRE(ctx, k >= 1 && k <= SLAM_MATCH_MAX_K, "k must be between 1 and 8");
    SLAM_MATCH_ARGS_OK(ctx, n0, n1);
    if (k > 4) return match_knnk_host<8>(ctx, des0, n0, des1, n1, k, index, distance);
    if (k > 2) return match_knnk_host<4>(ctx, des0, n0, des1, n1, k, index, distance);
    if (k == 2) return match_host(ctx, des0, nullptr, n0, des1, nullptr, n1, nullptr, nullptr, nullptr, index, distance);
    std::vector<int32_t> i2((size_t)n0 * 2), d2((size_t)n0 * 2);
    SLAM_TRY(match_host(ctx, des0, nullptr, n0, des1, nullptr, n1, nullptr, nullptr, nullptr, i2.data(), d2.data()));
    for (int q = 0; q < n0; ++q) { index[q] = i2[2 * (size_t)q]; distance[q] = d2[2 * (size_t)q]; }
    return SLAM_OK;
}

// ================================================================================================
// a frame pair from keypoints and descriptors
// ================================================================================================
extern "C" int slam_frame_pair(slam_ctx* ctx, const uint8_t* des0, const double* pts0, int n0, const uint8_t* des1,
                               const double* pts1, int n1, const double* K, int iterations, double* matched0,
                               double* matched1, int32_t* n_matched, double* F, double* E, double* pose, double* X,
                               uint8_t* good, double* record, int32_t* counts4, int32_t* winner) {
    if (!ctx) return SLAM_EINVAL;
    SLAM_REQUIRE(ctx, des0 && des1 && pts0 && pts1 && K && matched0 && matched1 && n_matched, "NULL pointer");
    SLAM_MATCH_ARGS_OK(ctx, n0, n1);
    SLAM_REQUIRE(ctx, iterations >= 1, "iterations < 1");
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    // match buffer: partial keys | des0 | des1 | pts0 | pts1 | matched0 | matched1 | count
    const int slices = match_slices(n0, n1);
    const size_t b_part = align256((size_t)n0 * slices * 2 * sizeof(uint32_t));
    const size_t b_d0 = align256((size_t)n0 * 32), b_d1 = align256((size_t)n1 * 32);
    const size_t b_p0 = align256((size_t)n0 * 16), b_p1 = align256((size_t)n1 * 16), b_m = align256((size_t)n0 * 16);
    SLAM_TRY(slam_reserve(ctx, ctx->match, b_part + b_d0 + b_d1 + b_p0 + b_p1 + 2 * b_m + 256));
    char* at = (char*)ctx->match.p;
    auto take = [&](size_t bytes) { char* p = at; at += bytes; return p; };
    uint32_t* d_part = (uint32_t*)take(b_part);
    uint8_t* d_d0 = (uint8_t*)take(b_d0);
    uint8_t* d_d1 = (uint8_t*)take(b_d1);
    double* d_p0 = (double*)take(b_p0);
    double* d_p1 = (double*)take(b_p1);
    double* d_m0 = (double*)take(b_m);
    double* d_m1 = (double*)take(b_m);
    int32_t* d_n = (int32_t*)take(256);
    // one upload from the pinned area (d0 | d1 | p0 | p1 as on the device); the matched points come back through it at
    // full capacity together with their count
    const size_t up_bytes = b_d0 + b_d1 + b_p0 + b_p1;
    char* h = stage_area(ctx, up_bytes + 2 * b_m);
    if (h) {
        memcpy(h, des0, (size_t)n0 * 32);
        memcpy(h + b_d0, des1, (size_t)n1 * 32);
        memcpy(h + b_d0 + b_d1, pts0, (size_t)n0 * 16);
        memcpy(h + b_d0 + b_d1 + b_p0, pts1, (size_t)n1 * 16);
        SLAM_HIP(ctx, hipMemcpyAsync(d_d0, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        SLAM_HIP(ctx, hipMemcpyAsync(d_d0, des0, (size_t)n0 * 32, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(d_d1, des1, (size_t)n1 * 32, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(d_p0, pts0, (size_t)n0 * 16, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP(ctx, hipMemcpyAsync(d_p1, pts1, (size_t)n1 * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    SLAM_TRY(match_enqueue(ctx, d_d0, d_p0, n0, d_d1, d_p1, n1, slices, d_part, d_m0, d_m1, d_n, nullptr, nullptr));
    // the number of matches sizes every later launch: the one value that has to come back in the middle
    int32_t* h_n = (int32_t*)(ctx->h_small + 128);
    SLAM_HIP(ctx, hipMemcpyAsync(h_n, d_n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (h) SLAM_HIP(ctx, hipMemcpyAsync(h + up_bytes, d_m0, 2 * b_m, hipMemcpyDeviceToHost, ctx->stream));  // m0 | m1
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int M = *h_n;
    *n_matched = M;
    if (M > 0) {
        if (h) {
            memcpy(matched0, h + up_bytes, (size_t)M * 16);
            memcpy(matched1, h + up_bytes + b_m, (size_t)M * 16);
        } else {
            SLAM_HIP(ctx, hipMemcpyAsync(matched0, d_m0, (size_t)M * 16, hipMemcpyDeviceToHost, ctx->stream));
            SLAM_HIP(ctx, hipMemcpyAsync(matched1, d_m1, (size_t)M * 16, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (M < 8) {  // the reference asserts here (src/epipolar.py:60)
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        snprintf(ctx->err, sizeof ctx->err, "invalid argument: only %d matches pass the ratio test, 8 are needed", M);
        return SLAM_EINVAL;
    }
    return init_pair_on_device(ctx, d_m0, d_m1, M, K, iterations, F, E, pose, X, good, record, counts4, winner);
}

#ifdef SLAM_STAMPS
// diagnostic build only (never declared in the public header)
// starts (on != 0, log cleared) or stops the workgroup log; when stopping, copies up to n entries of four values to
// out and returns their number through *count
extern "C" int slam_debug_wg_log(slam_ctx* ctx, int on, long long* out, int n, int* count) {
    SLAM_TRY(slam_sync_all(ctx));
    unsigned int flag = on ? 1u : 0u, zero = 0u;
    if (on) SLAM_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(slamk::g_wg_log_n), &zero, sizeof zero));
    SLAM_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(slamk::g_wg_log_on), &flag, sizeof flag));
    if (!on && out && count) {
        unsigned int have = 0;
        SLAM_HIP(ctx, hipMemcpyFromSymbol(&have, HIP_SYMBOL(slamk::g_wg_log_n), sizeof have));
        have = std::min<unsigned int>(have, std::min<unsigned int>((unsigned int)n, SLAM_WG_LOG_MAX));
        SLAM_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(slamk::g_wg_log), sizeof(long long) * 4 * (size_t)have));
        *count = (int)have;
    }
    return SLAM_OK;
}

extern "C" int slam_debug_stamps(slam_ctx* ctx, long long* out, int n) {
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SLAM_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(slamk::g_stamps), sizeof(long long) * (size_t)n));
    unsigned long long dbg[8];
    SLAM_HIP(ctx, hipMemcpyFromSymbol(dbg, HIP_SYMBOL(slamk::g_dbg), sizeof dbg));
    fprintf(stderr, "dbg (per wave): projections %llu; (per lane) first-order step refused with angle <= 1e-4: %llu, larger: %llu, bad determinant: %llu; (per wave) sweeps %llu; left after 1/2/3+ sweeps %llu %llu %llu\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]);
    return SLAM_OK;
}
#endif
