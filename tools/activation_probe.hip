// What does the first launch on a stream cost after the device has been idle?  L streams, one kernel each, from an idle
// device, against the same L kernels on ONE stream; empty kernels and kernels that spin for a given time.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/activation_probe tools/activation_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_log(long long ticks, long long* log) {   // as spin; workgroup 0 logs its start and end (constant clock)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (blockIdx.x == 0 && threadIdx.x == 0) { log[0] = t0; log[1] = wall_clock64(); }
}

__global__ void spin(long long ticks, int* sink) {   // ticks of the constant 100 MHz clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (ticks < 0) *sink = 1;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    int* sink = nullptr;
    CK(hipMalloc(&sink, 4));
    for (int grid : {1, 256}) {
        for (long long ticks : {0LL, 900LL}) {   // 0: empty; 900: 9 us
            for (int mode = 0; mode < 2; ++mode) {      // 0: L streams; 1: one stream
                for (int L = 1; L <= 4; ++L) {
                    std::vector<double> t;
                    for (int rep = 0; rep < 40; ++rep) {
                        CK(hipDeviceSynchronize());
                        const double t0 = now_us();
                        for (int k = 0; k < L; ++k) hipLaunchKernelGGL(spin, dim3(grid), dim3(512), 0, st[mode ? 0 : k], ticks, sink);
                        for (int k = 0; k < (mode ? 1 : L); ++k) CK(hipStreamSynchronize(st[k]));
                        t.push_back(now_us() - t0);
                    }
                    std::sort(t.begin(), t.end());
                    printf("grid %3d x 512, kernel %4.1f us, %d kernel(s) on %s: median %6.1f us (min %6.1f)\n", grid, ticks / 100.0, L,
                           mode ? "one stream " : "L streams  ", t[t.size() / 2], t[0]);
                }
            }
        }
    }
    // how the host waits for L streams that each ran one 9 us kernel: (0) a synchronise per stream, (1) one
    // hipDeviceSynchronize, (2) an event per stream that stream 0 waits for, then one synchronise of stream 0,
    // (3) as (0) with a second synchronise of every (now idle) stream
    hipEvent_t ev[4];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int how = 0; how < 8; ++how) {
        for (int L = 1; L <= 4; ++L) {
            std::vector<double> t;
            for (int rep = 0; rep < 40; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now_us();
                for (int k = 0; k < L; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(512), 0, st[k], 900LL, sink);
                if (how == 0 || how == 3) {
                    for (int k = 0; k < L; ++k) CK(hipStreamSynchronize(st[k]));
                    if (how == 3) for (int k = 0; k < L; ++k) CK(hipStreamSynchronize(st[k]));
                } else if (how == 1) {
                    CK(hipDeviceSynchronize());
                } else if (how == 6 || how == 7) {
                    // an event behind the last kernel of every stream (the markers travel at the same time), then wait
                    // for each event (7: and a hipDeviceSynchronize afterwards, which should find nothing to do)
                    for (int k = 0; k < L; ++k) CK(hipEventRecord(ev[k], st[k]));
                    for (int k = 0; k < L; ++k) CK(hipEventSynchronize(ev[k]));
                    if (how == 7) CK(hipDeviceSynchronize());
                } else if (how == 4 || how == 5) {
                    // poll every stream without blocking until all report completion (then, 5: synchronise each, which
                    // should find nothing left to wait for)
                    bool done[4] = {false, false, false, false};
                    int left = L;
                    while (left > 0)
                        for (int k = 0; k < L; ++k)
                            if (!done[k] && hipStreamQuery(st[k]) == hipSuccess) { done[k] = true; --left; }
                    if (how == 5) for (int k = 0; k < L; ++k) CK(hipStreamSynchronize(st[k]));
                } else {
                    for (int k = 1; k < L; ++k) { CK(hipEventRecord(ev[k], st[k])); CK(hipStreamWaitEvent(st[0], ev[k], 0)); }
                    CK(hipStreamSynchronize(st[0]));
                }
                t.push_back(now_us() - t0);
            }
            std::sort(t.begin(), t.end());
            const char* names[] = {"a synchronise per stream", "one hipDeviceSynchronize", "events joined on stream 0", "two synchronises per stream",
                                   "polling hipStreamQuery", "polling, then synchronise", "an event per stream", "events, then device sync"};
            printf("wait by %-28s, %d stream(s) with a 9 us kernel each: median %6.1f us (min %6.1f)\n", names[how], L, t[t.size() / 2], t[0]);
        }
    }
    // timeline of four 9 us kernels on four streams from an idle device: host time after each launch call, and every
    // kernel's start and end on the device (events attached to the launch), relative to the first kernel's start
    hipEvent_t es[8], ee[8];
    for (auto& e : es) CK(hipEventCreate(&e));
    for (auto& e : ee) CK(hipEventCreate(&e));
    for (int per_stream = 1; per_stream <= 2; ++per_stream) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            double host[8];
            const double t0 = now_us();
            for (int k = 0; k < 4 * per_stream; ++k) {
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(512), 0, st[k % 4], es[k], ee[k], 0, 900LL, sink);
                host[k] = now_us() - t0;
            }
            for (int k = 0; k < 4; ++k) CK(hipStreamSynchronize(st[k]));
            const double t_end = now_us() - t0;
            printf("%d per stream | host after launch:", per_stream);
            for (int k = 0; k < 4 * per_stream; ++k) printf(" %5.1f", host[k]);
            printf(" | all done %5.1f | device start-end:", t_end);
            for (int k = 0; k < 4 * per_stream; ++k) {
                float a = 0, b = 0;
                CK(hipEventElapsedTime(&a, es[0], es[k]));
                CK(hipEventElapsedTime(&b, es[0], ee[k]));
                printf(" %5.1f-%5.1f", a * 1e3, b * 1e3);
            }
            printf("\n");
        }
    }
    // the same without events: plain launches, the device side from stamps the kernels write themselves
    long long* log = nullptr;
    CK(hipMalloc(&log, 16 * 8 * 2));
    long long hlog[32];
    for (int per_stream = 1; per_stream <= 3; ++per_stream) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            double host[16];
            const double t0 = now_us();
            for (int k = 0; k < 4 * per_stream; ++k) {
                hipLaunchKernelGGL(spin_log, dim3(256), dim3(512), 0, st[k % 4], 900LL, log + 2 * k);
                host[k] = now_us() - t0;
            }
            for (int k = 0; k < 4; ++k) CK(hipStreamSynchronize(st[k]));
            const double t_end = now_us() - t0;
            CK(hipMemcpy(hlog, log, sizeof(long long) * 8 * per_stream, hipMemcpyDeviceToHost));
            printf("plain, %d per stream | host after launch:", per_stream);
            for (int k = 0; k < 4 * per_stream; ++k) printf(" %5.1f", host[k]);
            printf(" | all done %5.1f | device start-end:", t_end);
            for (int k = 0; k < 4 * per_stream; ++k) printf(" %5.1f-%5.1f", (hlog[2 * k] - hlog[0]) / 100.0, (hlog[2 * k + 1] - hlog[0]) / 100.0);
            printf("\n");
        }
    }
    return 0;
}
