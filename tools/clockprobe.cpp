// Step rate of back-to-back RANSAC bursts over a few seconds, with the shader clock measured between bursts (a
// one-wave kernel on a stream of its own, started in the middle of a burst, that spins for a fixed number of shader cycles and reads the constant 100 MHz clock around it):
// shows whether a change of the step rate under sustained load comes with a change of the clock.
//   clockprobe SECONDS [LANES [pair.bin|- [under-load]]]
// build: hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/_bin/clockprobe tools/clockprobe.cpp -Lslam_amd/lib -lslam_amd -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "slam_amd.h"

__global__ void k_freq(long long spin, long long* out) {
    const long long w0 = wall_clock64(), c0 = clock64();
    while (clock64() - c0 < spin) {}
    const long long c1 = clock64(), w1 = wall_clock64();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int L = argc > 2 ? atoi(argv[2]) : 4;
    const int N = 4096, H = 2048, K = 2000;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> p0(2 * N), p1(2 * N);
    for (int i = 0; i < 2 * N; ++i) { p0[i] = nd(rng); p1[i] = p0[i] + 0.05 * nd(rng); }
    if (argc > 3 && strcmp(argv[3], "-")) {  // "-": the built-in synthetic pair
        FILE* f = fopen(argv[3], "rb");
        if (!f || fread(p0.data(), 8, 2 * N, f) != (size_t)(2 * N) || fread(p1.data(), 8, 2 * N, f) != (size_t)(2 * N)) return 1;
        fclose(f);
    }
    // DUMMY_QUEUES=k: k streams of the default priority are created and used once before the library's context exists
    // (each makes the runtime create a hardware queue), to see how the order in which a process creates its queues
    // changes which of the library's streams can run side by side
    if (const char* dq = getenv("DUMMY_QUEUES")) {
        for (int q = 0; q < atoi(dq); ++q) {
            hipStream_t st;
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            long long* tmp;
            hipMalloc(&tmp, 16);
            hipLaunchKernelGGL(k_freq, dim3(1), dim3(64), 0, st, 10LL, tmp);
            hipStreamSynchronize(st);
        }
    }
    slam_ctx* ctx;
    if (slam_create(0, &ctx)) return 1;
    double *d_p0, *d_p1, *rec;
    long long* d_f;
    hipMalloc(&d_p0, p0.size() * 8); hipMalloc(&d_p1, p1.size() * 8); hipMalloc(&rec, 4 * 12 * 8); hipMalloc(&d_f, 16);
    hipMemcpy(d_p0, p0.data(), p0.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_p1, p1.data(), p1.size() * 8, hipMemcpyHostToDevice);
    slam_set_concurrency(ctx, L);
    slam_job* jobs[4];   // one prepared call per lane, each with a record and an inlier mask of its own (as bench.py has them)
    unsigned char* mask;
    hipMalloc(&mask, 4 * N);
    for (int l = 0; l < 4; ++l)
        if (slam_ransac_job_create(ctx, d_p0, d_p1, N, nullptr, SLAM_SAMPLER_CPYTHON, 0x1337BEEFull, 0, H, 0.005, rec + 12 * l,
                                   mask + (size_t)N * l, nullptr, nullptr, nullptr, &jobs[l])) { printf("%s\n", slam_last_error(ctx)); return 1; }
    // a fifth stream changes how the library's four streams map to the hardware queues (and with it the step rate):
    // the probe runs beside the load only when asked to, otherwise after each burst on the null stream
    const bool under_load = argc > 4 && !strcmp(argv[4], "under-load");
    hipStream_t probe_stream = nullptr;
    if (under_load) hipStreamCreateWithFlags(&probe_stream, hipStreamNonBlocking);
    const auto t_begin = std::chrono::steady_clock::now();
    for (int b = 0;; ++b) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) {
            slam_job_run(jobs[i % L]);
            if (under_load && i == K / 2) hipLaunchKernelGGL(k_freq, dim3(1), dim3(64), 0, probe_stream, 40000LL, d_f);  // runs beside the load
        }
        slam_synchronize(ctx);
        if (under_load) hipStreamSynchronize(probe_stream);
        else hipLaunchKernelGGL(k_freq, dim3(1), dim3(64), 0, 0, 40000LL, d_f);
        const auto t1 = std::chrono::steady_clock::now();
        long long f[2];
        hipMemcpy(f, d_f, 16, hipMemcpyDeviceToHost);
        const double at = std::chrono::duration<double>(t1 - t_begin).count();
        printf("t=%.3f s  burst %3d: %.2f us/step   shader clock %.0f MHz\n", at, b,
               std::chrono::duration<double>(t1 - t0).count() / K * 1e6, (double)f[0] / ((double)f[1] * 10e-9) / 1e6);
        if (at > seconds) break;
    }
    return 0;
}
