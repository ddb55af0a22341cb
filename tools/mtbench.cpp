// Aggregate RANSAC step rate when T host threads, each with its own context of L internal streams, enqueue prepared
// jobs on one GPU: separates the host's launch rate from the GPU's own limit.
//   mtbench STEPS [pair.bin|-] [T L [early]]    early: create the contexts before anything else touches the device
// build: hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/_bin/mtbench tools/mtbench.cpp -Lslam_amd/lib -lslam_amd -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "slam_amd.h"

static const int N = 4096, H = 2048;

static int run_config(int T, int L, int steps, const double* d_p0, const double* d_p1, std::vector<slam_ctx*> ctx) {
    std::vector<slam_job*> job(T);
    std::vector<double*> rec(T);
    for (int t = 0; t < T; ++t) {
        slam_set_concurrency(ctx[t], L);
        hipMalloc(&rec[t], 4 * 12 * 8);
        if (slam_ransac_job_create(ctx[t], d_p0, d_p1, N, nullptr, SLAM_SAMPLER_CPYTHON, 0x1337BEEFull, 0, H, 0.005, rec[t],
                                   nullptr, nullptr, nullptr, nullptr, &job[t])) {
            printf("job failed: %s\n", slam_last_error(ctx[t]));
            return 1;
        }
        for (int i = 0; i < (t == 0 ? 30000 : 50); ++i) slam_job_run(job[t]);  // also brings the clocks up
        slam_synchronize(ctx[t]);
    }
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    const auto body = [&](int t) {
        while (!go.load()) {}
        for (int i = 0; i < steps; ++i) slam_job_run(job[t]);
        slam_synchronize(ctx[t]);
    };
    for (int t = 0; t < T; ++t) th.emplace_back(body, t);
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1);
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d x streams %d: %.2f us per step, %.0f M hypotheses/s\n", T, L, dt / (steps * T) * 1e6,
           (double)H * steps * T / dt / 1e6);
    for (int t = 0; t < T; ++t) { slam_job_destroy(job[t]); hipFree(rec[t]); }
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4000;
    const int oneT = argc > 4 ? atoi(argv[3]) : 0, oneL = argc > 4 ? atoi(argv[4]) : 0;
    const bool early = argc > 5 && !strcmp(argv[5], "early");
    std::vector<slam_ctx*> first;
    const auto make = [&](int T, std::vector<slam_ctx*>& v) {
        v.resize(T);
        for (int t = 0; t < T; ++t)
            if (slam_create(0, &v[t])) { printf("create failed: %s\n", slam_last_error(nullptr)); exit(1); }
    };
    if (early) make(oneT, first);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> p0(2 * N), p1(2 * N);
    for (int i = 0; i < 2 * N; ++i) { p0[i] = nd(rng); p1[i] = p0[i] + 0.05 * nd(rng); }
    if (argc > 2 && strcmp(argv[2], "-")) {  // 2N doubles of the first image's normalised matches, then 2N of the second's
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(p0.data(), 8, 2 * N, f) != (size_t)(2 * N) || fread(p1.data(), 8, 2 * N, f) != (size_t)(2 * N)) {
            printf("cannot read %s\n", argv[2]);
            return 1;
        }
        fclose(f);
    }
    double *d_p0, *d_p1;
    hipMalloc(&d_p0, p0.size() * 8); hipMalloc(&d_p1, p1.size() * 8);
    hipMemcpy(d_p0, p0.data(), p0.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_p1, p1.data(), p1.size() * 8, hipMemcpyHostToDevice);
    if (oneT) {
        if (!early) make(oneT, first);
        const int rc = run_config(oneT, oneL, steps, d_p0, d_p1, first);
        for (slam_ctx* c : first) slam_destroy(c);
        return rc;
    }
    for (int T : {1, 2, 4, 8})
        for (int L : {1, 2, 4}) {
            std::vector<slam_ctx*> ctx;
            make(T, ctx);
            if (run_config(T, L, steps, d_p0, d_p1, ctx)) return 1;
            for (slam_ctx* c : ctx) slam_destroy(c);
        }
    return 0;
}
