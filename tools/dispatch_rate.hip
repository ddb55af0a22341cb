// Kernel dispatches per second the GPU front end sustains: T host threads, each launching a near-empty kernel on its own
// stream(s) (created at the greatest priority, like the library's), no dependencies between streams.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_bin/dispatch_rate tools/dispatch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_nop(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_busy(double* p, int iters) {   // a few microseconds of work on every CU
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = fma(x, 0.999, 1e-3);
    if (x == 12345.678) p[0] = x;
}

int main() {
    const int launches = 40000;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    double* d; hipMalloc(&d, 8);
    struct Shape { int grid, block, iters; };
    const Shape shapes[] = {{256, 512, 0}, {128, 512, 0}, {256, 64, 0}, {2048, 64, 0}, {4096, 512, 0}, {256, 512, 600}, {128, 512, 600}, {512, 256, 600}, {256, 256, 600}, {256, 64, 600}, {2048, 64, 600},
                            {256, 512, 60}, {256, 512, 2400}, {128, 512, 2400}, {256, 1024, 600}};
    for (const Shape& sh : shapes) {
        const int T = 2, L = 2;
        std::vector<hipStream_t> st(T * L);
        for (auto& s : st) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                while (!go.load()) {}
                for (int i = 0; i < launches; ++i)
                    hipLaunchKernelGGL(k_busy, dim3(sh.grid), dim3(sh.block), 0, st[t * L + i % L], d, sh.iters);
                for (int l = 0; l < L; ++l) hipStreamSynchronize(st[t * L + l]);
            });
        const auto t0 = std::chrono::steady_clock::now();
        go.store(1);
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("grid %4d x %4d threads, %4d dependent FMAs (%.1f us of work per wave): %.2f us per launch with 2 threads x 2 streams\n",
               sh.grid, sh.block, sh.iters, sh.iters * 9.0 / 2350.0, dt / (launches * T) * 1e6);
        for (auto& s : st) hipStreamDestroy(s);
    }
    // tiny grids (8 workgroups of one wave) with a long body: whatever limits their overlap is not a CU resource
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(T);
        for (auto& s : st) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        const int n = 2000;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                while (!go.load()) {}
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_busy, dim3(8), dim3(64), 0, st[t], d, 4000);
                hipStreamSynchronize(st[t]);
            });
        const auto t0 = std::chrono::steady_clock::now();
        go.store(1);
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("8 x 64-thread grids with a long body, %d thread(s) with a stream each: %.2f us per launch\n", T, dt / (n * T) * 1e6);
        for (auto& s : st) hipStreamDestroy(s);
    }
    // the same busy kernel on 6 streams from 3 threads: all at one priority, or two per priority level
    for (int mixed : {0, 1}) {
        const int T = 3, L = 2;
        std::vector<hipStream_t> st(T * L);
        for (int i = 0; i < T * L; ++i) {
            const int prio = mixed ? (i / 2 == 0 ? hi : (i / 2 == 1 ? 0 : lo)) : hi;
            hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, prio);
        }
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                while (!go.load()) {}
                for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_busy, dim3(256), dim3(512), 0, st[t * L + i % L], d, 600);
                for (int l = 0; l < L; ++l) hipStreamSynchronize(st[t * L + l]);
            });
        const auto t0 = std::chrono::steady_clock::now();
        go.store(1);
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("busy 256 x 512 kernel, 3 threads x 2 streams, %s: %.2f us per launch\n",
               mixed ? "one thread per priority level (greatest, default, least)" : "all streams at the greatest priority",
               dt / (launches * T) * 1e6);
        for (auto& s : st) hipStreamDestroy(s);
    }
    for (int busy : {0, 1})
        for (int T : {1, 2, 4})
            for (int L : {1, 2, 4}) {
                std::vector<hipStream_t> st(T * L);
                for (auto& s : st) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
                std::atomic<int> go{0};
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        while (!go.load()) {}
                        for (int i = 0; i < launches; ++i) {
                            hipStream_t s = st[t * L + i % L];
                            if (busy) hipLaunchKernelGGL(k_busy, dim3(256), dim3(512), 0, s, d, 600);
                            else hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (int*)nullptr);
                        }
                        for (int l = 0; l < L; ++l) hipStreamSynchronize(st[t * L + l]);
                    });
                const auto t0 = std::chrono::steady_clock::now();
                go.store(1);
                for (auto& x : th) x.join();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                printf("%s kernels, threads %d x streams %d: %.2f us per launch\n", busy ? "256x512-thread 3-us" : "empty", T, L,
                       dt / (launches * T) * 1e6);
                for (auto& s : st) hipStreamDestroy(s);
            }
    return 0;
}
