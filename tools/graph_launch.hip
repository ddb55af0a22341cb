// Host cost of two dependent kernel launches per step against one launch of a two-node graph, over four streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_a(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void k_b(double* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += p[n - 1]; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int L = 4, n = 256 * 512, K = 20000;
    std::vector<hipStream_t> s(L);
    std::vector<double*> buf(L);
    for (int l = 0; l < L; ++l) { CK(hipStreamCreateWithFlags(&s[l], hipStreamNonBlocking)); CK(hipMalloc(&buf[l], n * sizeof(double))); CK(hipMemset(buf[l], 0, n * sizeof(double))); }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = now();
        for (int i = 0; i < K; ++i) {
            const int l = i % L;
            hipLaunchKernelGGL(k_a, dim3(256), dim3(512), 0, s[l], buf[l], n);
            hipLaunchKernelGGL(k_b, dim3(1), dim3(1024), 0, s[l], buf[l], n);
        }
        auto t1 = now();
        CK(hipDeviceSynchronize());
        auto t2 = now();
        printf("two launches per step: enqueue %.2f us/step, drained %.2f us/step\n", us(t0, t1) / K, us(t0, t2) / K);
    }
    std::vector<hipGraphExec_t> ge(L);
    for (int l = 0; l < L; ++l) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[l], hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_a, dim3(256), dim3(512), 0, s[l], buf[l], n);
        hipLaunchKernelGGL(k_b, dim3(1), dim3(1024), 0, s[l], buf[l], n);
        CK(hipStreamEndCapture(s[l], &g));
        CK(hipGraphInstantiate(&ge[l], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = now();
        for (int i = 0; i < K; ++i) { const int l = i % L; CK(hipGraphLaunch(ge[l], s[l])); }
        auto t1 = now();
        CK(hipDeviceSynchronize());
        auto t2 = now();
        printf("one graph launch per step: enqueue %.2f us/step, drained %.2f us/step\n", us(t0, t1) / K, us(t0, t2) / K);
    }
    return 0;
}
