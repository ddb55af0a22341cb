// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side?  Two spinning kernels of one workgroup each:
// in order they take twice the time of one, side by side the time of one.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (cycles < 0) *sink = 1;
}

static double run(hipStream_t s, int n, unsigned flags, int* sink) {
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i)
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, i == 0 ? 0u : flags, 2400000LL, sink);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* sink;
    hipMalloc(&sink, 4);
    run(s, 1, 0, sink);
    printf("one kernel:                        %8.1f us\n", run(s, 1, 0, sink));
    printf("four kernels, in order:            %8.1f us\n", run(s, 4, 0, sink));
    printf("four kernels, hipExtAnyOrderLaunch: %8.1f us\n", run(s, 4, hipExtAnyOrderLaunch, sink));
    return 0;
}
