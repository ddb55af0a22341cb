// Duration of EMPTY kernels as the per-launch events (hipExtLaunchKernelGGL start/stop) and a kernel trace see it, by
// grid size, workgroup size, registers and LDS: the part of a short kernel's duration that is not its own work.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/launch_floor tools/launch_floor.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int THREADS, int WAVES_PER_SIMD>
__global__ __launch_bounds__(THREADS, WAVES_PER_SIMD) void k_empty(int* out, int flag) {
    __shared__ char s[LDS_BYTES > 0 ? LDS_BYTES : 1];
    if (flag) {  // never taken: keeps the LDS array and a few registers alive
        s[threadIdx.x % (LDS_BYTES > 0 ? LDS_BYTES : 1)] = (char)threadIdx.x;
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x * 7) % (LDS_BYTES > 0 ? LDS_BYTES : 1)];
    }
}

template <typename K>
static void run(const char* name, K kern, int grid, int threads, int* d) {
    const int reps = 300;
    std::vector<hipEvent_t> a(reps), b(reps);
    for (int i = 0; i < reps; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); }
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, 0);
    hipDeviceSynchronize();
    // back to back, as a library call issues them: the queue never runs dry between two launches
    for (int i = 0; i < reps; ++i) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a[i], b[i], 0, d, 0);
    hipDeviceSynchronize();
    double total = 0;
    for (int i = 0; i < reps; ++i) {
        float ms;
        hipEventElapsedTime(&ms, a[i], b[i]);
        total += ms;
        hipEventDestroy(a[i]);
        hipEventDestroy(b[i]);
    }
    printf("%-44s grid %5d x %4d threads: %.2f us per launch\n", name, grid, threads, total / reps * 1e3);
}

int main() {
    int* d;
    hipMalloc(&d, 4096);
    run("no LDS, 8 waves/SIMD", k_empty<0, 512, 8>, 256, 512, d);
    run("35 KB LDS, 4 waves/SIMD (the fused kernel's)", k_empty<35056, 512, 4>, 256, 512, d);
    run("35 KB LDS, 4 waves/SIMD", k_empty<35056, 512, 4>, 128, 512, d);
    run("35 KB LDS, 4 waves/SIMD", k_empty<35056, 512, 4>, 512, 512, d);
    run("no LDS, 4 waves/SIMD", k_empty<0, 512, 4>, 256, 512, d);
    run("no LDS, 256 threads", k_empty<0, 256, 4>, 256, 256, d);
    run("no LDS, 256 threads", k_empty<0, 256, 4>, 512, 256, d);
    run("no LDS, 64 threads", k_empty<0, 64, 4>, 256, 64, d);
    run("no LDS, 64 threads", k_empty<0, 64, 4>, 2048, 64, d);
    run("no LDS, 1024 threads", k_empty<0, 1024, 4>, 128, 1024, d);
    run("no LDS, 1024 threads", k_empty<0, 1024, 4>, 1, 1024, d);
    run("no LDS, 64 threads", k_empty<0, 64, 4>, 1, 64, d);
    return 0;
}
