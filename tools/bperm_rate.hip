// Throughput of the cross-lane instructions that go through the LDS unit (ds_bpermute_b32, ds_swizzle_b32) against
// DPP moves, per CU, by number of waves that issue them at the same time.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/bperm_rate tools/bperm_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(int iters, long long* cycles, int* sink) {
    int v = threadIdx.x * 7 + 1;
    const int addr = ((threadIdx.x & ~7) | ((threadIdx.x * 5) & 7)) << 2;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) v = __builtin_amdgcn_ds_bpermute(addr, v) + 1;
            if (MODE == 1) v = __builtin_amdgcn_ds_swizzle(v, (3 << 5) | 0x18) + 1;
            if (MODE == 2) v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false) + 1;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    if (v == 123456789) sink[0] = v;
}

int main() {
    long long* d_c; int* d_s;
    hipMalloc(&d_c, 8); hipMalloc(&d_s, 4);
    const int iters = 2000;
    const char* names[] = {"ds_bpermute_b32", "ds_swizzle_b32", "dpp mov"};
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {1, 4, 8, 16}) {
            auto fn = mode == 0 ? k<0> : (mode == 1 ? k<1> : k<2>);
            hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, iters, d_c, d_s);
            long long c; hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
            printf("%-16s %2d waves on one CU: %.1f cycles per instruction and wave (dependent chain), %.2f cycles per instruction per CU\n",
                   names[mode], waves, (double)c / (iters * 16), (double)c / (iters * 16) / waves);
        }
    return 0;
}
