// Do FP64 matrix instructions (v_mfma_f64_16x16x4_f64) and FP64 vector FMAs share an execution resource on gfx950?
// One workgroup on one CU; per SIMD either one wave that interleaves both kinds, or two waves with one kind each.
// Prints shader cycles per instruction so that "both together" can be compared with "each alone".
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fp64_coexec tools/fp64_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// MODE 0: FMA only (NF per iteration); 1: MFMA only (NM per iteration); 2: both in one wave's stream;
// 3: waves 0..3 MFMA only, waves 4..7 FMA only (needs 8 waves)
template <int MODE, int NM, int NF>
__global__ __launch_bounds__(512) void k(double a, double b, int iters, long long* cycles, double* sink) {
    double x[8];
    double4_t acc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_f = MODE == 0 || MODE == 2 || (MODE == 3 && wave >= 4);
    double am = a * (threadIdx.x & 15), bm = b * (threadIdx.x >> 4);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (do_m && do_f) {
            // interleave: after each MFMA, NF / NM FMAs
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm, acc[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);  // keep the interleave as written
#pragma unroll
                for (int f = 0; f < NF / NM; ++f) x[(m * (NF / NM) + f) & 7] = fma(x[(m * (NF / NM) + f) & 7], a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm, acc[m & 3], 0, 0, 0);
        } else if (do_f) {
#pragma unroll
            for (int f = 0; f < NF; ++f) x[f & 7] = fma(x[f & 7], a, b);
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[wave] = t1 - t0;
}

template <int MODE, int NM, int NF>
void run(const char* what, int waves, long long* d_cycles, double* d_sink) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((k<MODE, NM, NF>), dim3(1), dim3(64 * waves), 0, 0, 0.999, 1e-3, iters, d_cycles, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h(16);
    hipMemcpy(h.data(), d_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0, mx_m = 0, mx_f = 0;
    for (int w = 0; w < waves; ++w) {
        mx = h[w] > mx ? h[w] : mx;
        if (w < 4) mx_m = h[w] > mx_m ? h[w] : mx_m; else mx_f = h[w] > mx_f ? h[w] : mx_f;
    }
    if (MODE == 3)
        printf("%-44s waves %d: mfma waves %8lld cycles (%.1f per MFMA), fma waves %8lld cycles (%.2f per FMA)\n", what, waves,
               mx_m, (double)mx_m / (iters * NM), mx_f, (double)mx_f / (iters * NF));
    else
        printf("%-44s waves %d: %8lld cycles = %.1f per iteration (%d MFMA + %d FMA per wave, %d wave(s) per SIMD)\n", what,
               waves, mx, (double)mx / iters, MODE == 0 ? 0 : NM, MODE == 1 ? 0 : NF, (waves + 3) / 4);
}

int main() {
    long long* d_cycles; double* d_sink;
    hipMalloc(&d_cycles, 16 * sizeof(long long)); hipMalloc(&d_sink, 8);
    run<0, 4, 32>("fma only, 32 per iteration", 4, d_cycles, d_sink);
    run<0, 4, 32>("fma only, 32 per iteration", 8, d_cycles, d_sink);
    run<1, 4, 32>("mfma only, 4 per iteration", 4, d_cycles, d_sink);
    run<1, 4, 32>("mfma only, 4 per iteration", 8, d_cycles, d_sink);
    run<2, 4, 32>("one stream: 4 mfma + 32 fma", 4, d_cycles, d_sink);
    run<2, 4, 32>("one stream: 4 mfma + 32 fma", 8, d_cycles, d_sink);
    run<2, 4, 16>("one stream: 4 mfma + 16 fma", 4, d_cycles, d_sink);
    run<2, 4, 48>("one stream: 4 mfma + 48 fma", 4, d_cycles, d_sink);
    run<2, 4, 64>("one stream: 4 mfma + 64 fma", 4, d_cycles, d_sink);
    run<3, 4, 32>("split: waves 0-3 mfma x4, waves 4-7 fma x32", 8, d_cycles, d_sink);
    run<3, 4, 64>("split: waves 0-3 mfma x4, waves 4-7 fma x64", 8, d_cycles, d_sink);
    return 0;
}
