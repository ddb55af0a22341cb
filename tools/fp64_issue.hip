// Issue rate and dependent latency of v_fma_f64 on one CU: CHAINS independent chains per lane, W waves per workgroup.
// Prints shader-clock cycles per FMA instruction per SIMD and the shader-clock frequency seen by clock64().
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue tools/fp64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void k(double a, double b, int iters, long long* cycles, double* sink) {
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) x[i] = fma(x[i], a, b);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS>
void run(int waves, long long* d_cycles, double* d_sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(1), dim3(64 * waves), 0, 0, 0.999, 1e-3, iters, d_cycles, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(1), dim3(64 * waves), 0, 0, 0.999, 1e-3, iters, d_cycles, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(16);
    hipMemcpy(h.data(), d_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double instr_per_simd = (double)iters * 8 * CHAINS * ((waves + 3) / 4);
    printf("chains %d waves %2d: %8lld cycles, %.2f cycles per FMA per SIMD, clock %.0f MHz\n", CHAINS, waves, mx,
           mx / instr_per_simd, mx / (ms * 1e3));
}

// The whole chip busy with FP64 FMAs for a few milliseconds: the shader clock it sustains (clock64 against the
// constant 100 MHz wall_clock64) and the rate it reaches.
__global__ void k_full(double a, double b, int iters, long long* out, double* sink) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

// The same with operands whose bits keep changing: x <- c - x*x (the logistic map in its chaotic regime, one FMA per
// step) on eight chains per lane with different starts.  Power, and with it the clock the chip sustains, depends on
// how many bits toggle: this is the rate real data can reach, the constant-operand loop above is the ceiling.
__global__ void k_full_chaotic(double c, int iters, long long* out, double* sink) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.1 + 1e-3 * (threadIdx.x + 1) + 0.013 * i + 1e-5 * blockIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(-x[i], x[i], c);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

void full_chip_chaotic(double* d_sink, int threads) {
    const int blocks = 512 * 1024 / threads, iters = 20000;
    long long* d_out;
    hipMalloc(&d_out, blocks * 2 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_full_chaotic, dim3(blocks), dim3(threads), 0, 0, 1.9, iters, d_out, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_full_chaotic, dim3(blocks), dim3(threads), 0, 0, 1.9, iters, d_out, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mhz_lo = 1e9, mhz_hi = 0;
    for (int b = 0; b < blocks; ++b) {
        const double mhz = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
        mhz_lo = mhz < mhz_lo ? mhz : mhz_lo; mhz_hi = mhz > mhz_hi ? mhz : mhz_hi;
    }
    const double flop = 2.0 * 64.0 * iters * blocks * (double)threads;
    printf("full chip, chaotic operands: %d workgroups x %d waves, %.2f ms, %.1f TFLOP/s, shader clock %.0f..%.0f MHz while it ran\n",
           blocks, threads / 64, ms, flop / (ms * 1e-3) / 1e12, mhz_lo, mhz_hi);
}

void full_chip(double* d_sink) {
    const int blocks = 512, iters = 20000;
    long long* d_out;
    hipMalloc(&d_out, blocks * 2 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_full, dim3(blocks), dim3(1024), 0, 0, 0.999, 1e-3, iters, d_out, d_sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_full, dim3(blocks), dim3(1024), 0, 0, 0.999, 1e-3, iters, d_out, d_sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mhz_lo = 1e9, mhz_hi = 0;
    for (int b = 0; b < blocks; ++b) {
        const double mhz = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
        mhz_lo = mhz < mhz_lo ? mhz : mhz_lo; mhz_hi = mhz > mhz_hi ? mhz : mhz_hi;
    }
    const double flop = 2.0 * 64.0 * iters * blocks * 1024.0;
    printf("full chip: %d workgroups x 16 waves, %.2f ms, %.1f TFLOP/s, shader clock %.0f..%.0f MHz while it ran\n", blocks, ms,
           flop / (ms * 1e-3) / 1e12, mhz_lo, mhz_hi);
}

int main() {
    long long* d_cycles; double* d_sink;
    hipMalloc(&d_cycles, 16 * 16 * sizeof(long long)); hipMalloc(&d_sink, 8);
    for (int waves : {1, 4, 8, 16}) {
        run<1>(waves, d_cycles, d_sink);
        run<2>(waves, d_cycles, d_sink);
        run<4>(waves, d_cycles, d_sink);
        run<8>(waves, d_cycles, d_sink);
    }
    full_chip(d_sink);
    full_chip_chaotic(d_sink, 1024);
    full_chip_chaotic(d_sink, 512);
    full_chip(d_sink);
    return 0;
}
