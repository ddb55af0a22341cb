// Issue cost of single vector instructions on one CU (shader cycles per instruction per SIMD), with W waves per SIMD and
// four independent chains per lane: FP64 fma / mul / add / max, the FP64 reciprocal, reciprocal square root and square
// root estimates, a 64-bit select (two v_cndmask_b32), the FP32 reciprocal with its two conversions.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/op_rates tools/op_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { FMA, MUL, ADD, MAX, RCP, RSQ, SQRT, SELECT, RCP32, NOPS };
static const char* names[NOPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
                                  "select f64 (2 x v_cndmask_b32)", "cvt + v_rcp_f32 + cvt"};

template <int OP>
__device__ __forceinline__ double apply(double x, double a, double b) {
    if (OP == FMA) return fma(x, a, b);
    if (OP == MUL) return x * a;
    if (OP == ADD) return x + b;
    if (OP == MAX) return fmax(x, b) ;
    if (OP == RCP) return __builtin_amdgcn_rcp(x);
    if (OP == RSQ) return __builtin_amdgcn_rsq(x);
    if (OP == SQRT) return __builtin_amdgcn_sqrt(x);
    if (OP == SELECT) return x > a ? b : x;   // compare + two cndmask: counted as three below
    if (OP == RCP32) return (double)__builtin_amdgcn_rcpf((float)x);
    return x;
}

template <int OP>
__global__ void k(double a, double b, int iters, long long* cycles, double* sink) {
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = apply<OP>(x[i], a, b);
    }
    const long long t1 = clock64();
    double s = x[0] + x[1] + x[2] + x[3];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
}

template <int OP>
void run(int waves, long long* d_cycles, double* d_sink) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, 0.999, 1e-3, iters, d_cycles, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h(16);
    hipMemcpy(h.data(), d_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = (double)iters * 8 * 4 * ((waves + 3) / 4);
    printf("%-34s %2d waves per SIMD: %6.2f cycles per source operation per SIMD\n", names[OP], (waves + 3) / 4, mx / per_simd);
}

int main() {
    long long* d_cycles; double* d_sink;
    hipMalloc(&d_cycles, 16 * sizeof(long long)); hipMalloc(&d_sink, 8);
    for (int waves : {4, 16}) {
        run<FMA>(waves, d_cycles, d_sink); run<MUL>(waves, d_cycles, d_sink); run<ADD>(waves, d_cycles, d_sink);
        run<MAX>(waves, d_cycles, d_sink); run<RCP>(waves, d_cycles, d_sink); run<RSQ>(waves, d_cycles, d_sink);
        run<SQRT>(waves, d_cycles, d_sink); run<SELECT>(waves, d_cycles, d_sink); run<RCP32>(waves, d_cycles, d_sink);
    }
    return 0;
}
