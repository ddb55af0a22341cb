// Cost of the compare-into-exec / masked-accumulate idiom of the scoring loops, by waves per SIMD:
//   A: 8 FMA + [s_mov exec,live ; v_cmpx ; v_fma ; s_mov exec,full]      (what the kernels do)
//   B: 8 FMA + [v_cmpx ; v_fma ; s_mov exec,full]                        (no leading exec write)
//   C: 8 FMA + v_cmp (to an SGPR pair) + v_fma + 2 v_cndmask               (no exec writes at all)
//   D: 10 plain FMA                                                       (the issue floor)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/exec_hazard tools/exec_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(double a, double b, double rc, int iters, long long* cycles, double* sink) {
    double x[8], es = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const unsigned long long full = __ballot(true);
    unsigned long long live = full;
    asm volatile("s_mov_b64 %0, %0" : "+s"(live));
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
            unsigned long long lt;
            if (MODE == 0)
                asm volatile("s_mov_b64 exec, %[live]\n\tv_cmpx_lt_f64_e64 %[lt], |%[r]|, %[rc]\n\tv_fma_f64 %[es], %[r], %[r], %[es]\n\ts_mov_b64 exec, %[full]"
                             : [es] "+v"(es), [lt] "=&s"(lt) : [r] "v"(x[u]), [rc] "s"(rc), [live] "s"(live), [full] "s"(full));
            else if (MODE == 1)
                asm volatile("v_cmpx_lt_f64_e64 %[lt], |%[r]|, %[rc]\n\tv_fma_f64 %[es], %[r], %[r], %[es]\n\ts_mov_b64 exec, %[full]"
                             : [es] "+v"(es), [lt] "=&s"(lt) : [r] "v"(x[u]), [rc] "s"(rc), [full] "s"(full));
            else if (MODE == 2) {
                const bool in = fabs(x[u]) < rc;
                es = fma(in ? x[u] : 0.0, x[u], es);
            } else {
                es = fma(x[u], x[u], es);
                x[u] = fma(x[u], a, b);
            }
        }
    }
    const long long t1 = clock64();
    double s = es;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* what, int waves, long long* d_cycles, double* d_sink) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, 0.999, 1e-3, 0.5, iters, d_cycles, d_sink);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(16);
    (void)hipMemcpy(h.data(), d_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double evals_per_simd = (double)iters * 8 * ((waves + 3) / 4);
    printf("%-58s %2d waves: %6.1f cycles per evaluation and SIMD (40 = ten FP64 instructions at full rate)\n", what, waves,
           mx / evals_per_simd);
}

int main() {
    long long* d_cycles; double* d_sink;
    (void)hipMalloc(&d_cycles, 16 * sizeof(long long)); (void)hipMalloc(&d_sink, 8);
    for (int waves : {4, 8, 16}) {
        run<0>("A  s_mov exec, cmpx, fma, s_mov exec", waves, d_cycles, d_sink);
        run<1>("B  cmpx, fma, s_mov exec", waves, d_cycles, d_sink);
        run<2>("C  cmp + select + fma (no exec writes)", waves, d_cycles, d_sink);
        run<3>("D  ten plain fma", waves, d_cycles, d_sink);
    }
    return 0;
}
