// Issue cost of the float32 counting loop's ingredients on one CU, W waves per SIMD (shader cycles per instruction per
// SIMD): packed float32 fma chains alone; with compares into scalar pairs counted on the scalar unit; with compares
// into VCC added per lane with carry-in.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/_bin/pk32_rate tools/pk32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int MODE>
__global__ void k(const float* F, float lo, float hi, int iters, long long* cycles, int* sink) {
    f2 a0 = {1.0f + threadIdx.x * 1e-3f, 0.5f}, a1 = {0.25f, threadIdx.x * 1e-2f}, b0 = {0.7f, 0.1f}, b1 = {0.3f, 0.9f};
    float f[4][9];
    for (int h = 0; h < 4; ++h)
        for (int i = 0; i < 9; ++i) f[h][i] = F[h * 12 + i + (threadIdx.x & 0)];
    __shared__ float4 s_m[8][4][2][64];
    for (int e = threadIdx.x; e < 8 * 4 * 2 * 64; e += blockDim.x) (&s_m[0][0][0][0])[e] = make_float4(0.5f + e * 1e-5f, 0.25f, 0.1f, 0.9f);
    const int lane = threadIdx.x & 63;
    f2 cf_lo = {0.f, 0.f}, cf_hi = {0.f, 0.f};
    const float BIG = 1e30f;
    const float c_lo = lo * lo * BIG, c_hi = hi * hi * BIG;
    int cnt_s = 0, cnt_v = 0;
    float mind[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    unsigned bits[4] = {0, 0, 0, 0};
    unsigned long long band = 0;
    f2 acc = {0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            const float4 A = s_m[it & 7][(it >> 3) & 3][0][lane], B = s_m[it & 7][(it >> 3) & 3][1][lane];
            a0 = (f2){A.x, A.y}; a1 = (f2){A.z, A.w}; b0 = (f2){B.x, B.y}; b1 = (f2){B.z, B.w};
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f2 l0 = pk(b0, (f2)(f[h][0]), pk(b1, (f2)(f[h][3]), (f2)(f[h][6])));
            const f2 l1 = pk(b0, (f2)(f[h][1]), pk(b1, (f2)(f[h][4]), (f2)(f[h][7])));
            const f2 l2 = pk(b0, (f2)(f[h][2]), pk(b1, (f2)(f[h][5]), (f2)(f[h][8])));
            const f2 r = pk(l0, a0, pk(l1, a1, l2));
            if (MODE == 0) acc += r;
            if (MODE == 5) {
                const f2 d2 = pk(r, r, (f2)(-lo * lo));
                mind[h] = __builtin_fminf(__builtin_fminf(mind[h], fabsf(d2.x)), fabsf(d2.y));
                bits[h] = __builtin_amdgcn_alignbit(bits[h], __float_as_uint(d2.x), 31);
                bits[h] = __builtin_amdgcn_alignbit(bits[h], __float_as_uint(d2.y), 31);
            }
            if (MODE == 4) {
                const f2 q = r * r;
                f2 slo = pk(q, (f2)(-BIG), (f2)(c_lo)), shi = pk(q, (f2)(-BIG), (f2)(c_hi));
                slo = __builtin_elementwise_min(__builtin_elementwise_max(slo, (f2)(0.f)), (f2)(1.f));
                shi = __builtin_elementwise_min(__builtin_elementwise_max(shi, (f2)(0.f)), (f2)(1.f));
                cf_lo += slo; cf_hi += shi;
            }
            if (MODE == 1 || MODE == 3) {
                const unsigned long long ix = __ballot(fabsf(r.x) < lo), iy = __ballot(fabsf(r.y) < lo);
                const unsigned long long ux = __ballot(fabsf(r.x) < hi), uy = __ballot(fabsf(r.y) < hi);
                cnt_s += __popcll(ix) + __popcll(iy);
                band |= (ux & ~ix) | (uy & ~iy);
            }
            if (MODE == 2) {
                cnt_v += (fabsf(r.x) < lo) + (fabsf(r.y) < lo);
                cnt_s += (fabsf(r.x) < hi) + (fabsf(r.y) < hi);
            }
        }
        if (MODE == 5 && (it & 3) == 3) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                cnt_v += __popc(bits[h]);
                band |= __ballot(mind[h] < 1e-6f);
                mind[h] = 1e30f;
            }
        }
        a0.x += 1e-6f; b1.y -= 1e-6f;   // the inputs change, so that nothing is hoisted
    }
    const long long t1 = clock64();
    if (cf_lo.x + cf_lo.y + cf_hi.x + cf_hi.y == 3.5f) sink[1] = 1;
    if (acc.x + acc.y == 12345.f || cnt_s + cnt_v + (int)band == -7) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
void run(int waves, const float* dF, long long* d_cycles, int* d_sink, const char* name, int valu_per_iter) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, dF, 0.07f, 0.0701f, iters, d_cycles, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h(16);
    hipMemcpy(h.data(), d_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd_iters = (double)iters * ((waves + 3) / 4);
    printf("%-44s %d waves per SIMD: %7.1f cycles per iteration (8 evaluations) per SIMD, %5.2f per vector instruction\n", name,
           (waves + 3) / 4, mx / per_simd_iters, mx / per_simd_iters / valu_per_iter);
}

int main() {
    long long* d_cycles; int* d_sink; float* dF;
    hipMalloc(&d_cycles, 16 * sizeof(long long)); hipMalloc(&d_sink, 8); hipMalloc(&dF, 48 * 4);
    std::vector<float> F(48);
    for (int i = 0; i < 48; ++i) F[i] = 0.01f * (i + 1);
    hipMemcpy(dF, F.data(), 48 * 4, hipMemcpyHostToDevice);
    for (int waves : {4, 8, 16}) {
        run<0>(waves, dF, d_cycles, d_sink, "32 v_pk_fma_f32 + 4 v_pk_add_f32", 36 + 2);
        run<1>(waves, dF, d_cycles, d_sink, "32 v_pk_fma_f32 + 16 v_cmp -> sgpr, scalar count", 48 + 2);
        run<2>(waves, dF, d_cycles, d_sink, "32 v_pk_fma_f32 + 16 v_cmp -> vcc + 16 v_addc", 64 + 2);
        run<3>(waves, dF, d_cycles, d_sink, "the sgpr form with its operands from LDS", 48);
        run<5>(waves, dF, d_cycles, d_sink, "LDS operands, sign bits shifted in, min |d| tracked", 48);
        run<4>(waves, dF, d_cycles, d_sink, "LDS operands, clamped steps added in float32", 52);
    }
    return 0;
}
