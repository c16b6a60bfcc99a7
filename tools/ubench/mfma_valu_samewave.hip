// Microbenchmark: does ONE wave overlap its own VALU work with its own f32 MFMAs?  Per loop iteration: 4 independent
// v_mfma_f32_16x16x4_f32 (128 clk of matrix-pipe time) interleaved with K independent VALU FMAs (and, optionally, T v_exp_f32).
// Reports cycles per iteration for K = 0..48 at 1 and 2 waves per SIMD.  Perfect overlap: max(128, VALU time); none: the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int T>
__global__ void __launch_bounds__(512) k(int iters, unsigned long long* out, float* sink) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x + i;
    const float m = 1.0000001f, d = 1e-9f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            if (q == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            if (q == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            if (q == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < K / 4; ++u) x[(q * (K / 4) + u) % 12] = __builtin_fmaf(x[(q * (K / 4) + u) % 12], m, d);
#pragma unroll
            for (int u = 0; u < T / 4; ++u) x[(q * (T / 4) + u + 6) % 12] = __builtin_amdgcn_exp2f(x[(q * (T / 4) + u + 6) % 12]);
            __builtin_amdgcn_sched_barrier(0);   // keep the interleave as written
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = c0[0] + c1[1] + c2[2] + c3[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc += x[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int K, int T>
static void run(unsigned long long* out, float* sink) {
    const int iters = 20000;
    unsigned long long h[256 * 8];
    double r[2];
    for (int occ = 1; occ <= 2; ++occ) {
        for (int rep = 0; rep < 2; ++rep) { k<K, T><<<256, 256 * occ>>>(iters, out, sink); hipDeviceSynchronize(); }
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * occ; ++w) s += h[b * 8 + w];
        r[occ - 1] = s / (256 * 4 * occ) / iters;
    }
    printf("4 MFMA + %2d v_fma + %2d v_exp per iteration: %6.1f clk at 1 wave/SIMD, %6.1f clk at 2 waves/SIMD (per wave)\n", K, T, r[0], r[1]);
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4);
    run<0, 0>(out, sink); run<8, 0>(out, sink); run<16, 0>(out, sink); run<24, 0>(out, sink); run<32, 0>(out, sink); run<48, 0>(out, sink);
    run<0, 8>(out, sink); run<16, 8>(out, sink); run<32, 16>(out, sink);
    return 0;
}
