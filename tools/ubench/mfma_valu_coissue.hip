// Microbenchmark: how much VALU issue bandwidth does a co-resident wave keep while its SIMD partner streams f32 MFMAs?
// Block = 8 waves (2 per SIMD).  Waves 0-3 run the MFMA stream (shape selected at run time), waves 4-7 run a chain-free
// VALU FMA loop.  Reports cycles for the VALU waves alone, the MFMA waves alone, and both together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(int mode_mfma, int run_mfma, int run_valu, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
    if (wave < 4) {
        if (run_mfma) {
            if (mode_mfma == 0) {  // 16x16x4, 4 independent accumulators
                f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
                const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
                for (int i = 0; i < iters; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                }
                acc = c0[0] + c1[1] + c2[2] + c3[3];
            } else if (mode_mfma == 1) {  // 32x32x2, 2 accumulators; same FLOPs per iteration as mode 0 (4 x 1024 = 2 x 2048 MACs)
                f32x16 c0, c1;
                for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; }
                const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
                for (int i = 0; i < iters; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                }
                acc = c0[0] + c1[5];
            } else {  // 4x4x1 x16 blocks, 4 accumulators, 16 per iteration (= 4096 MACs too)
                f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
                const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
                        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
                    }
                }
                acc = c0[0] + c1[1] + c2[2] + c3[3];
            }
        }
    } else if (run_valu) {
        float x0 = threadIdx.x, x1 = 1.0f, x2 = 2.0f, x3 = 3.0f, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
        const float m = 1.0000001f, d = 1e-9f;
        for (int i = 0; i < iters; ++i) {  // 32 independent-ish FMAs per iteration (8 chains x 4)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0 = __builtin_fmaf(x0, m, d); x1 = __builtin_fmaf(x1, m, d); x2 = __builtin_fmaf(x2, m, d); x3 = __builtin_fmaf(x3, m, d);
                x4 = __builtin_fmaf(x4, m, d); x5 = __builtin_fmaf(x5, m, d); x6 = __builtin_fmaf(x6, m, d); x7 = __builtin_fmaf(x7, m, d);
            }
        }
        acc = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4);
    unsigned long long h[256 * 8];
    const int iters = 20000;
    const char* names[3] = {"16x16x4 ", "32x32x2 ", "4x4x1x16"};
    for (int mode = 0; mode < 3; ++mode) {
        double res[3][2] = {{0}};
        for (int cfg = 0; cfg < 3; ++cfg) {  // 0: valu alone, 1: mfma alone, 2: both
            const int rm = cfg != 0, rv = cfg != 1;
            for (int rep = 0; rep < 2; ++rep) { k<<<256, 512>>>(mode, rm, rv, iters, out, sink); hipDeviceSynchronize(); }
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
            res[cfg][0] = m / 1024 / iters; res[cfg][1] = v / 1024 / iters;
        }
        printf("%s  per iteration (4096 MACs of MFMA | 32 VALU FMA): mfma alone %.1f cyc, valu alone %.1f cyc;  together: mfma %.1f cyc, valu %.1f cyc"
               "  -> VALU keeps %.0f%% of its rate, MFMA keeps %.0f%%\n", names[mode], res[1][0], res[0][1], res[2][0], res[2][1],
               100 * res[0][1] / res[2][1], 100 * res[1][0] / res[2][0]);
    }
    return 0;
}
