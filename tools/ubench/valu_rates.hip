// Microbenchmark: issue cost (cycles per wave-instruction) of the VALU ops the rollout / update kernels lean on,
// for 1 and 2 waves per SIMD (512-thread block = 2 waves/SIMD, 256-thread block = 1 wave/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(int iters, unsigned long long* out, float* sink) {
    float x[8]; double d[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; d[i] = threadIdx.x * 0.5 + i; p[i] = f32x2{(float)i, (float)threadIdx.x}; }
    const float m = 1.0000001f, c = 1e-9f; const double dm = 1.0000001, dc = 1e-9; const f32x2 pm = {m, m}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], m, c);
                if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
                if (OP == 2) d[i] = __builtin_fma(d[i], dm, dc);
                if (OP == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
                if (OP == 4) x[i] = __builtin_amdgcn_rcpf(x[i]);
                if (OP == 5) d[i] = d[i] * dm;
                if (OP == 6) d[i] = d[i] + dc;
                if (OP == 7) { unsigned v = __builtin_bit_cast(unsigned, x[i]); v = __umulhi(v, 0xD2511F53u) ^ v; x[i] = __builtin_bit_cast(float, v); }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0; for (int i = 0; i < 8; ++i) acc += x[i] + (float)d[i] + p[i][0] + p[i][1];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int OP> void run(const char* name, unsigned long long* out, float* sink) {
    unsigned long long h[256 * 8];
    double r[2];
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int threads = cfg ? 512 : 256, iters = 4000;
        for (int rep = 0; rep < 2; ++rep) { k<OP><<<256, threads>>>(iters, out, sink); hipDeviceSynchronize(); }
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += h[b * 8 + w]; ++n; }
        r[cfg] = s / n / iters / 32.0;
    }
    printf("%-14s 1 wave/SIMD: %5.2f cyc per wave-instruction;  2 waves/SIMD: %5.2f (= %.2f per SIMD)\n", name, r[0], r[1], r[1] / 2);
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4);
    run<0>("v_fma_f32", out, sink); run<1>("v_pk_fma_f32", out, sink); run<2>("v_fma_f64", out, sink); run<5>("v_mul_f64", out, sink);
    run<6>("v_add_f64", out, sink); run<3>("v_exp_f32", out, sink); run<4>("v_rcp_f32", out, sink); run<7>("v_mul_hi_u32+xor", out, sink);
    return 0;
}
