// Decision gate for the split-bf16 experiment: one 64-wide contraction of 16 rows as grad_kernel does it —
//   f32  : 64 x v_mfma_f32_16x16x4_f32 (4 accumulators)                      + V extra VALU instructions (tanh / loss stand-ins)
//   bf16 : 48 x v_mfma_f32_16x16x32_bf16 (6 products x 2 k-halves x 4 tiles) + V + the split of 16 activations into 3 bf16 each
// at 2 waves per SIMD on every CU, random data.  Prints ns per "tile" for both.   hipcc -O3 --offload-arch=gfx950 bf16x6_vs_f32.hip -o bf16x6_vs_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    bf2 r = __builtin_convertvector(f2{a, b}, bf2);
    return __builtin_bit_cast(unsigned, r);
}

template <int V>
__global__ void __launch_bounds__(512, 2) k_f32(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    float a[16], x[16];
    for (int k = 0; k < 16; ++k) { a[k] = in[lane + 64 * k]; x[k] = in[1024 + lane + 64 * k]; }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v = in[lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + t) & 15], x[k], acc[t], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < V; ++e) v = __builtin_fmaf(v, 1.0001f, 0.5f);
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = acc[k & 3][k >> 2] * 1e-3f + v * 1e-6f;   // next tile's activations depend on this one's result
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v;
}

template <int V>
__global__ void __launch_bounds__(512, 2) k_bf16(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    // pre-split weights: 4 out tiles x 2 k-halves x 3 parts, 8 bf16 per lane each
    bf16x8 w[4][2][3];
    for (int t = 0; t < 4; ++t) for (int s = 0; s < 2; ++s) for (int p = 0; p < 3; ++p) {
        s16x8 r;
        for (int e = 0; e < 8; ++e) r[e] = (short)(__builtin_bit_cast(unsigned, in[(lane + 7 * (t + 4 * s + 8 * p) + e) & 1023]) >> 16);
        w[t][s][p] = __builtin_bit_cast(bf16x8, r);
    }
    float x[16];
    for (int k = 0; k < 16; ++k) x[k] = in[1024 + lane + 64 * k];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v = in[lane];
    for (int it = 0; it < iters; ++it) {
        // split 16 f32 activations into hi / mid / lo bf16, packed as 2 x 8 values per part
        unsigned ph[8], pm[8], pl[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a0 = x[2 * k], a1 = x[2 * k + 1];
            const unsigned h = pk_bf16(a0, a1);
            const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
            const unsigned m = pk_bf16(r0, r1);
            const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
            ph[k] = h; pm[k] = m; pl[k] = pk_bf16(q0, q1);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bh[s] = __builtin_bit_cast(bf16x8, u32x4{ph[4 * s], ph[4 * s + 1], ph[4 * s + 2], ph[4 * s + 3]});
            bm[s] = __builtin_bit_cast(bf16x8, u32x4{pm[4 * s], pm[4 * s + 1], pm[4 * s + 2], pm[4 * s + 3]});
            bl[s] = __builtin_bit_cast(bf16x8, u32x4{pl[4 * s], pl[4 * s + 1], pl[4 * s + 2], pl[4 * s + 3]});
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][2], bh[s], acc[t], 0, 0, 0);   // lo x hi
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][0], bl[s], acc[t], 0, 0, 0);   // hi x lo
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][1], bm[s], acc[t], 0, 0, 0);   // mid x mid
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][1], bh[s], acc[t], 0, 0, 0);   // mid x hi
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][0], bm[s], acc[t], 0, 0, 0);   // hi x mid
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][s][0], bh[s], acc[t], 0, 0, 0);   // hi x hi
        }
#pragma unroll
        for (int e = 0; e < V; ++e) v = __builtin_fmaf(v, 1.0001f, 0.5f);
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = acc[k & 3][k >> 2] * 1e-3f + v * 1e-6f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v;
}

template <typename K>
static float run(K kern, const float* in, float* out, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 512>>>(in, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<256, 512>>>(in, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e6f * ms / iters;   // ns per tile per wave (all waves in parallel)
}

int main() {
    float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int it = 20000;
    printf("ns per 16-row tile contraction (64x64), 2 waves/SIMD, all CUs; V = extra VALU instructions per tile\n");
    printf("V=0   : f32 %.0f   bf16x6 %.0f\n", run(k_f32<0>, in, out, it), run(k_bf16<0>, in, out, it));
    printf("V=64  : f32 %.0f   bf16x6 %.0f\n", run(k_f32<64>, in, out, it), run(k_bf16<64>, in, out, it));
    printf("V=128 : f32 %.0f   bf16x6 %.0f\n", run(k_f32<128>, in, out, it), run(k_bf16<128>, in, out, it));
    return 0;
}
