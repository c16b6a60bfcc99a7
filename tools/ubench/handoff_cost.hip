// What does a hand-off between the waves of ONE workgroup cost?  (candidate "the row-group kernels' ~20 workgroup barriers replaced by per-wave LDS
// counters where only one consumer wave waits".)  256 workgroups (one per CU), each loops N times over {every wave writes 64 floats to LDS; hand-off; the consumer side
// reads them}: (a) __syncthreads() with 4 / 8 waves, (b) every producer wave bumps an LDS counter (ds_add), ONE consumer wave polls it (ds_read + s_sleep 0) and then
// releases the producers through a second counter (they may not overwrite before the consumer has read).  Cycles per iteration by s_memtime of wave 0, median over workgroups.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/handoff_cost.hip -o tools/ubench/handoff_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long mt() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int WAVES, int MODE>   // MODE 0: no hand-off (the loop's own cost), 1: __syncthreads (one per iteration), 2: two barriers per iteration, 3: LDS counters
__global__ void __launch_bounds__(64 * WAVES) kern(unsigned long long* out, float* sink, int n) {
    __shared__ float buf[WAVES][64];
    __shared__ unsigned cnt[2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    float acc = (float)threadIdx.x;
    const unsigned long long t0 = mt();
    for (int it = 1; it <= n; ++it) {
        buf[w][lane] = acc + (float)it;
        if (MODE == 1 || MODE == 2) __syncthreads();
        if (MODE == 3) {
            if (w != 0) { if (lane == 0) __hip_atomic_fetch_add(&cnt[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else { while (__hip_atomic_load(&cnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(it * (WAVES - 1))) __builtin_amdgcn_s_sleep(0); }
        }
        if (MODE != 3 || w == 0) {
#pragma unroll
            for (int k = 0; k < WAVES; ++k) acc += buf[k][(lane + k) & 63];
        }
        if (MODE == 2) __syncthreads();
        if (MODE == 3) {   // the producers may go on once the consumer has read
            if (w == 0) { if (lane == 0) __hip_atomic_store(&cnt[1], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else { while (__hip_atomic_load(&cnt[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)it) __builtin_amdgcn_s_sleep(0); }
        }
    }
    const unsigned long long t1 = mt();
    if (acc == 1.2345f) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int WAVES, int MODE>
static double run(unsigned long long* d, float* sink, int n) {
    std::vector<unsigned long long> h(256);
    for (int rep = 0; rep < 3; ++rep) {
        kern<WAVES, MODE><<<256, 64 * WAVES>>>(d, sink, n);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[128] / n;
}

int main() {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, 256 * 8); (void)hipMalloc(&sink, 1024);
    const int n = 2000;
    printf("shader-clock cycles (s_memtime) per iteration, median of 256 workgroups, %d iterations\n", n);
    const double a4 = run<4, 0>(d, sink, n), b4 = run<4, 1>(d, sink, n), c4 = run<4, 2>(d, sink, n), e4 = run<4, 3>(d, sink, n);
    const double a8 = run<8, 0>(d, sink, n), b8 = run<8, 1>(d, sink, n), c8 = run<8, 2>(d, sink, n), e8 = run<8, 3>(d, sink, n);
    printf("4 waves: no hand-off %.1f | 1 barrier %.1f (+%.1f) | 2 barriers %.1f (+%.1f per barrier) | LDS counters, one consumer wave %.1f (+%.1f)\n", a4, b4, b4 - a4, c4, (c4 - a4) / 2, e4, e4 - a4);
    printf("8 waves: no hand-off %.1f | 1 barrier %.1f (+%.1f) | 2 barriers %.1f (+%.1f per barrier) | LDS counters, one consumer wave %.1f (+%.1f)\n", a8, b8, b8 - a8, c8, (c8 - a8) / 2, e8, e8 - a8);
    return 0;
}
