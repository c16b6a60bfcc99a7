// What does the boundary between two dependent launches cost, seen from INSIDE the kernels?  Every workgroup marks its entry and its exit with the 100 MHz
// s_memrealtime counter; a chain of launches runs back to back on one stream and the host prints, per variant, the time from the LAST exit of launch k to the FIRST
// entry of launch k + 1, and the spread of the entries.  Variants: static LDS per workgroup, a VGPR-heavy kernel, kernel-argument bytes, bytes stored per workgroup
// right before the exit, returning atomics at the exit, number of workgroups.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/boundary_gap.hip -o tools/ubench/boundary_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define NL 24          // launches in a chain
#define MAXWG 1024
struct pad_t { float v[64]; };   // 256 bytes of by-value kernel argument
__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int LDS_BYTES, int VREGS, bool PAD>
__global__ void __launch_bounds__(256) kern(unsigned long long* marks, int launch, float* sink, int spin_ticks, int store_floats, int* atom, pad_t pad) {
    __shared__ float lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) marks[(launch * 2 + 0) * MAXWG + blockIdx.x] = t0;
    float r[VREGS];
#pragma unroll
    for (int i = 0; i < VREGS; ++i) r[i] = sink[(blockIdx.x * 256 + threadIdx.x + i * 7) & 65535] + (PAD ? pad.v[i & 63] : 0.0f);
    if (LDS_BYTES > 4) { lds[threadIdx.x] = r[0]; __syncthreads(); r[0] += lds[(threadIdx.x + 1) & 255]; }
    while ((long long)(rt() - t0) < spin_ticks) {
#pragma unroll
        for (int i = 0; i < VREGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < VREGS; ++i) acc += r[i];
    for (int i = threadIdx.x; i < store_floats; i += 256) sink[65536 + (size_t)blockIdx.x * store_floats + i] = acc + i;   // dirty lines right before the exit
    if (atom && threadIdx.x == 0) acc += (float)atomicAdd(atom, 1);
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x == 0) marks[(launch * 2 + 1) * MAXWG + blockIdx.x] = rt();
}

// the shape of an acting launch's env state: 16 lanes of one wave read 10 small arrays at entry (mode bit 0) and write them back right before the exit (bit 1: the same
// lines, bit 2: other lines)
__global__ void __launch_bounds__(256) state_kern(unsigned long long* marks, int launch, double* st, int n, int spin_ticks, int mode) {
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) marks[(launch * 2 + 0) * MAXWG + blockIdx.x] = t0;
    const bool writer = threadIdx.x >= 192 && threadIdx.x < 208;
    const int g = blockIdx.x * 16 + (threadIdx.x & 15);
    double v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = (writer && (mode & 1)) ? st[(size_t)k * n + g] : 1.0;
    while ((long long)(rt() - t0) < spin_ticks) {
#pragma unroll
        for (int k = 0; k < 10; ++k) v[k] = v[k] * 1.0001 + 0.5;
    }
    if (writer && (mode & 6)) {
        double* dst = st + ((mode & 4) ? (size_t)10 * n : 0);
#pragma unroll
        for (int k = 0; k < 10; ++k) dst[(size_t)k * n + g] = v[k];
    }
    if (threadIdx.x == 0) marks[(launch * 2 + 1) * MAXWG + blockIdx.x] = rt();
}
static void run_state(const char* name, int mode, unsigned long long* dmarks, double* st, hipStream_t s) {
    const int wgs = 256;
    std::vector<unsigned long long> h((size_t)NL * 2 * MAXWG);
    for (int rep = 0; rep < 2; ++rep) {
        for (int l = 0; l < NL; ++l) state_kern<<<wgs, 256, 0, s>>>(dmarks, l, st, 4096, 2000, mode);
        hipStreamSynchronize(s);
    }
    hipMemcpy(h.data(), dmarks, h.size() * 8, hipMemcpyDeviceToHost);
    double gap = 0; int n = 0;
    for (int l = 4; l + 1 < NL; ++l) {
        unsigned long long x1 = 0, ne0 = ~0ull;
        for (int w = 0; w < wgs; ++w) { x1 = std::max(x1, h[(size_t)(l * 2 + 1) * MAXWG + w]); ne0 = std::min(ne0, h[(size_t)((l + 1) * 2) * MAXWG + w]); }
        gap += (double)(ne0 - x1) / 100.0; ++n;
    }
    printf("%-64s last exit -> next first entry %5.2f us\n", name, gap / n);
}

template <int LDS_BYTES, int VREGS, bool PAD>
static void run(const char* name, int wgs, int spin_us, int store_floats, bool atomics, unsigned long long* dmarks, float* sink, int* atom, hipStream_t s) {
    pad_t pad; memset(&pad, 0, sizeof(pad));
    std::vector<unsigned long long> h((size_t)NL * 2 * MAXWG);
    for (int rep = 0; rep < 2; ++rep) {
        for (int l = 0; l < NL; ++l) kern<LDS_BYTES, VREGS, PAD><<<wgs, 256, 0, s>>>(dmarks, l, sink, spin_us * 100, store_floats, atomics ? atom : nullptr, pad);
        hipStreamSynchronize(s);
    }
    hipMemcpy(h.data(), dmarks, h.size() * 8, hipMemcpyDeviceToHost);
    double gap = 0, skew = 0, span = 0; int n = 0;
    for (int l = 4; l + 1 < NL; ++l) {
        unsigned long long e0 = ~0ull, e1 = 0, x1 = 0, ne0 = ~0ull;
        for (int w = 0; w < wgs; ++w) {
            e0 = std::min(e0, h[(size_t)(l * 2) * MAXWG + w]); e1 = std::max(e1, h[(size_t)(l * 2) * MAXWG + w]);
            x1 = std::max(x1, h[(size_t)(l * 2 + 1) * MAXWG + w]); ne0 = std::min(ne0, h[(size_t)((l + 1) * 2) * MAXWG + w]);
        }
        gap += (double)(ne0 - x1) / 100.0; skew += (double)(e1 - e0) / 100.0; span += (double)(x1 - e0) / 100.0; ++n;
    }
    printf("%-64s last exit -> next first entry %5.2f us | entry spread %5.2f us | span %6.2f us\n", name, gap / n, skew / n, span / n);
}

int main() {
    unsigned long long* dmarks; hipMalloc(&dmarks, (size_t)NL * 2 * MAXWG * 8);
    float* sink; hipMalloc(&sink, (65536 + (size_t)MAXWG * 16384) * 4); hipMemset(sink, 0, (65536 + (size_t)MAXWG * 16384) * 4);
    int* atom; hipMalloc(&atom, 4); hipMemset(atom, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    run<0, 4, false>("256 WGs, 4 VGPR values, no LDS, 5 us spin", 256, 5, 0, false, dmarks, sink, atom, s);
    run<0, 4, false>("256 WGs, 20 us spin", 256, 20, 0, false, dmarks, sink, atom, s);
    run<0, 4, false>("1024 WGs, 5 us spin", 1024, 5, 0, false, dmarks, sink, atom, s);
    run<0, 4, true>("256 WGs, + 256 B of kernel arguments", 256, 5, 0, false, dmarks, sink, atom, s);
    run<40960, 4, false>("256 WGs, 40 KB LDS", 256, 5, 0, false, dmarks, sink, atom, s);
    run<0, 200, false>("256 WGs, 200 live VGPR values", 256, 5, 0, false, dmarks, sink, atom, s);
    run<0, 4, false>("256 WGs, 1 KB stored per WG before the exit", 256, 5, 256, false, dmarks, sink, atom, s);
    run<0, 4, false>("256 WGs, 16 KB stored per WG before the exit (4 MB)", 256, 5, 4096, false, dmarks, sink, atom, s);
    run<0, 4, false>("256 WGs, 64 KB stored per WG before the exit (16 MB)", 256, 5, 16384, false, dmarks, sink, atom, s);
    run<0, 4, false>("256 WGs, one returning atomic per WG at the exit", 256, 5, 0, true, dmarks, sink, atom, s);
    run<40960, 200, true>("256 WGs, 40 KB LDS + 200 VGPR + 256 B args + 4 KB stores", 256, 20, 1024, false, dmarks, sink, atom, s);
    double* st; hipMalloc(&st, (size_t)20 * 4096 * 8); hipMemset(st, 0, (size_t)20 * 4096 * 8);
    run_state("state: no loads, no stores", 0, dmarks, st, s);
    run_state("state: 10 x 128 B loaded at entry, nothing stored", 1, dmarks, st, s);
    run_state("state: nothing loaded, 10 x 128 B stored at the exit", 2, dmarks, st, s);
    run_state("state: loaded at entry, the SAME lines stored at the exit", 3, dmarks, st, s);
    run_state("state: loaded at entry, OTHER lines stored at the exit", 5, dmarks, st, s);
    return 0;
}
