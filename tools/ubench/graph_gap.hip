// Does a hipGraph shorten the gap between small dependent kernels?  A chain of N kernels (each ~2 us of dependent FMAs on 256 workgroups), launched (a) on a stream,
// (b) as a captured graph; wall time per kernel from hipEvents.  hipcc -O3 --offload-arch=gfx950 tools/ubench/graph_gap.hip -o tools/ubench/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, int iters) {
    float v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4); hipMemset(d, 0, 256 * 256 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 64;
    for (int iters : {64, 1024}) {
        for (int rep = 0; rep < 3; ++rep) { for (int i = 0; i < N; ++i) k<<<256, 256, 0, s>>>(d, iters); }
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int rep = 0; rep < 20; ++rep) for (int i = 0; i < N; ++i) k<<<256, 256, 0, s>>>(d, iters);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("iters %4d  stream: %.2f us per kernel\n", iters, 1e3 * ms / (20 * N));
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) k<<<256, 256, 0, s>>>(d, iters);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int rep = 0; rep < 20; ++rep) hipGraphLaunch(ge, s);
        hipEventRecord(b, s); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("iters %4d  graph : %.2f us per kernel\n", iters, 1e3 * ms / (20 * N));
    }
    return 0;
}
