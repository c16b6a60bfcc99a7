#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    for (int cfg = 0; cfg < 3; ++cfg) {
        dim3 g = cfg == 0 ? dim3(24, 2) : cfg == 1 ? dim3(32, 4) : dim3(48, 1);
        hipMemset(d, 0xff, 4096 * 4);
        k<<<g, 256>>>(d);
        int h[512]; hipMemcpy(h, d, g.x * g.y * 4, hipMemcpyDeviceToHost);
        printf("grid (%d,%d): ", g.x, g.y);
        for (unsigned i = 0; i < g.x * g.y && i < 40; ++i) printf("%d", h[i]);
        int ok = 1; for (unsigned i = 0; i < g.x * g.y; ++i) ok &= (h[i] == (int)(i % 8));
        printf("  xcc == flat %% 8 for all: %s\n", ok ? "yes" : "NO");
    }
    return 0;
}
