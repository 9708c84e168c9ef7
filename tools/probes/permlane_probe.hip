// Probe: lane maps of v_permlane32_swap / v_permlane16_swap on gfx950 (hipcc --offload-arch=gfx950 permlane_probe.hip -o permlane_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
    for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * v + l]); printf("\n"); }
    return 0;
}
