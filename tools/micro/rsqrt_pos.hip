// Is the select-free 1/sqrt(x) (v_rsq_f64 + the correction of rsqrt(), wave_target.h::rsqrt_pos) bit-identical to rsqrt(x) for positive normal x?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/rsqrt_pos tools/micro/rsqrt_pos.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <cstring>
__device__ __forceinline__ double rsqrt_pos(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(y0 * -x, y0, 1.0);
    const double u = y0 * e;
    const double c = fma(e, 0.375, 0.5);
    return fma(u, c, y0);
}
__global__ void k(const double* x, unsigned long long* diff, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double a = rsqrt(x[i]), b = rsqrt_pos(x[i]);
        if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(diff, 1ull);
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<double> x(n);
    std::mt19937_64 g(1);
    for (int i = 0; i < n; ++i) {                      // random positive normal doubles over the whole exponent range
        uint64_t bits = (g() & 0x000fffffffffffffull) | ((uint64_t)(1 + g() % 2046) << 52);
        memcpy(&x[i], &bits, 8);
    }
    double* dx; unsigned long long* dd; unsigned long long h = 0;
    hipMalloc(&dx, sizeof(double) * n); hipMalloc(&dd, 8);
    hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice); hipMemcpy(dd, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dd, n);
    hipMemcpy(&h, dd, 8, hipMemcpyDeviceToHost);
    printf("%d positive normal inputs over the whole exponent range: %llu results differ from rsqrt()\n", n, h);
    return 0;
}
