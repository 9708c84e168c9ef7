// Micro-benchmark: one step of a triangular substitution as (2 x v_readlane + v_fma_f64) against v_fmac_f64_dpp row_newbcast (gfx90a+ DP-ALU DPP).
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/dpp_fmac tools/micro/dpp_fmac.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int J>
__device__ __forceinline__ double fmac_row_bcast(double acc, double src, double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    return acc;
}
template <int J>
__device__ __forceinline__ double fmac_row_bcast_nonop(double acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    return acc;
}
__device__ __forceinline__ double bcast(double v, int src) {
    const unsigned lo = __builtin_amdgcn_readlane((int)__double_as_longlong(v), src);
    const unsigned hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | lo);
}

template <int J, int N, int MODE> struct Steps {
    static __device__ __forceinline__ void run(double& y, const double (&row)[N]) {
        if constexpr (J < N) {
            if constexpr (MODE == 0) { const double yj = bcast(y, J); y = fma(-row[J], yj, y); }
            else if constexpr (MODE == 1) y = fmac_row_bcast<J>(y, y, row[J]);
            else y = fmac_row_bcast_nonop<J>(y, y, row[J]);
            Steps<J + 1, N, MODE>::run(y, row);
        }
    }
};

template <int MODE>
__global__ void k(const double* L, const double* x, double* out, long long* cyc, int reps) {
    constexpr int N = 16;
    const int lane = threadIdx.x & 63, p = lane & 15;
    double row[N];
    for (int j = 0; j < N; ++j) row[j] = (j < p) ? (MODE == 0 ? L[p * N + j] : -L[p * N + j]) : 0.0;   // strictly lower, unit diagonal
    double y = x[p], acc = 0.0;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        double z = y + 1e-3 * acc;
        Steps<0, N, MODE>::run(z, row);
        acc = z;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 16, reps = 2000;
    std::vector<double> L(N * N), x(N);
    srand(1);
    for (auto& v : L) v = (rand() / (double)RAND_MAX - 0.5) * 0.3;
    for (auto& v : x) v = rand() / (double)RAND_MAX;
    double *dL, *dx, *dout; long long* dc;
    const int blocks = 2048;
    hipMalloc(&dL, sizeof(double) * N * N); hipMalloc(&dx, sizeof(double) * N); hipMalloc(&dout, sizeof(double) * 64 * blocks); hipMalloc(&dc, sizeof(long long) * blocks);
    hipMemcpy(dL, L.data(), sizeof(double) * N * N, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), sizeof(double) * N, hipMemcpyHostToDevice);
    std::vector<double> ref;
    for (int mode = 0; mode < 3; ++mode) {
        for (int wpb : {64, 128}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(wpb), 0, 0, dL, dx, dout, dc, reps);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(wpb), 0, 0, dL, dx, dout, dc, reps);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(wpb), 0, 0, dL, dx, dout, dc, reps);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<double> out(64 * blocks); std::vector<long long> cyc(blocks);
            hipMemcpy(out.data(), dout, sizeof(double) * 64 * blocks, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), dc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
            if (mode == 0 && ref.empty()) ref.assign(out.begin(), out.begin() + 64);
            double dmax = 0.0;
            for (int i = 0; i < 64; ++i) dmax = fmax(dmax, fabs(out[i] - ref[i]));
            printf("mode %d (%s) threads/block %d: %.3f ms, %.1f cycles per 16-step solve (wave 0), max |diff to readlane form| %.3e, out[5] %.15g\n", mode,
                   mode == 0 ? "readlane+fma" : mode == 1 ? "fmac_dpp + s_nop 1" : "fmac_dpp, no nop", wpb, ms, (double)cyc[0] / reps, dmax, out[5]);
        }
    }
    return 0;
}
