// Micro-benchmark behind every "fraction of the fp64 issue limit" in DESIGN.md 4 and bench.py's fp64_valu block: how many cycles does one
// wave64 fp64 VALU instruction occupy a SIMD's issue port?  Independent instruction streams (eight accumulators per lane, no dependence
// between consecutive instructions), 1 / 2 / 4 wavefronts per SIMD, the whole device busy; time by HIP events.
//   v_fma_f64 | v_mul_f64 | v_add_f64 | v_fmac_f64_dpp row_newbcast (DP-ALU DPP, the eigen-solves' cross-lane operand) | v_mov_b32_dpp row_shr (32-bit: a 64-bit row shift is two of them)
// Prints instructions per ns per SIMD and the cycles per instruction that implies at 2.4 GHz (the clock DESIGN.md assumes; the measured shader
// clock is reported beside it: s_memrealtime ticks at 100 MHz, clock64() at the shader clock).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/fp64_issue tools/micro/fp64_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, long long* clk, int reps) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (MODE == 0) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                             "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if constexpr (MODE == 1) {
                asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                             "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if constexpr (MODE == 2) {
                asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                             "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if constexpr (MODE == 3) {
                asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %4, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %6, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(m));
            } else {
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
            }
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int waves_per_simd, int reps) {
    // workgroups of 256 threads: the four wavefronts of a workgroup land on the four SIMDs of one CU, `waves_per_simd` workgroups per CU
    const int simds = 256 * 4;
    const int blocks = 256 * waves_per_simd;
    const int grid = blocks * 4;                                                // wavefronts
    double* out; long long* clk;
    hipMalloc(&out, (size_t)grid * 64 * sizeof(double));
    hipMalloc(&clk, (size_t)grid * 2 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, reps);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * (size_t)grid);
    hipMemcpy(h.data(), clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    cyc /= grid; wall /= grid;
    const double insts = (double)reps * 64.0;                                  // per wavefront
    const double per_simd_per_ns = insts * grid / simds / (ms * 1e6);
    const double shader_ghz = cyc / (wall * 10.0);                              // wall_clock64: 100 MHz -> 10 ns per tick
    printf("%-28s %d wave(s)/SIMD: %7.3f ms, %.4f inst/ns/SIMD -> %.2f cycles/inst at 2.4 GHz; in-kernel: %.2f clock64 ticks per inst per wave, "
           "clock64 runs at %.3f GHz -> %.2f cycles/inst/SIMD at that clock\n",
           name, waves_per_simd, ms, per_simd_per_ns, 2.4 / per_simd_per_ns, cyc / insts, shader_ghz, cyc / insts / waves_per_simd);
    hipFree(out); hipFree(clk);
}

int main() {
    const int reps = 20000;
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, reps);
        run<1>("v_mul_f64", w, reps);
        run<2>("v_add_f64", w, reps);
        run<3>("v_fmac_f64_dpp row_newbcast", w, reps);
        run<4>("v_mov_b32_dpp row_shr:1 (32-bit)", w, reps);
    }
    return 0;
}
