// TEST INFRASTRUCTURE ONLY.
//
// A tiny lock-step emulator of the HIP execution model subset that the
// kernels in tft_vs_fund_amd/csrc use: one std::thread per lane, a barrier per
// 64-lane wavefront for the cross-lane primitives, a barrier per workgroup for
// __syncthreads, `static` storage for __shared__.  It exists so that the
// kernel *logic* can be run, debugged and sanitised (ASan/UBSan) in the
// GPU-less build container by `pytest -m "not gpu"`.  It is never part of the
// shipped library: libtftfund.so is built by hipcc for gfx950 only and has no
// CPU path.
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

namespace emu {
struct Idx { unsigned x = 0, y = 0, z = 0; };
struct Block {
    unsigned nthreads = 0;
    std::unique_ptr<std::barrier<>> bar;
    std::vector<std::unique_ptr<std::barrier<>>> wavebar;
    std::vector<uint64_t> slot;
    std::vector<char> dyn;
};
inline thread_local Idx tl_threadIdx, tl_blockIdx, tl_blockDim, tl_gridDim;
inline thread_local Block* tl_block = nullptr;

// One barrier per exchange: the slots are double-buffered by the parity of the lane's exchange count (every lane of a wavefront
// executes the same sequence of exchanges), so slot set p is not rewritten before the barrier of the NEXT exchange, which no lane
// passes until every lane has finished reading set p.
inline thread_local unsigned tl_xchg = 0;
inline uint64_t exchange(uint64_t v, int src) {
    Block& b = *tl_block;
    const unsigned t = tl_threadIdx.x, w = t >> 6, half = (tl_xchg++ & 1u) * b.nthreads;
    b.slot[half + t] = v;
    b.wavebar[w]->arrive_and_wait();
    return b.slot[half + (w << 6) + (unsigned(src) & 63u)];
}
inline void wave_barrier() { tl_block->wavebar[tl_threadIdx.x >> 6]->arrive_and_wait(); }
inline void block_barrier() { tl_block->bar->arrive_and_wait(); }
inline void* dyn_smem() { return tl_block->dyn.data(); }

template <class F, class... A>
void launch(F kernel, unsigned grid, unsigned block, size_t smem, A... args) {
    if (block % 64) { std::fprintf(stderr, "emu: block size must be a multiple of 64\n"); std::abort(); }
    for (unsigned bid = 0; bid < grid; ++bid) {
        Block blk;
        blk.nthreads = block;
        blk.bar = std::make_unique<std::barrier<>>(block);
        for (unsigned w = 0; w < block / 64; ++w) blk.wavebar.push_back(std::make_unique<std::barrier<>>(64));
        blk.slot.assign(2 * block, 0);
        blk.dyn.assign(smem + 16, 0);
        std::vector<std::thread> th;
        th.reserve(block);
        for (unsigned t = 0; t < block; ++t) {
            th.emplace_back([&, t]() {
                tl_threadIdx.x = t;
                tl_blockIdx.x = bid;
                tl_blockDim.x = block;
                tl_gridDim.x = grid;
                tl_block = &blk;
                tl_xchg = 0;
                kernel(args...);
            });
        }
        for (auto& x : th) x.join();
    }
}
}  // namespace emu

#define threadIdx (emu::tl_threadIdx)
#define blockIdx (emu::tl_blockIdx)
#define blockDim (emu::tl_blockDim)
#define gridDim (emu::tl_gridDim)

struct double2 { double x, y; };
inline void __syncthreads() { emu::block_barrier(); }
inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }   // (k_collect_retry: lanes are threads here)
inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
