// Wavefront-level primitives for the one-wavefront-per-correspondence-set
// kernels (gfx950: 64 lanes, LDS shared by the lanes of a wave).
//
// Every kernel in this directory is written against this small vocabulary:
//   lane_id, wave_sync, wave_bcast, wave_sum / wave_max / wave_sum_i,
//   wave_any, wave_shfl_xor, TFF_DYNAMIC_LDS.
// The only place the build target shows through is the include below: the
// GPU-less unit tests compile the same kernels against tests/emu/hip_emu.h
// (a thread-per-lane emulator, test infrastructure only).
#pragma once
#ifdef TFF_CPU_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace tff {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_in_block() { return (int)(threadIdx.x >> 6); }

#ifdef TFF_CPU_EMU
#define TFF_DYNAMIC_LDS(type, name) type* name = reinterpret_cast<type*>(emu::dyn_smem())
__device__ inline void wave_sync() { emu::wave_barrier(); }
__device__ inline double wave_shfl_xor(double v, int mask) {
    uint64_t u; std::memcpy(&u, &v, 8);
    u = emu::exchange(u, lane_id() ^ mask);
    std::memcpy(&v, &u, 8); return v;
}
__device__ inline double wave_bcast(double v, int src) {
    uint64_t u; std::memcpy(&u, &v, 8);
    u = emu::exchange(u, src);
    std::memcpy(&v, &u, 8); return v;
}
__device__ inline int wave_shfl_xor_i(int v, int mask) { return (int)emu::exchange((uint64_t)(uint32_t)v, lane_id() ^ mask); }
__device__ inline int wave_bcast_i(int v, int src) { return (int)emu::exchange((uint64_t)(uint32_t)v, src); }
__device__ inline double wave_uniform(double v) { return v; }
__device__ inline int wave_uniform_i(int v) { return v; }
typedef double* lds_ptr;
__device__ inline lds_ptr to_lds(double* p) { return p; }
__device__ inline void sched_fence() {}
#else
#define TFF_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
// Lanes of one wavefront exchange data through LDS without a workgroup
// barrier: DS operations of a wave execute in program order, so only the
// compiler has to be told not to move accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int wave_shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }
// src must be wave-uniform: lowers to v_readlane_b32 pairs (no LDS crossbar).
__device__ __forceinline__ double wave_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// v holds the same value in every lane: move it to scalar registers (v_readfirstlane) so
// that it costs SGPRs, not VGPRs, while it stays live across a per-lane loop.
__device__ __forceinline__ double wave_uniform(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A pointer into LDS that went through a non-inlined call is a generic pointer (flat_load / flat_store); casting it back to the
// local address space restores ds_read / ds_write.
typedef __attribute__((address_space(3))) double* lds_ptr;
__device__ __forceinline__ lds_ptr to_lds(double* p) { return (lds_ptr)p; }
// the instruction scheduler does not move anything across this point
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

// Coarse phase stamps for the *_debug_dev entry points (shader clock, lane 0):
// dbg[80 + slot].  A null dbg (every production entry point) skips them.
__device__ __forceinline__ void phase_stamp(double* dbg, int slot, int writer_lane = 0) {
#ifndef TFF_CPU_EMU
    if (dbg) {
        const long long t = clock64();
        if (lane_id() == writer_lane) dbg[80 + slot] = (double)t;
    }
#else
    (void)dbg; (void)slot; (void)writer_lane;
#endif
}

// Reductions: every lane ends with the same value, summed in a fixed order, so
// results are bit-reproducible run to run.
#ifdef TFF_CPU_EMU
__device__ inline double wave_sum(double v) {
    for (int m = 32; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
    return v;
}
#else
// DPP row operations + two cross-row steps (v_readlane): ~20 VALU instructions
// and no LDS crossbar round trips (the ds_bpermute butterfly costs ~40
// instructions and six dependent LDS latencies).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);     // bound_ctrl: out-of-row sources read 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0x111>(v);            // row_shr:1
    v += dpp_mov<0x112>(v);            // row_shr:2
    v += dpp_mov<0x114>(v);            // row_shr:4
    v += dpp_mov<0x118>(v);            // row_shr:8   -> lane 15 of every row of 16 holds the row sum
    const double r0 = wave_bcast(v, 15), r1 = wave_bcast(v, 31), r2 = wave_bcast(v, 47), r3 = wave_bcast(v, 63);
    return (r0 + r1) + (r2 + r3);
}
#endif
#ifdef TFF_CPU_EMU
__device__ inline double wave_max(double v) {
    for (int m = 32; m >= 1; m >>= 1) { double o = wave_shfl_xor(v, m); v = (o > v) ? o : v; }
    return v;
}
__device__ inline double wave_max32_finite(double v) { return wave_max((lane_id() < 32) ? v : -1.7e308); }
// lowest lane whose predicate holds (64 if none); wave-uniform
__device__ inline int wave_first_lane(bool p) {
    int c = p ? lane_id() : 64;
    for (int m = 32; m >= 1; m >>= 1) { const int o = wave_shfl_xor_i(c, m); c = (o < c) ? o : c; }
    return c;
}
#else
// as dpp_mov, but lanes whose source falls outside the row keep their own value
template <int CTRL>
__device__ __forceinline__ double dpp_mov_keep(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
    double o;
    o = dpp_mov_keep<0x111>(v); v = (o > v) ? o : v;       // row_shr:1
    o = dpp_mov_keep<0x112>(v); v = (o > v) ? o : v;       // row_shr:2
    o = dpp_mov_keep<0x114>(v); v = (o > v) ? o : v;       // row_shr:4
    o = dpp_mov_keep<0x118>(v); v = (o > v) ? o : v;       // row_shr:8   -> lane 15 of every row holds the row maximum
    const double r0 = wave_bcast(v, 15), r1 = wave_bcast(v, 31), r2 = wave_bcast(v, 47), r3 = wave_bcast(v, 63);
    const double a = (r0 > r1) ? r0 : r1, b = (r2 > r3) ? r2 : r3;
    return (a > b) ? a : b;
}
__device__ __forceinline__ int wave_first_lane(bool p) {
    const unsigned long long m = __ballot(p);
    return m ? (__ffsll((long long)m) - 1) : 64;
}
// maximum over lanes 0..31 only, v_max_f64 per stage (NaNs are dropped: for pivot searches over finite data)
__device__ __forceinline__ double wave_max32_finite(double v) {
    v = fmax(v, dpp_mov_keep<0x111>(v));
    v = fmax(v, dpp_mov_keep<0x112>(v));
    v = fmax(v, dpp_mov_keep<0x114>(v));
    v = fmax(v, dpp_mov_keep<0x118>(v));
    return fmax(wave_bcast(v, 15), wave_bcast(v, 31));
}
#endif
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += wave_shfl_xor_i(v, m);
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return wave_sum_i(p ? 1 : 0) != 0; }

// ---- lane groups ------------------------------------------------------------------------------
// The lane-sparse stages (27x27 / 15x15 eigen-solves, epipoles, 3x3 SVDs: 27, 15, 6 or 2 busy
// lanes) are written against a lane GROUP: Group<64> is the whole wavefront, Group<32> one half of
// it, so that one wavefront can run those stages for two triplets at once, one per half.
template <int G> struct Group;
template <> struct Group<64> {
    static constexpr int size = 64;
    __device__ static __forceinline__ int lane() { return lane_id(); }           // lane index inside the group
    __device__ static __forceinline__ int index() { return 0; }                  // which group of the wavefront
    __device__ static __forceinline__ double bcast(double v, int src) { return wave_bcast(v, src); }
    __device__ static __forceinline__ double sum(double v) { return wave_sum(v); }
};
template <> struct Group<32> {
    static constexpr int size = 32;
    __device__ static __forceinline__ int lane() { return lane_id() & 31; }
    __device__ static __forceinline__ int index() { return lane_id() >> 5; }
#ifdef TFF_CPU_EMU
    __device__ static inline double bcast(double v, int src) {
        uint64_t u; std::memcpy(&u, &v, 8);
        u = emu::exchange(u, (lane_id() & 32) | src);
        std::memcpy(&v, &u, 8); return v;
    }
    __device__ static inline double sum(double v) {
        for (int m = 16; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
        return v;
    }
#else
    // per-half broadcast: the source differs between the halves, so it goes through the LDS crossbar
    __device__ static __forceinline__ double bcast(double v, int src) { return __shfl(v, (lane_id() & 32) | src, 64); }
    __device__ static __forceinline__ double sum(double v) {
        v += dpp_mov<0x111>(v);        // row_shr:1
        v += dpp_mov<0x112>(v);        // row_shr:2
        v += dpp_mov<0x114>(v);        // row_shr:4
        v += dpp_mov<0x118>(v);        // row_shr:8 -> lane 15 of every 16-lane row holds its row sum
        const int h = lane_id() & 32;
        return __shfl(v, h | 15, 64) + __shfl(v, h | 31, 64);
    }
#endif
};

// Reduce K (= 32) per-lane values over the 64 lanes with a halving
// ("transposing") butterfly: K-1+1 exchanges instead of 6K.  On return lane l
// holds the full 64-lane sum of value index  reduce32_index(l).
#ifdef TFF_CPU_EMU
__device__ inline double xchg_sum(double lo_val, double hi_val, int mask) {   // lane keeps one, sends the other to lane ^ mask
    const bool up = (lane_id() & mask) != 0;
    const double keep = up ? hi_val : lo_val, send = up ? lo_val : hi_val;
    return keep + wave_shfl_xor(send, mask);
}
template <int MASK> __device__ inline double halve_sum(double a, double b) { return xchg_sum(a, b, MASK); }
#else
// v_permlane32_swap / v_permlane16_swap (gfx950) do the keep/send exchange of one halving step in place:
// swap(a, b) -> r0 = {a on the lower half-blocks, b's lower half-blocks moved up}, r1 = {a's upper half-blocks moved
// down, b on the upper half-blocks}; r0 + r1 is a[l] + a[l ^ mask] where bit `mask` of l is clear and
// b[l] + b[l ^ mask] where it is set.  No select, no LDS crossbar.
template <int MASK>
__device__ __forceinline__ double halve_sum(double a, double b) {
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (MASK == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    } else if constexpr (MASK == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const bool up = (lane_id() & MASK) != 0;
        const double keep = up ? b : a, send = up ? a : b;
        double got;
        if constexpr (MASK == 8) got = dpp_mov<0x128>(send);                 // row_ror:8 = lane ^ 8 inside a row of 16
        else if constexpr (MASK == 4) { const double dn = dpp_mov<0x114>(send), upv = dpp_mov<0x104>(send); got = up ? dn : upv; }   // row_shr:4 / row_shl:4
        else if constexpr (MASK == 2) got = dpp_mov<0x4E>(send);             // quad_perm [2,3,0,1]
        else got = dpp_mov<0xB1>(send);                                      // quad_perm [1,0,3,2]
        return keep + got;
    }
}
#endif
template <int K>
__device__ __forceinline__ double wave_reduce_scatter(double (&v)[K]) {
    static_assert(K == 32, "tuned for 32 values on a 64-lane wave");
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = halve_sum<32>(v[i], v[i + 16]);      // 32 -> 16 values
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = halve_sum<16>(v[i], v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = halve_sum<8>(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = halve_sum<4>(v[i], v[i + 2]);
    v[0] = halve_sum<2>(v[0], v[1]);
#ifdef TFF_CPU_EMU
    v[0] += wave_shfl_xor(v[0], 1);
#else
    v[0] += dpp_mov<0xB1>(v[0]);
#endif
    return v[0];
}
// value index owned by `lane` after wave_reduce_scatter<32>
__device__ __forceinline__ int reduce32_index(int lane) {
    return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

}  // namespace tff
