// Wavefront-level vocabulary of the one-wavefront-per-correspondence-set kernels (gfx950: 64 lanes, LDS shared by the lanes
// of a wave).  Every kernel in this directory is written against it:
//   lane_id, wave_sync, wave_bcast, wave_uniform, wave_sum / wave_max / wave_sum_i, wave_any, wave_first_lane,
//   wave_reduce_scatter, Group<64|32>, TFF_DYNAMIC_LDS, phase_stamp.
// The handful of hardware primitives underneath (cross-lane moves, DPP, permlane swaps, fences, the shader clock) come from
// <wave_target.h>, found on the include path: csrc/wave_target.h for the product (gfx950 builtins, no alternative inside),
// tests/emu/wave_target.h for the GPU-less unit tests (a thread-per-lane emulation of the SAME primitives, so that every
// reduction below runs in the same order and emulated results are bit-comparable with the GPU's).
#pragma once
#include <wave_target.h>
#include <stdint.h>

namespace tff {

constexpr int WAVE = 64;

// (opaque: one workgroup handles one triplet, so what the optimiser hoists out of the kernels' triplet loops -- dozens of lane predicates as
// 64-bit scalar masks -- is used once and has to be spilled to vector lanes and read back: ~1000 v_writelane / v_readlane per triplet in
// k_linear_tft_pose.  A lane index the optimiser cannot see through stays where it is asked for; loops INSIDE a function still see it as invariant.)
__device__ __forceinline__ int lane_id() { return opaque_lane_int((int)(threadIdx.x & 63u)); }
__device__ __forceinline__ int wave_in_block() { return opaque_lane_int((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int thread_in_block() { return opaque_lane_int((int)threadIdx.x); }

// Coarse phase stamps for the *_debug_dev entry points (shader clock, lane 0):
// dbg[80 + slot].  A null dbg (every production entry point) skips them.
__device__ __forceinline__ void phase_stamp(double* dbg, int slot, int writer_lane = 0) {
    if (dbg) {
        const long long t = shader_clock();
        if (lane_id() == writer_lane) dbg[80 + slot] = (double)t;
    }
}

// Reductions: every lane ends with the same value, summed in a fixed order, so results are bit-reproducible run to run.
// DPP row operations + two cross-row steps (v_readlane): ~20 VALU instructions and no LDS crossbar round trips (the
// ds_bpermute butterfly costs ~40 instructions and six dependent LDS latencies).
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0x111>(v);            // row_shr:1
    v += dpp_mov<0x112>(v);            // row_shr:2
    v += dpp_mov<0x114>(v);            // row_shr:4
    v += dpp_mov<0x118>(v);            // row_shr:8   -> lane 15 of every row of 16 holds the row sum
    const double r0 = wave_bcast(v, 15), r1 = wave_bcast(v, 31), r2 = wave_bcast(v, 47), r3 = wave_bcast(v, 63);
    return (r0 + r1) + (r2 + r3);
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
// maximum over lanes 0..31 only, v_max_f64 per stage (NaNs are dropped: for pivot searches over finite data)
__device__ __forceinline__ double wave_max32_finite(double v) {
    v = fmax(v, dpp_mov_keep<0x111>(v));
    v = fmax(v, dpp_mov_keep<0x112>(v));
    v = fmax(v, dpp_mov_keep<0x114>(v));
    v = fmax(v, dpp_mov_keep<0x118>(v));
    return fmax(wave_bcast(v, 15), wave_bcast(v, 31));
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += wave_shfl_xor_i(v, m);
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return wave_vote_any(p); }          // (one v_cmp + a scalar compare on the GPU: wave_target.h)

// every lane: the sum over its row of 16 lanes (DPP row shifts + one DP-ALU row broadcast; nothing leaves the row)
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_mov<0x111>(v);            // row_shr:1
    v += dpp_mov<0x112>(v);            // row_shr:2
    v += dpp_mov<0x114>(v);            // row_shr:4
    v += dpp_mov<0x118>(v);            // row_shr:8   -> position 15 holds the row sum
    return row_bcast<15>(v);
}
// does the predicate hold on any lane of the caller's row of 16 lanes?  (one ballot for the wavefront, each row looks at its own 16 bits)
__device__ __forceinline__ bool row_any(bool p) {
    const unsigned long long m = wave_ballot(p);
    return ((m >> (lane_id() & 48)) & 0xffffull) != 0ull;
}

// ---- lane groups ------------------------------------------------------------------------------
// The lane-sparse stages (27x27 / 15x15 eigen-solves, epipoles, 3x3 SVDs: 27, 15, 6 or 2 busy
// lanes) are written against a lane GROUP: Group<64> is the whole wavefront, Group<32> one half of it.
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
    // per-half broadcast: the source lane differs between the halves, so it goes through the LDS crossbar (half_bcast)
    __device__ static __forceinline__ double bcast(double v, int src) { return half_bcast(v, src); }
    __device__ static __forceinline__ double sum(double v) {
        v += dpp_mov<0x111>(v);        // row_shr:1
        v += dpp_mov<0x112>(v);        // row_shr:2
        v += dpp_mov<0x114>(v);        // row_shr:4
        v += dpp_mov<0x118>(v);        // row_shr:8 -> lane 15 of every 16-lane row holds its row sum
        return half_bcast(v, 15) + half_bcast(v, 31);
    }
};

// One row of 16 lanes: the unit of the four-triplets-per-wavefront kernels (tft_rows_kernel.h).  Every cross-lane operation stays inside the
// row (DPP), so the four rows of a wavefront work on four different problems in the same instruction stream.
template <> struct Group<16> {
    static constexpr int size = 16;
    __device__ static __forceinline__ int lane() { return lane_id() & 15; }
    __device__ static __forceinline__ int index() { return lane_id() >> 4; }
    __device__ static __forceinline__ double sum(double v) { return row_sum16(v); }
};

// Reduce K (= 32) per-lane values over the 64 lanes with a halving ("transposing") butterfly: K exchanges instead of 6K.
// On return lane l holds the full 64-lane sum of value index reduce32_index(l).  One halving step: a lane keeps one of two
// values and adds the other one's copy from lane ^ MASK -- v_permlane32_swap / v_permlane16_swap (swap_sum) for the two widest
// steps, DPP row / quad operations for the rest; no select on the wide steps, no LDS crossbar anywhere.
template <int MASK>
__device__ __forceinline__ double halve_sum(double a, double b) {
    if constexpr (MASK == 32 || MASK == 16) {
        return swap_sum<MASK>(a, b);
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
    v[0] += dpp_mov<0xB1>(v[0]);
    return v[0];
}
// 64 values: one more halving step at either end, every lane ends with the full sum of ONE value (index reduce64_index(lane))
__device__ __forceinline__ double wave_reduce_scatter64(double (&v)[64]) {
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = halve_sum<32>(v[i], v[i + 32]);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = halve_sum<16>(v[i], v[i + 16]);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = halve_sum<8>(v[i], v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = halve_sum<4>(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = halve_sum<2>(v[i], v[i + 2]);
    return halve_sum<1>(v[0], v[1]);
}
__device__ __forceinline__ int reduce64_index(int lane) {
    return ((lane >> 5) & 1) * 32 + ((lane >> 4) & 1) * 16 + ((lane >> 3) & 1) * 8 + ((lane >> 2) & 1) * 4 + ((lane >> 1) & 1) * 2 + (lane & 1);
}
// value index owned by `lane` after wave_reduce_scatter<32>
__device__ __forceinline__ int reduce32_index(int lane) {
    return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

}  // namespace tff
