// TEST INFRASTRUCTURE ONLY: the hardware primitives underneath tft_vs_fund_amd/csrc/wave.h, emulated on the thread-per-lane
// emulator (hip_emu.h).  Same names and the same semantics as csrc/wave_target.h -- DPP row / quad controls with their
// out-of-row rule, the permlane swaps, readlane broadcasts -- so that the reductions of wave.h (common code) run in the
// GPU's order and emulated results are bit-comparable with it.  Selected by the include path of tests/emu/emu_build.py.
#pragma once
#include "hip_emu.h"

namespace tff {

#define TFF_DYNAMIC_LDS(type, name) type* name = reinterpret_cast<type*>(emu::dyn_smem())

inline int emu_lane() { return (int)(threadIdx.x & 63u); }
inline uint64_t emu_bits(double v) { uint64_t u; std::memcpy(&u, &v, 8); return u; }
inline double emu_dbl(uint64_t u) { double v; std::memcpy(&v, &u, 8); return v; }

inline void wave_sync() { emu::wave_barrier(); }
inline void store_fence() { emu::wave_barrier(); }
inline double wave_shfl_xor(double v, int mask) { return emu_dbl(emu::exchange(emu_bits(v), emu_lane() ^ mask)); }
inline int wave_shfl_xor_i(int v, int mask) { return (int)emu::exchange((uint64_t)(uint32_t)v, emu_lane() ^ mask); }
inline double wave_bcast(double v, int src) { return emu_dbl(emu::exchange(emu_bits(v), src)); }
inline int wave_bcast_i(int v, int src) { return (int)emu::exchange((uint64_t)(uint32_t)v, src); }
inline double wave_shfl(double v, int src) { return emu_dbl(emu::exchange(emu_bits(v), src)); }
inline double half_bcast(double v, int src) { return emu_dbl(emu::exchange(emu_bits(v), (emu_lane() & 32) | src)); }
inline double wave_uniform(double v) { return v; }
inline int wave_uniform_i(int v) { return v; }
typedef double* lds_ptr;
inline lds_ptr to_lds(double* p) { return p; }
inline double fast_rcp(double v) { return 1.0 / v; }
inline double rsqrt_pos(double v) { return rsqrt(v); }
inline double sqrt_nonneg(double v) { return std::sqrt(v); }
inline int opaque_int(int v) { return v; }
inline int opaque_lane_int(int v) { return v; }
inline int hw_simd_id() { return (int)((threadIdx.x >> 6) & 3u); }
inline int hw_workgroup_slot() { return (int)(blockIdx.x & 15u); }
inline void sched_fence() {}
inline void pin_value(double&) {}
inline long long shader_clock() { return 0; }
inline bool wave_vote_any(bool p) {
    int c = p ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) c |= wave_shfl_xor_i(c, m);
    return c != 0;
}
inline unsigned long long wave_ballot(bool p) {
    uint64_t c = p ? (1ull << emu_lane()) : 0ull;
    for (int m = 32; m >= 1; m >>= 1) c |= emu::exchange(c, emu_lane() ^ m);
    return c;
}
inline int wave_first_lane(bool p) {
    int c = p ? emu_lane() : 64;
    for (int m = 32; m >= 1; m >>= 1) { const int o = wave_shfl_xor_i(c, m); c = (o < c) ? o : c; }
    return c;
}

// source lane of a DPP control for lane l (or -1 when it falls outside the row of 16)
template <int CTRL>
inline int emu_dpp_source(int l) {
    const int row = l & ~15, rl = l & 15;
    if constexpr (CTRL >= 0x111 && CTRL <= 0x11F) { const int s = rl - (CTRL - 0x110); return s >= 0 ? row + s : -1; }          // row_shr:n
    else if constexpr (CTRL >= 0x101 && CTRL <= 0x10F) { const int s = rl + (CTRL - 0x100); return s < 16 ? row + s : -1; }     // row_shl:n
    else if constexpr (CTRL >= 0x121 && CTRL <= 0x12F) { return row + ((rl - (CTRL - 0x120)) & 15); }                           // row_ror:n
    else { static_assert(CTRL >= 0 && CTRL <= 0xFF, "unsupported DPP control"); return (l & ~3) + ((CTRL >> (2 * (l & 3))) & 3); }   // quad_perm
}
template <int CTRL>
inline double dpp_mov(double v) {
    const int s = emu_dpp_source<CTRL>(emu_lane());
    const double got = emu_dbl(emu::exchange(emu_bits(v), s < 0 ? emu_lane() : s));
    return s < 0 ? 0.0 : got;
}
template <int CTRL>
inline double dpp_mov_keep(double v) {
    const int s = emu_dpp_source<CTRL>(emu_lane());
    const double got = emu_dbl(emu::exchange(emu_bits(v), s < 0 ? emu_lane() : s));
    return s < 0 ? v : got;
}
// DP-ALU DPP row_newbcast operands (csrc/wave_target.h): lane J of the caller's row of 16
template <int J, int WAIT = 2>
inline double fnmac_row_bcast(double acc, double src, double mul) {
    const double got = emu_dbl(emu::exchange(emu_bits(src), (emu_lane() & ~15) | J));
    return std::fma(got, -mul, acc);
}
template <int J>
inline double row_bcast(double v) { return emu_dbl(emu::exchange(emu_bits(v), (emu_lane() & ~15) | J)); }
// v_permlane32_swap / v_permlane16_swap followed by the sum of the two results (see csrc/wave_target.h)
template <int MASK>
inline double swap_sum(double a, double b) {
    static_assert(MASK == 32 || MASK == 16, "permlane swaps exist for the two widest steps");
    const bool up = (emu_lane() & MASK) != 0;
    const double keep = up ? b : a, send = up ? a : b;
    const double got = emu_dbl(emu::exchange(emu_bits(send), emu_lane() ^ MASK));
    // r0 + r1: lower half-block a[l] + a[l ^ MASK], upper half-block b[l ^ MASK] + b[l]
    return up ? got + keep : keep + got;
}

// v_mfma_f64_16x16x4_f64 (csrc/wave_target.h): D = A B + C, lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], C / D[(l >> 4) + 4 v][l & 15];
// the four products of an entry are added in k order with fused multiply-adds.
inline void mfma_f64_16x16x4(const double a, const double b, double (&c)[4]) {
    const int l = emu_lane(), col = l & 15, rg = l >> 4;
    double bk[4];
    for (int k = 0; k < 4; ++k) bk[k] = emu_dbl(emu::exchange(emu_bits(b), 16 * k + col));
    for (int v = 0; v < 4; ++v) {
        const int row = rg + 4 * v;
        for (int k = 0; k < 4; ++k) {
            const double ak = emu_dbl(emu::exchange(emu_bits(a), 16 * k + row));
            c[v] = std::fma(ak, bk[k], c[v]);
        }
    }
}

}  // namespace tff
