// gfx950 (CDNA4) implementation of the hardware primitives underneath csrc/wave.h.  Nothing here has an alternative: the
// GPU-less unit tests put tests/emu/ in front of this directory on the include path and get tests/emu/wave_target.h instead.
#pragma once
#include <hip/hip_runtime.h>

namespace tff {

#define TFF_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]

// Lanes of one wavefront exchange data through LDS without a workgroup barrier: DS operations of a wave execute in program
// order, so only the compiler has to be told not to move accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// every global store this wavefront has issued is complete before a later one (rare clean-up paths that rewrite outputs other lanes stored)
__device__ __forceinline__ void store_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
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
// any lane's value, the source chosen per lane: ds_bpermute (LDS crossbar)
__device__ __forceinline__ double wave_shfl(double v, int src) { return __shfl(v, src, 64); }
// lane (l & 32) | src of the caller's own half-wavefront: the source differs between the halves -> ds_bpermute
__device__ __forceinline__ double half_bcast(double v, int src) { return __shfl(v, ((int)(threadIdx.x & 32u)) | src, 64); }
// v holds the same value in every lane: move it to scalar registers (v_readfirstlane) so that it costs SGPRs, not VGPRs,
// while it stays live across a per-lane loop.
__device__ __forceinline__ double wave_uniform(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A pointer into LDS that went through a non-inlined call is a generic pointer (flat_load / flat_store); casting it back to
// the local address space restores ds_read / ds_write.
typedef __attribute__((address_space(3))) double* lds_ptr;
__device__ __forceinline__ lds_ptr to_lds(double* p) { return (lds_ptr)p; }
// 1 / sqrt(x) for x known to be positive, finite and normal (floored Cholesky pivots): v_rsq_f64 and the correction rsqrt() applies to it, without
// rsqrt()'s special-case test and selects -- three VALU instructions shorter and bit-identical on that domain (tools/micro/rsqrt_pos.hip:
// 0 of 16.8 M inputs over the whole exponent range differ)
__device__ __forceinline__ double rsqrt_pos(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(y0 * -x, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
// sqrt(x) for x >= 0 as x * rsqrt_pos(x): 9 VALU instructions where sqrt() takes 20 (its exponent scaling for subnormal / huge arguments and the
// class test at the end), within 2 ulp of the correctly rounded root; 0 -> 0, NaN -> NaN, +inf -> NaN.  For SUMS of hundreds of distances
// (Normalize2Ddata.m:35 in the row kernels: three roots per two correspondences), where the ulp disappears in the sum's own rounding.
__device__ __forceinline__ double sqrt_nonneg(double x) {
    const double s = x * rsqrt_pos(x);
    return (x == 0.0) ? 0.0 : s;
}
// approximate reciprocal (v_rcp_f64: ~1e-7 relative), for sign / margin tests only
__device__ __forceinline__ double fast_rcp(double v) { return __builtin_amdgcn_rcp(v); }
// a wave-uniform integer the optimiser cannot see through (keeps a loop with a small constant trip count rolled)
__device__ __forceinline__ int opaque_int(int v) { asm volatile("" : "+s"(v)); return v; }
// the same for a per-lane integer: index arithmetic derived from it is redone where it is used instead of being hoisted out of the
// enclosing loop and kept (or spilled) across its whole body
__device__ __forceinline__ int opaque_lane_int(int v) { asm volatile("" : "+v"(v)); return v; }
// the value is computed HERE, in a vector register: keeps the optimiser from sinking the arithmetic that produces it towards a distant use
// (which would stretch the live ranges of all its operands instead)
__device__ __forceinline__ void pin_value(double& v) { asm volatile("" : "+v"(v)); }
// the instruction scheduler does not move anything across this point
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ long long shader_clock() { return clock64(); }
// Where this wavefront runs: HW_ID (hardware register 4) bits [5:4] = SIMD of the CU, bits [19:16] = slot of its workgroup on the CU.
// s_getreg_b32 immediate = (size - 1) << 11 | offset << 6 | register.
__device__ __forceinline__ int hw_simd_id() { return (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); }
__device__ __forceinline__ int hw_workgroup_slot() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4); }
// does the predicate hold on any lane?  wave-uniform
__device__ __forceinline__ bool wave_vote_any(bool p) { return __ballot(p) != 0ull; }
// bit l = the predicate of lane l; wave-uniform
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }
// lowest lane whose predicate holds (64 if none); wave-uniform
__device__ __forceinline__ int wave_first_lane(bool p) {
    const unsigned long long m = __ballot(p);
    return m ? (__ffsll((long long)m) - 1) : 64;
}

// DPP move (v_mov_b32 with a dpp control, both halves of the double): row_shr:n = 0x110 + n, row_shl:n = 0x100 + n,
// row_ror:n = 0x120 + n, quad_perm = 0x00..0xFF.  dpp_mov: lanes whose source falls outside the row of 16 read 0;
// dpp_mov_keep: they keep their own value.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_keep(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// DP-ALU DPP (gfx90a+): a 64-bit VALU operand read from lane J of the caller's row of 16 lanes (row_newbcast:J) inside the instruction.
//   fnmac_row_bcast<J>(acc, src, mul) = fma(src[row | J], -mul, acc)      one v_fmac_f64_dpp
//   row_bcast<J>(v)                   = v[row | J]                        one v_mov_b64_dpp
// WAIT: wait states inserted in front.  A VGPR written by a VALU instruction may be read through DPP two issue slots later at the earliest
// and nothing tells the assembler what precedes an asm statement: 2 is always safe, 0 is for a source the caller knows to be older (the
// statements are volatile: they stay in program order among themselves).
template <int J, int WAIT = 2>
__device__ __forceinline__ double fnmac_row_bcast(double acc, double src, double mul) {
    static_assert(J >= 0 && J < 16 && (WAIT == 0 || WAIT == 2), "lane of a row of 16");
    if constexpr (WAIT == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    else asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    return acc;
}
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
    static_assert(J >= 0 && J < 16, "lane of a row of 16");
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
// v_permlane32_swap / v_permlane16_swap (gfx950) do the keep/send exchange of one halving step in place:
// swap(a, b) -> r0 = {a on the lower half-blocks, b's lower half-blocks moved up}, r1 = {a's upper half-blocks moved down,
// b on the upper half-blocks}; r0 + r1 is a[l] + a[l ^ MASK] where bit MASK of l is clear and b[l] + b[l ^ MASK] where it is set.
template <int MASK>
__device__ __forceinline__ double swap_sum(double a, double b) {
    static_assert(MASK == 32 || MASK == 16, "permlane swaps exist for the two widest steps");
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (MASK == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
}

// The matrix core, fp64: D = A B + C with A 16 x 4, B 4 x 16, C / D 16 x 16 (v_mfma_f64_16x16x4_f64: 32 cycles of the matrix pipe per SIMD, one
// VALU issue slot).  Lane l holds A[l & 15][l >> 4] in `a`, B[l >> 4][l & 15] in `b`, and C / D[(l >> 4) + 4 v][l & 15] in c[v], v = 0 .. 3
// (cdna_hip_programming.md: the f64 form has its own C / D row map).  Used where the path HAS a dense contraction: Gram matrices of per-
// correspondence vectors, sum_i e_i e_i' (gh_kernel.h::StrongGram).
typedef double tff_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_f64_16x16x4(const double a, const double b, double (&c)[4]) {
    tff_f64x4 v = {c[0], c[1], c[2], c[3]};
    v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, v, 0, 0, 0);
    c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
}

}  // namespace tff
