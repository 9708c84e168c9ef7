// Launch geometry shared by the C ABI (hipcc) and the emulator test library.
#pragma once
#include "tft_kernel.h"
#include "tft_rows_kernel.h"
#include "tft_rows_exact_kernel.h"
#include "f_kernel.h"
#include "f_rows_kernel.h"
#include "gh_kernel.h"
#include "wave_trid.h"
#include "blocks_kernel.h"
#include "pi_kernel.h"
#include "gh_wg_kernel.h"
#include "gh_rows_kernel.h"
#include "optimf_rows_kernel.h"
#include "gh_fp_kernel.h"
#include "pi_wg_kernel.h"
#include "ba_kernel.h"
#include "ba_views_kernel.h"

namespace tff {

// dynamic LDS bytes of the pose kernels for N correspondences
inline size_t pose_lds_bytes(int N, int flags, bool jacobi) {
    size_t d = (size_t)((POSE_LDS_DOUBLES + 1) & ~1);
    if (jacobi) d += (size_t)((JACOBI_LDS_DOUBLES + 1) & ~1);
    if (flags & FLAG_STAGE_LDS) d += 6 * (size_t)N;
    return d * sizeof(double);
}
// ... of the fundamental-matrix pose kernels (9 x 9 exact-tier workspace)
inline size_t f_pose_lds_bytes(int N, int flags, bool jacobi) {
    size_t d = (size_t)((POSE_LDS_DOUBLES + 1) & ~1);
    if (jacobi) d += (size_t)((JACOBI_F_LDS_DOUBLES + 1) & ~1);
    if (flags & FLAG_STAGE_LDS) d += 6 * (size_t)N;
    return d * sizeof(double);
}
// Stage the correspondences in LDS only while that does not cost occupancy: measured on MI355X (tools/bench_n_sweep.py, STAGE=0 / 1)
// re-reading them through L2 / MALL wins from N ~ 220 for the trifocal kernel (26 vs 19 M/s at N = 300, 10.9 vs 4.4 M/s at
// N = 1000, where the staged points would leave two wavefronts per CU) and from N ~ 64 for the fundamental-matrix kernel, whose
// smaller register footprint allows more wavefronts than a staged LDS does (37.9 vs 30.4 M/s at N = 200).
constexpr int STAGE_MAX_N_TFT = 200, STAGE_MAX_N_F = 48;
inline int pose_auto_flags(int N, int flags, bool jacobi, int max_n = STAGE_MAX_N_TFT) {
    if (N <= max_n && pose_lds_bytes(N, flags | FLAG_STAGE_LDS, jacobi) <= 64 * 1024) flags |= FLAG_STAGE_LDS;
    return flags;
}
// Gauss-Helmert kernels: correspondences are re-read through L2 (never staged); the LDS holds
// the GH workspace instead (20 N + ~2.7k doubles for Ressl's 20 parameters / 2 constraints).
template <class Model>
inline size_t gh_lds_bytes(int N, int /*flags*/, bool jacobi) {
    size_t d = (size_t)((POSE_LDS_DOUBLES + 1) & ~1);
    if (jacobi) d += (size_t)((JACOBI_LDS_DOUBLES + 1) & ~1);
    d += (size_t)gh_lds_doubles(Model::U, Model::C, N);
    return d * sizeof(double);
}
// Pi-matrix Gauss-Helmert kernels: xi (6N), per-correspondence W+ / W+w (14N or 20N), 36x37 / 38x39 KKT system
template <class Model>
inline size_t pi_lds_bytes(int N, int /*flags*/, bool jacobi) {
    size_t d = (size_t)((POSE_LDS_DOUBLES + 1) & ~1);
    if (jacobi) d += (size_t)((JACOBI_LDS_DOUBLES + 1) & ~1);
    d += (size_t)pi_lds_doubles(Model::E, Model::C, N, Model::PINV_KKT);
    return d * sizeof(double);
}
// OptimFPoseEstimation: xi (4N) + the 11 x 11 KKT workspace
inline size_t optimf_lds_bytes(int N, int flags, bool jacobi) {
    size_t d = (size_t)((POSE_LDS_DOUBLES + 1) & ~1);
    if (jacobi) d += (size_t)((JACOBI_F_LDS_DOUBLES + 1) & ~1);
    d += (size_t)((OPTIMF_FIXED_DOUBLES + 1) & ~1) + 4 * (size_t)N + 2;
    if (flags & FLAG_STAGE_LDS) d += 6 * (size_t)N;
    return d * sizeof(double);
}
// Batches of fewer correspondences per triplet than this go to the exact kernel as a whole (minimal samples: the two smallest singular
// values of the design matrix nearly coincide too often for the flag-and-redo path to pay).  TFF_OPT_EXACT_BELOW overrides.
constexpr int EXACT_BELOW_N = 12;
// one workgroup per triplet: the hardware dispatcher balances better than a persistent grid with a stride loop (measured round 3, 10 000 x 200:
// 0.420 ms against 0.439 ms with 2048 workgroups, 0.426 with 4096, 0.786 with 1024; LinearF 0.332 against 0.492 / 0.357 / 0.703)
inline unsigned pose_grid(long B) { return (unsigned)((B < (1L << 30)) ? (B > 0 ? B : 1) : (1L << 30)); }

}  // namespace tff
