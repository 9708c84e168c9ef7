// The linear stage of the iterative trifocal methods (k_gh_linear<false>, gh_wg_kernel.h) with four triplets per wavefront: the data
// passes and linearTFT of tft_rows_kernel.h, one triplet per row of 16 lanes, then the 64-double record the block kernels start from
// (t 27 | a 18 | epipoles 6 | normalisations 9).  A triplet a fast tier cannot finish is marked ST_RETRY for k_gh_linear<true>.
// Reference: TFT_methods/ResslTFTPoseEstimation.m:48-53 (the same lines open the Nordberg / FaugPapa / Pi / PiCol wrappers), linearTFT.m:33-91.
#pragma once
#include "tft_rows_kernel.h"
#include "tft_rows_exact_kernel.h"
#include "gh_wg_kernel.h"

namespace tff {

// PRE: normalisations and moment sums from k_tft_moments (a.pre) instead of the two data passes.
template <bool PRE>
__global__ void __launch_bounds__(64, 2) k_gh_linear_rows(const GhWgArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    const int lane = lane_id();
    const int p = lane & 15, row = lane >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const long b_raw = blk * ROW_TRIPLETS + row;
        const bool valid = b_raw < a.B;                                      // (a tail row repeats the last triplet and stores nothing)
        const long b = valid ? b_raw : a.B - 1;
        RowSrc src;
        src.idx = nullptr; src.pts = a.corresp + b * 6 * (long)N; src.ns = 0; src.sampled = false;
        wave_sync();
        int status = ST_OK;
        if (N < 7) {                                                         // wave-uniform
            status = ST_TOO_FEW;
        } else {
            if constexpr (PRE) {
                rows_load_pre(a.pre, b, w->mom, w->nrm);
            } else {
                double cen[6], nr[9];
                rows_centroids(src, N, cen);
                rows_distances_moments(src, N, cen, w->nrm, nr, w->mom);
            }
            wave_sync();
            const bool ok = rows_linear_tft_middle(w, nullptr, true);
            if (!ok) status = ST_RETRY;
            if (valid && ok) {
                double* r = a.rec + b * GH_REC_DOUBLES;
                r[p] = w->t[p];
                if (p < 11) r[16 + p] = w->t[16 + p];
                r[27 + p] = w->pa[p];
                if (p < 2) r[27 + 16 + p] = w->pa[16 + p];
                if (p < 6) r[45 + p] = w->epi[p];
                if (p < 9) r[51 + p] = w->nrm[p];
            }
        }
        if (p == 0 && valid) { a.status[b] = status; if (a.iter) a.iter[b] = 0; }
    }
}

// The last stage of the iterative trifocal methods (k_gh_finish, gh_wg_kernel.h: transform_TFT, R_t_from_TFT, optional Reconst from the optimised
// tensor) with four triplets per wavefront: the pose tail of tft_rows_kernel.h on the fast tiers, and -- for the wavefronts in which some row
// could not finish or certify its part -- once more on the exact tiers (tft_rows_exact_kernel.h's), storing only those rows.
// Reference: TFT_methods/ResslTFTPoseEstimation.m:96-103 (the same lines close the other iterative wrappers).
__global__ void __launch_bounds__(64, 2) k_gh_finish_rows(const GhWgArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    const int lane = lane_id();
    const int p = lane & 15, row = lane >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    LinearTftArgs la{};                                                      // what rows_pose_tail reads: outputs, flags
    la.corresp = a.corresp; la.calm = a.calm; la.calm_stride = a.calm_stride; la.B = a.B; la.N = a.N; la.flags = a.flags;
    la.Rt2 = a.Rt2; la.Rt3 = a.Rt3; la.T = a.T; la.reconst = a.reconst; la.iter = nullptr; la.status = a.status; la.dbg = nullptr;
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        RowJob j;
        const long b_raw = blk * ROW_TRIPLETS + row;
        j.valid = b_raw < a.B;
        j.b = j.valid ? b_raw : a.B - 1;
        j.bad_index = false; j.dbg = nullptr;
        j.src.idx = nullptr; j.src.pts = a.corresp + j.b * 6 * (long)N; j.src.ns = 0; j.src.sampled = false;
        wave_sync();
        const int s0 = a.status[j.b];
        const bool dead = s0 > 0;                                            // ST_TOO_FEW (or an unresolved retry): no outputs
        if (dead) rows_store_nan(la, j, N);
        w->calm[p] = a.calm[j.b * a.calm_stride + p];
        if (p < 11) w->calm[16 + p] = a.calm[j.b * a.calm_stride + 16 + p];
        w->t[p] = dead ? ((p == 0) ? 1.0 : 0.0) : a.topt[j.b * 27 + p];     // (a dead row works on a harmless tensor and stores nothing)
        if (p < 11) w->t[16 + p] = dead ? 0.0 : a.topt[j.b * 27 + 16 + p];
        if (p < 9) w->nrm[p] = dead ? ((p % 3 == 0) ? 1.0 : 0.0) : a.rec[j.b * GH_REC_DOUBLES + 51 + p];
        wave_sync();
        j.valid = j.valid && !dead;
        rows_transform_tft_inverse(w->t, rt->T1, rt->mats, [w](int v) { return normal_matrix(w->nrm, v); });
        bool ok = rows_rt_prepare<false>(w, rt, nullptr);
        int status = rows_pose_tail<false, false>(la, w, rt, j, N, ok);
        if (wave_any(status == ST_RETRY)) {                                  // a fast tier gave up in some row: the exact tiers, for those rows only
            RowJob jx = j;
            jx.valid = j.valid && status == ST_RETRY;
            wave_sync();
            rows_transform_tft_inverse(w->t, rt->T1, rt->mats, [w](int v) { return normal_matrix(w->nrm, v); });
            ok = rows_rt_prepare<true>(w, rt, nullptr);
            const int sx = rows_pose_tail<false, true>(la, w, rt, jx, N, ok);
            status = (status == ST_RETRY) ? sx : status;
        }
        if (s0 < 0) status = -s0;
        if (p == 0 && j.valid) a.status[j.b] = status;
    }
}

}  // namespace tff
