// The linear stage of the iterative trifocal methods (k_gh_linear<false>, gh_wg_kernel.h) with four triplets per wavefront: the data
// passes and linearTFT of tft_rows_kernel.h, one triplet per row of 16 lanes, then the 64-double record the block kernels start from
// (t 27 | a 18 | epipoles 6 | normalisations 9).  A triplet a fast tier cannot finish is marked ST_RETRY for k_gh_linear<true>.
// Reference: TFT_methods/ResslTFTPoseEstimation.m:48-53 (the same lines open the Nordberg / FaugPapa / Pi / PiCol wrappers), linearTFT.m:33-91.
#pragma once
#include "tft_rows_kernel.h"
#include "gh_wg_kernel.h"

namespace tff {

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
            {
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

}  // namespace tff
