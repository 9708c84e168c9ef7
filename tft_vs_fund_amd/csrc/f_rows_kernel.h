// LinearFPoseEstimation with four triplets per wavefront (one per row of 16 lanes): the layout of tft_rows_kernel.h, whose data passes, eigen-solver,
// cheirality votes, t3 scale and stores it shares.  Per triplet: Normalize2Ddata x3 -> linearF(x1,x2), linearF(x1,x3) [each normalising its inputs
// again, 8-point DLT from the 36 moment sums of its pair, inner de-normalisation, rank-2 projection] -> outer de-normalisation -> E = K' F K ->
// recover_R_t x2 -> t3 scale -> optional Reconst -> T = TFT_from_P.  A triplet a fast tier cannot finish or certify is marked ST_RETRY for
// k_f_pose<true, 0>.
//
// Data passes: centroids | mean distances + the 72 moment sums | votes | t3 scale -- the one-triplet kernel makes ten (two normalisations of two
// passes each, three moment sweeps, two vote passes, scale).  The moments are summed over the CENTRED coordinates and scaled afterwards
// (tft_rows_kernel.h::rows_distances_moments); linearF's own normalisation of the already normalised points (linearF.m:45-46 after
// LinearFPoseEstimation.m:46-48: the identity to rounding -- centroid ~1e-16, scale 1 +- 2 ulp) enters through its scale; its centroid shift,
// sixteen digits below the coordinates, is not applied to the sums.
//
// Reference: F_methods/LinearFPoseEstimation.m:42-109, F_methods/linearF.m:32-62, TFT_methods/TFT_from_P.m:25-33.
#pragma once
#include "tft_rows_kernel.h"
#include "tft_rows_exact_kernel.h"
#include "f_kernel.h"

namespace tff {

// Pass 2: mean distances (Normalize2Ddata.m:35) and the 72 sums  mom[36 pair + 6 a + b] = sum m1[a] m_{2|3}[b]  over the monomials
// m = {x^2, xy, x, y^2, y, 1} of the normalised coordinates (f_kernel.h::accumulate_moments_f), one correspondence per lane.
// nrm: outer normalisation (s, ox, oy per view); nrm2: linearF's inner one.
__device__ __forceinline__ void rows_distances_moments_f(const RowSrc& s, const int N, const double (&c)[6], double* nrm, double* nrm2, double* mom) {
    const int p = rows_p();
    double acc[80];                                                          // 72 sums + 8 zeros: 80 -> 5 per lane in four halvings
#pragma unroll
    for (int k = 0; k < 80; ++k) acc[k] = 0.0;
    double d[3] = {0.0, 0.0, 0.0};
    Pt6 pnext = rows_load(s, (p < N) ? p : 0);
#pragma unroll 1
    for (int i = p; i < N; i += ROWL) {
        const Pt6 q = pnext;
        if (i + ROWL < N) pnext = rows_load(s, i + ROWL);
        const double x1 = q.v[0] - c[0], y1 = q.v[1] - c[1];
        const double x2 = q.v[2] - c[2], y2 = q.v[3] - c[3];
        const double x3 = q.v[4] - c[4], y3 = q.v[5] - c[5];
        const double m1[6] = {x1 * x1, x1 * y1, x1, y1 * y1, y1, 1.0};
        const double m2[6] = {x2 * x2, x2 * y2, x2, y2 * y2, y2, 1.0};
        const double m3[6] = {x3 * x3, x3 * y3, x3, y3 * y3, y3, 1.0};
        d[0] += sqrt_nonneg(m1[0] + m1[3]); d[1] += sqrt_nonneg(m2[0] + m2[3]); d[2] += sqrt_nonneg(m3[0] + m3[3]);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                acc[6 * a + b] += m1[a] * m2[b];
                acc[36 + 6 * a + b] += m1[a] * m3[b];
            }
    }
    const double r2c = sqrt(2.0);
    double sc[3];                                                            // total scale per view: outer x inner
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double norm0 = row_sum16(d[v]) / (double)N;                    // Normalize2Ddata.m:35
        const double so = r2c / norm0;                                       // :36
        const double di = so * norm0;                                        // mean distance of the normalised points from their (~1e-16) centroid
        const double si = r2c / di;                                          // linearF.m:45-46: 1 to rounding
        sc[v] = so * si;
        if (p == 3 * v) { nrm[3 * v] = so; nrm2[3 * v] = si; }
        if (p == 3 * v + 1) { nrm[3 * v + 1] = -r2c * c[2 * v] / norm0; nrm2[3 * v + 1] = 0.0; }          // :37
        if (p == 3 * v + 2) { nrm[3 * v + 2] = -r2c * c[2 * v + 1] / norm0; nrm2[3 * v + 2] = 0.0; }
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = halve_sum<8>(acc[i], acc[i + 40]);
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = halve_sum<4>(acc[i], acc[i + 20]);
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = halve_sum<2>(acc[i], acc[i + 10]);
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = halve_sum<1>(acc[i], acc[i + 5]);
    const int base = 5 * (p & 1) + 10 * ((p >> 1) & 1) + 20 * ((p >> 2) & 1) + 40 * ((p >> 3) & 1);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int e = base + i;
        if (e < 72) {
            const int pair = e / 36, a = (e % 36) / 6, b = e % 6;
            // degrees of {x^2, xy, x, y^2, y, 1}: {2, 2, 1, 2, 1, 0}
            const double s1 = sc[0], sv = pair ? sc[2] : sc[1];
            const double f1 = (a == 5) ? 1.0 : ((a == 2 || a == 4) ? s1 : s1 * s1);
            const double fv = (b == 5) ? 1.0 : ((b == 2 || b == 4) ? sv : sv * sv);
            mom[e] = acc[i] * (f1 * fv);
        }
    }
}

// linearF.m:48-62 for both view pairs from the moment sums, then LinearFPoseEstimation.m:55-56 and E = K' F K (recover_R_t) -> rt->Ein.
// Fm: 18 doubles of the row's LDS (F21, F31 row-major).  Returns false (per row) when a fast tier could not finish.
// X_FRAME (OptimFPoseEstimation, optimf_rows_kernel.h): stop after linearF itself and leave F / |F|_F (optimF.m:49-50) in Fm, in the frame of the
// normalised points; rt is not used.
template <bool X_FRAME = false>
__device__ __forceinline__ bool rows_linear_f_middle(RowLds* w, RowRt* rt, double* Fm, const double* nrm2, double* dbg) {
    const int p = opaque_lane_int(rows_p());
    bool ok = true;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {                                   // linearF(x1,x2), linearF(x1,x3)
        double g[9], none[1] = {0.0}, diag = 0.0, x0, x1, r2, risk;
        const bool have = p < 9;
        const int r = have ? p : 0, i = r / 3, jj = r % 3;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            g[c] = have ? w->mom[36 * pair + 6 * hht_index(i, c / 3) + hht_index(jj, c % 3)] : 0.0;
            diag = (c == r) ? g[c] : diag;
        }
        int its = 0;
        wave_sync();                                                         // (the previous pair's factor is dead)
        rows_min_eigvec<9>(g, none, diag, 0.0, w->ov, EIG_MAXIT, &its, &r2, false, 0.0, 0.0, &risk, x0, x1);
        ok = ok && eig_converged(r2) && risk == 0.0;
        if (dbg && p == 0) dbg[69 + pair] = (double)its;
        // F = reshape(V(:,9),3,3): F(rr,cc) = v[rr + 3 cc]   (linearF.m:55); stored row-major
        if (have) Fm[9 * pair + 3 * (p % 3) + p / 3] = x0;
        wave_sync();
    }
    bool nok = true;
    if (p < 2) {
        const int v2 = p + 1;
        Mat3 F;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] = Fm[9 * p + 3 * r + c];
        F = mat3_mul(mat3_mul(mat3_T(normal_matrix(nrm2, v2)), F), normal_matrix(nrm2, 0));          // linearF.m:58: inner de-normalisation
        double v3[3], fv[3];
        nok = null3<false>(F, v3);                                           // :61-62: rank 2
#pragma unroll
        for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] -= fv[r] * v3[c];
        if constexpr (X_FRAME) {
            double nn = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) nn += F.m[r][c] * F.m[r][c];
            const double sc = rsqrt(nn);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Fm[9 * p + 3 * r + c] = F.m[r][c] * sc;
        } else {
            F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, v2)), F), normal_matrix(w->nrm, 0));      // LinearFPoseEstimation.m:55-56: back to pixels
            const Mat3 E = mat3_mul(mat3_mul(mat3_T(load_K(w->calm, v2)), F), load_K(w->calm, 0));       // recover_R_t: E = K2' F K1
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) rt->Ein[9 * p + 3 * r + c] = E.m[r][c];
        }
    }
    const bool bad = row_any(!nok);
    wave_sync();
    return ok && !bad;
}

__global__ void __launch_bounds__(64, 2) k_linear_f_pose_rows(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    if (a.retry_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.retry_zero = 0;   // (the counter of the context's next call; this call's was zeroed during the previous one)
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const RowJob j = rows_begin(a, w, blk, N);
        int status;
        if (N < 8) {                                                         // linearF.m:35-37 (wave-uniform: N is the batch's)
            status = ST_TOO_FEW;
            rows_store_nan(a, j, N);
        } else {
            {
                double cen[6];
                rows_centroids(j.src, N, cen);                               // LinearFPoseEstimation.m:46-48
                rows_distances_moments_f(j.src, N, cen, w->nrm, w->pa, w->mom);      // (w->pa[0..8]: linearF's inner normalisation)
            }
            wave_sync();
            bool ok = rows_linear_f_middle(w, rt, w->t, w->pa, j.dbg);       // (w->t: F21, F31)
            rows_recover_prepare(w, rt);
            status = rows_pose_tail<true>(a, w, rt, j, N, ok);
        }
        if (p == 0 && j.valid) {
            if (a.iter) a.iter[j.b] = 0;                                     // LinearFPoseEstimation.m:77
            a.status[j.b] = status;
            if (status == ST_RETRY && a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1)] = (int)j.b;   // the list the exact kernel walks
        }
    }
}

// M rows (correspondences base .. base + M - 1) of the N x 9 system of one view pair appended to the row's packed R: entry h1 (x) h2 at position 3a + b
// (linearF.m:48-53), on the points normalised twice (LinearFPoseEstimation.m:46-48, linearF.m:45-46).
template <int M>
__device__ __forceinline__ void rows_f_system_chunk(const RowSrc& s, const int N, const int base, const double* nrm, const double* nrm2, const int p, const int ca,
                                                    const int cb, const int pair, double* Rp, double* xch) {
    double a0[M], a1[M];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const int i = base + r;                                              // row-uniform: every lane of the row reads the same correspondence
        const Pt6 q = premap(premap(rows_load(s, (i < N) ? i : 0), nrm), nrm2);
        const double h1 = (ca == 0) ? q.v[0] : (ca == 1) ? q.v[1] : 1.0;
        const double h2 = (cb == 0) ? (pair ? q.v[4] : q.v[2]) : (cb == 1) ? (pair ? q.v[5] : q.v[3]) : 1.0;
        a0[r] = (i < N && p < 9) ? h1 * h2 : 0.0;
        a1[r] = 0.0;
    }
    rows_qr_append<9, M>(a0, a1, Rp, xch);
}

// ---- the exact tiers of LinearFPoseEstimation, four triplets per wavefront (k_f_pose<true, 0> in the row layout; minimal samples / TFF_OPT_SOLVER = 1) ----
// Streaming Householder QR of the explicit N x 9 system of each view pair (rows_qr.h: position p < 9 owns column p, 16 rows per chunk), inverse
// iteration with L = R', certified 3 x 3 null vectors, all four votes with the exact re-score behind them, certified DLT ladder for t3 scale and
// Reconst.  linearF's own normalisation of the normalised points (linearF.m:45-46) is computed from them as the reference does.
// A row whose inverse iteration hits its cap is marked ST_RETRY for k_f_pose<true, 0> (one-sided Jacobi on R).
__device__ __forceinline__ bool rows_linear_f_middle_exact(RowLds* w, RowRt* rt, const RowSrc& s, const int N, double* Fm, double* nrm2, double* dbg) {
    const int p = opaque_lane_int(rows_p());
    double* Rp = w->ov;
    double* xch = w->mom;
    double* dinv = w->mom + 32;
    {                                                                        // linearF.m:45-46: Normalize2Ddata of the already normalised points
        double sm[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
        for (int i = p; i < N; i += ROWL) {
            const Pt6 q = premap(rows_load(s, i), w->nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) sm[k] += q.v[k];
        }
        double c[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = row_sum16(sm[k]) / (double)N;
        double d[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
        for (int i = p; i < N; i += ROWL) {
            const Pt6 q = premap(rows_load(s, i), w->nrm);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double dx = q.v[2 * v] - c[2 * v], dy = q.v[2 * v + 1] - c[2 * v + 1];
                d[v] += sqrt(dx * dx + dy * dy);
            }
        }
        const double r2c = sqrt(2.0);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double norm0 = row_sum16(d[v]) / (double)N;
            if (p == 3 * v) nrm2[3 * v] = r2c / norm0;
            if (p == 3 * v + 1) nrm2[3 * v + 1] = -r2c * c[2 * v] / norm0;
            if (p == 3 * v + 2) nrm2[3 * v + 2] = -r2c * c[2 * v + 1] / norm0;
        }
        wave_sync();
    }
    rows_stamp(dbg, 4);
    bool ok = true;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {                                   // linearF(x1,x2), linearF(x1,x3): rows h1 (x) h2, position 3a + b (linearF.m:48-53)
        rows_qr_clear<9>(Rp);
        const int col = (p < 9) ? p : 0, ca = col / 3, cb = col % 3;
        if (N <= 8) {                                                        // (wave-uniform) an eight-point sample: one chunk of eight rows -- the other eight would be zero rows that
            rows_f_system_chunk<8>(s, N, 0, w->nrm, nrm2, p, ca, cb, pair, Rp, xch);   // change nothing (fma(0, 0, x) = x) and cost half of the factorisation
        } else {
#pragma unroll 1
            for (int base = 0; base < N; base += 16) rows_f_system_chunk<16>(s, N, base, w->nrm, nrm2, p, ca, cb, pair, Rp, xch);
        }
        int its = 0;
        double x0, x1, r2;
        rows_invit_from_R<9>(Rp, dinv, EIG_MAXIT, &its, &r2, x0, x1);
        ok = ok && eig_converged(r2);
        if (dbg && p == 0) dbg[69 + pair] = (double)(20000 + its);
        if (p < 9) Fm[9 * pair + 3 * (p % 3) + p / 3] = x0;                  // F = reshape(V(:,9),3,3), row-major
        wave_sync();
        rows_stamp(dbg, 5 + pair);
    }
    if (p < 2) {
        const int v2 = p + 1;
        Mat3 F;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] = Fm[9 * p + 3 * r + c];
        F = mat3_mul(mat3_mul(mat3_T(normal_matrix(nrm2, v2)), F), normal_matrix(nrm2, 0));          // linearF.m:58
        double v3[3], fv[3];
        null3<true>(F, v3);                                                  // :61-62 (certified tier, one-sided Jacobi behind it)
#pragma unroll
        for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] -= fv[r] * v3[c];
        F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, v2)), F), normal_matrix(w->nrm, 0));      // LinearFPoseEstimation.m:55-56
        const Mat3 E = mat3_mul(mat3_mul(mat3_T(load_K(w->calm, v2)), F), load_K(w->calm, 0));       // E = K2' F K1
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) rt->Ein[9 * p + 3 * r + c] = E.m[r][c];
    }
    wave_sync();
    return ok;
}

__global__ void __launch_bounds__(64, 2) k_linear_f_pose_rows_exact(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    if (a.retry_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.retry_zero = 0;   // (the counter of the context's next call; this call's was zeroed during the previous one)
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const RowJob j = rows_begin(a, w, blk, N);
        int status;
        if (N < 8) {                                                         // linearF.m:35-37
            status = ST_TOO_FEW;
            rows_store_nan(a, j, N);
        } else {
            rows_stamp(j.dbg, 0);
            {
                double cen[6];
                rows_centroids(j.src, N, cen);                               // LinearFPoseEstimation.m:46-48
                rows_distances(j.src, N, cen, w->nrm);
            }
            rows_stamp(j.dbg, 1);
            // (the rank-2 F matrices go to w->t, linearF's inner normalisation to w->pa; the packed R lives in the overlay until E is formed --
            //  rt->Ein overlaps it, so E is written after the last solve)
            bool ok = rows_linear_f_middle_exact(w, rt, j.src, N, w->t, w->pa, j.dbg);
            rows_stamp(j.dbg, 2);
            rows_recover_prepare(w, rt);
            rows_stamp(j.dbg, 10);
            status = rows_pose_tail<true, true>(a, w, rt, j, N, ok);
        }
        if (p == 0 && j.valid) {
            if (a.iter) a.iter[j.b] = 0;                                     // LinearFPoseEstimation.m:77
            a.status[j.b] = status;
            if (status == ST_RETRY && a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1)] = (int)j.b;   // the list the exact kernel walks
        }
    }
}

}  // namespace tff
