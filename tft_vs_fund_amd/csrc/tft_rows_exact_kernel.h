// The EXACT tiers of LinearTFTPoseEstimation with four triplets per wavefront: what k_linear_tft_pose<true> (tft_kernel.h) does for one triplet per
// wavefront -- Householder QR of the explicit 4N x 27 system, inverse iteration with L = R', the 4N x 15 re-solve from R * Up, certified null
// vectors, cheirality votes and DLT points -- in the row layout of tft_rows_kernel.h.  It exists for the batches that go to the exact tiers as a
// whole: minimal samples (N < TFF_OPT_EXACT_BELOW; BASELINE.json configs[3]: a million seven-point hypotheses of one scene), where a
// wavefront per hypothesis leaves 57 of 64 lanes idle in every per-correspondence stage and 37 in the QR.
//   * QR and inverse iteration: rows_qr.h (the owner of a column publishes its chunk through LDS; four systems per wavefront);
//   * 3 x 3 null vectors: epipoles_from_tensor<16, true> (certified tier, one-sided Jacobi behind it);
//   * votes: all four fast certified scores in one pass, a candidate with an uncertified correspondence is re-scored from the converged
//     homogeneous points (rows_vote_exact); t3 scale and Reconst through the certified DLT ladder;
//   * the one thing a row cannot do here is the gap-independent fall-back of the two big solves (one-sided Jacobi on R when the inverse
//     iteration hits its cap: sigma_n / sigma_(n-1) > ~0.97, 5 of 100 000 seven-point samples): such a triplet is marked ST_RETRY and redone
//     by k_linear_tft_pose<true>.
// Reference: as tft_kernel.h / tft_rows_kernel.h.
#pragma once
#include "tft_rows_kernel.h"
#include "rows_qr.h"

namespace tff {

// distances only (Normalize2Ddata.m:35-39): nrm[3v..3v+2] = s, ox, oy for the row's triplet
__device__ __forceinline__ void rows_distances(const RowSrc& s, const int N, const double (&c)[6], double* nrm) {
    const int p = rows_p();
    double d[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
    for (int i = p; i < N; i += ROWL) {
        const Pt6 q = rows_load(s, i);
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
        if (p == 3 * v) nrm[3 * v] = r2c / norm0;
        if (p == 3 * v + 1) nrm[3 * v + 1] = -r2c * c[2 * v] / norm0;
        if (p == 3 * v + 2) nrm[3 * v + 2] = -r2c * c[2 * v + 1] / norm0;
    }
    wave_sync();
}

// R of the 4N x 27 system of linearTFT.m:51-62 on the normalised correspondences (tft_kernel.h::tft_system_qr), seven correspondences (28 rows)
// per chunk: entry (e, j + 3k + 9i) of a correspondence's four rows is h1[i] c3[k] c2[j].  Position p builds columns p and 16 + p.
__device__ __forceinline__ void rows_tft_system_qr(const RowSrc& s, const int N, const double* nrm, double* Rp, double* xch) {
    const int p = rows_p();
    rows_qr_clear<27>(Rp);
    const int colA = p, colB = (p < 11) ? 16 + p : 0;
#pragma unroll 1
    for (int base = 0; base < N; base += 7) {
        double a0[28], a1[28];
#pragma unroll
        for (int qi = 0; qi < 7; ++qi) {
            const int i = base + qi;                                         // row-uniform: every lane of the row reads the same correspondence
            const Pt6 q = premap(rows_load(s, (i < N) ? i : 0), nrm);
            const bool have = i < N;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = h ? colB : colA;
                const bool own = have && (h ? p < 11 : true);
                const int ci = col / 9, ck = (col % 9) / 3, cj = col % 3;
                const double hh = own ? ((ci == 0) ? q.v[0] : (ci == 1) ? q.v[1] : 1.0) : 0.0;
                const double c2x = (cj == 0) ? 1.0 : (cj == 1) ? 0.0 : -q.v[2], c2y = (cj == 0) ? 0.0 : (cj == 1) ? 1.0 : -q.v[3];
                const double c3x = (ck == 0) ? 1.0 : (ck == 1) ? 0.0 : -q.v[4], c3y = (ck == 0) ? 0.0 : (ck == 1) ? 1.0 : -q.v[5];
                const double hx = hh * c3x, hy = hh * c3y;
                double* dst = h ? a1 : a0;
                dst[4 * qi + 0] = hx * c2x; dst[4 * qi + 1] = hx * c2y; dst[4 * qi + 2] = hy * c2x; dst[4 * qi + 3] = hy * c2y;
            }
        }
        rows_qr_append<27, 28>(a0, a1, Rp, xch);
    }
}

// linearTFT.m:64-91 at the accuracy of the reference's svd() calls (tft_kernel.h::linear_tft_middle<true, 64>), one triplet per row.
// Rp: the row's packed R workspace (w->ov), xch / dinv: 28 + 27 doubles of the row's LDS (w->mom).  Returns false (per row) when one of the
// two inverse iterations hit its cap.
// *capped (per row): bit 0 / bit 1 set when the 27- / 15-column inverse iteration hit its cap (the hints of the retry list, tft_kernel.h).
__device__ __forceinline__ bool rows_linear_tft_middle_exact(RowLds* w, const RowSrc& s, const int N, double* dbg, int* capped) {
    const int p = opaque_lane_int(rows_p());
    double* Rp = w->ov;
    double* xch = w->mom;
    double* dinv = w->mom + 32;
    bool ok = true;
    int it1 = 0, it2 = 0;
    {                                                                        // :64-67
        rows_tft_system_qr(s, N, w->nrm, Rp, xch);
        rows_stamp(dbg, 4);
        double x0, x1, r2;
        rows_invit_from_R<27>(Rp, dinv, EIG_MAXIT, &it1, &r2, x0, x1);
        rows_stamp(dbg, 5);
        ok = ok && eig_converged(r2);
        *capped = eig_converged(r2) ? 0 : 1;
        wave_sync();
        w->t[p] = x0;
        if (p < 11) w->t[16 + p] = x1;
        wave_sync();
    }
    if (dbg) { dbg[p] = w->t[p]; if (p < 11) dbg[16 + p] = w->t[16 + p]; }
    // :71-79; the slice null vectors go to w->mom + 64 (18 doubles): the overlay holds R, which the re-solve below still needs
    const bool eok = epipoles_from_tensor<16, true>(w->t, w->mom + 64, w->epi, false);
    ok = !row_any(!eok) && ok;
    if (dbg && p < 6) dbg[27 + p] = w->epi[p];
    if (p == 0) frame_of(w->epi, w->Q);                                      // Q2 from e21
    if (p == 1) frame_of(w->epi + 3, w->Q + 9);                              // Q3 from e31
    wave_sync();
    rows_stamp(dbg, 6);
    {                                                                        // :84 from R: svd(A Up) == svd(R Up), A = Q R
        // column c = 5 i + m of B = R Up (27 x 15) on position c < 15: B[r][c] = sum_{k,j} R[r][j + 3k + 9i] Q2[j][jj] Q3[k][kk]
        double a0[27], a1[27];
        const bool have = p < 15;
        const int cc = have ? p : 0, i = cc / 5, m = cc % 5, jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
        double qq[9];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) qq[j + 3 * k] = w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk];
        const int i9 = 9 * i;
#pragma unroll
        for (int r = 0; r < 27; ++r) {                                       // (fully unrolled: r indexes registers; the loads are unconditional, the mask is on the value)
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const int col = e + i9;                                      // (lane-dependent through i: an LDS address, not a register index)
                const bool in = col >= r && have;
                const double rv = Rp[in ? r * 27 - (r * (r - 1)) / 2 + (col - r) : 0];
                acc = fma(in ? rv : 0.0, qq[e], acc);
            }
            a0[r] = acc;
            a1[r] = 0.0;
            pin_value(a0[r]);
            sched_fence();                                                   // (one row at a time: 243 loads hoisted to the top would spill everything around them)
        }
        wave_sync();                                                         // R is read; the 15 x 15 factor takes its place
        rows_qr_clear<15>(Rp);
        rows_qr_append<15, 27>(a0, a1, Rp, xch);
        rows_stamp(dbg, 7);
        double x0, x1, r2;
        rows_invit_from_R<15>(Rp, dinv, EIG_MAXIT, &it2, &r2, x0, x1);
        ok = ok && eig_converged(r2);
        *capped |= eig_converged(r2) ? 0 : 2;
        if (have) w->tp[p] = x0;
        wave_sync();
    }
    {                                                                        // t = Up * tp   (:85)
        double tv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = (16 * h + p < 27) ? 16 * h + p : 0;
            const int i = e / 9, k = (e % 9) / 3, j = e % 3;
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
                acc += w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk] * w->tp[5 * i + m];
            }
            tv[h] = acc;
        }
        wave_sync();
        w->t[p] = tv[0];
        if (p < 11) w->t[16 + p] = tv[1];
        wave_sync();
    }
    if (dbg) {
        dbg[33 + p] = w->t[p];
        if (p < 11) dbg[33 + 16 + p] = w->t[16 + p];
        if (p == 0) { dbg[69] = (double)(20000 + it1); dbg[70] = (double)(20000 + it2); }   // (20000 + iterations: this kernel's stamp)
    }
    return ok;
}

__global__ void __launch_bounds__(64, 2) k_linear_tft_pose_rows_exact(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    if (a.retry_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.retry_zero = 0;   // (the counter of the context's next call; this call's was zeroed during the previous one)
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const RowJob j = rows_begin(a, w, blk, N);
        double* dbg = j.dbg;
        int status, hint = 0;
        if (N < 7) {                                                         // experiments.m:99 (wave-uniform: N is the batch's)
            status = ST_TOO_FEW;
            rows_store_nan(a, j, N);
        } else {
            rows_stamp(dbg, 0);
            {
                double cen[6];
                rows_centroids(j.src, N, cen);                               // LinearTFTPoseEstimation.m:45-47
                rows_distances(j.src, N, cen, w->nrm);
                if (dbg && p < 9) dbg[71 + p] = w->nrm[p];
            }
            rows_stamp(dbg, 1);
            int capped = 0;
            bool ok = rows_linear_tft_middle_exact(w, j.src, N, dbg, &capped);   // :50
            hint = capped;
            rows_stamp(dbg, 2);
            rows_transform_tft_inverse(w->t, rt->T1, rt->mats, [w](int v) { return normal_matrix(w->nrm, v); });   // :53
            ok = rows_rt_prepare<true>(w, rt, dbg) && ok;                    // :56
            rows_stamp(dbg, 10);
            status = rows_pose_tail<false, true>(a, w, rt, j, N, ok);
        }
        if (p == 0 && j.valid) {
            if (a.iter) a.iter[j.b] = 0;                                     // :62
            a.status[j.b] = status;
            if (status == ST_RETRY && a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1)] = (int)j.b | (hint << RETRY_HINT_SHIFT);   // the list the exact kernel walks, with what is known about the failure
        }
    }
}

}  // namespace tff
