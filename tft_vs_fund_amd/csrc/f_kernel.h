// LinearFPoseEstimation as one fused kernel: one wavefront per triplet.
//
//   Normalize2Ddata x3 -> linearF(x1,x2), linearF(x1,x3) [each normalising its
//   inputs again, 8-point DLT, inner de-normalisation, rank-2 projection] ->
//   outer de-normalisation -> E = K' F K -> recover_R_t with the cheirality vote
//   -> t3 scale -> optional Reconst -> T = TFT_from_P(K1[I|0], K2 R_t_2, K3 R_t_3).
//
// Reference: F_methods/LinearFPoseEstimation.m:42-109, F_methods/linearF.m:32-62,
// TFT_methods/TFT_from_P.m:25-33, auxiliar_functions/Normalize2Ddata.m:33-39.
//
// The N x 9 design matrix of linearF.m:48-53 is never formed: its rows are
// h1 (x) h2 with h = (x, y, 1), so A'A = sum_n (h1 h1') (x) (h2 h2') has 6 x 6 = 36
// distinct entries per view pair (products of the monomials {x^2, xy, x, y^2, y, 1}).
// The wave accumulates the 72 sums of both pairs in one sweep, lane r builds row r
// of the 9x9 Gram matrix in registers, and wave_min_eigvec_reg<9> returns
// V(:,9) of svd(A) (linearF.m:54-55).
#pragma once
#include "pose_common.h"
#include "tft_kernel.h"

namespace tff {

// 72 moment sums: mom[pair*36 + 6*a + b] = sum_n m1[a] * m{2,3}[b], pair 0 = views (1,2), 1 = views (1,3).
// Three sweeps of 24 accumulators (2 a-values x 6 b-values x 2 pairs) per lane.
__device__ inline void accumulate_moments_f(PoseLds* w, const double* pts, int N) {
    const int lane = lane_id();
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        double acc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = 0.0;
        for (int i = lane; i < N; i += WAVE) {
            // outer normalisation (LinearFPoseEstimation.m:46-48), then linearF's own (linearF.m:45-46)
            const Pt6 p = premap(premap(load_pt(pts, i), w->nrm), w->nrm2);
            const double pa = (pass == 0) ? p.v[0] * p.v[0] : ((pass == 1) ? p.v[0] : p.v[1]);
            const double pb = (pass == 0) ? p.v[0] * p.v[1] : ((pass == 1) ? p.v[1] * p.v[1] : 1.0);
            const double m2[6] = {p.v[2] * p.v[2], p.v[2] * p.v[3], p.v[2], p.v[3] * p.v[3], p.v[3], 1.0};
            const double m3[6] = {p.v[4] * p.v[4], p.v[4] * p.v[5], p.v[4], p.v[5] * p.v[5], p.v[5], 1.0};
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                acc[2 * b + 0] += pa * m2[b];
                acc[2 * b + 1] += pa * m3[b];
                acc[12 + 2 * b + 0] += pb * m2[b];
                acc[12 + 2 * b + 1] += pb * m3[b];
            }
        }
        const double tot = wave_reduce_scatter<32>(acc);
        const int idx = reduce32_index(lane);
        if ((lane & 1) == 0 && idx < 24) {
            const int al = idx / 12, b = (idx % 12) / 2, pair = idx & 1;
            w->mom[36 * pair + 6 * (2 * pass + al) + b] = tot;
        }
    }
    wave_sync();
}

// TFT_from_P.m:25-33 with P1 = K1 [I|0]:  T(j,k,i) = (-1)^(i+1) det[P1 without row i; P2(j,:); P3(k,:)]
__device__ __forceinline__ double det4(const double (&m)[4][4]) {
    const double s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1], s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    const double s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3], s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    const double s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3], s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    const double c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const double c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const double c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
}
__device__ inline void tft_from_cameras(PoseLds* w, double* tout) {     // cameras w->Pfin[0..2] (row-major 3x4)
    const int lane = lane_id();
    double val = 0.0;
    if (lane < 27) {
        const int i = lane / 9, k = (lane % 9) / 3, j = lane % 3;
        const int r0 = (i == 0) ? 1 : 0, r1 = (i == 2) ? 1 : 2;           // rows of P1 kept
        double m[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            m[0][c] = w->Pfin[0][4 * r0 + c];
            m[1][c] = w->Pfin[0][4 * r1 + c];
            m[2][c] = w->Pfin[1][4 * j + c];
            m[3][c] = w->Pfin[2][4 * k + c];
        }
        val = ((i == 1) ? -1.0 : 1.0) * det4(m);
    }
    const double nn = wave_sum(val * val);
    if (lane < 27) tout[lane] = val * rsqrt(nn);                           // :33
}

template <bool JAC>
__global__ void __launch_bounds__(64, 2) k_linear_f_pose(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLds* jw = JAC ? reinterpret_cast<JacobiLds*>(smem + base) : nullptr;
    double* lds_pts = smem + base + (JAC ? ((JACOBI_LDS_DOUBLES + 1) & ~1) : 0);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;      // wave-uniform
        const int N = a.N;
        double* dbg = a.dbg ? a.dbg + b * DBG_STRIDE : nullptr;
        const double* src = a.corresp + b * 6 * (long)N;
        const double* pts = src;
        wave_sync();
        if (a.sample_idx) {
            gather_points(a.corresp, a.sample_idx + b * (long)N, lds_pts, N);
            pts = lds_pts;
        } else if (a.flags & FLAG_STAGE_LDS) {
            stage_points(src, lds_pts, N);
            pts = lds_pts;
        }
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        int status = ST_OK;
        if (N < 8) {                                                         // linearF.m:35-37
            status = ST_TOO_FEW;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        } else {
            normalise3(pts, N, w->nrm);                                      // LinearFPoseEstimation.m:46-48
            normalise3(pts, N, w->nrm2, w->nrm);                             // linearF.m:45-46 (on the normalised points)
            accumulate_moments_f(w, pts, N);
            bool ok = true;
#pragma unroll 1
            for (int pair = 0; pair < 2; ++pair) {                           // linearF(x1,x2), linearF(x1,x3)   (:51-52)
                double g[9], diag = 0.0, x;
                const int r = (lane < 9) ? lane : 0, i = r / 3, j = r % 3;
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    g[c] = w->mom[36 * pair + 6 * hht_index(i, c / 3) + hht_index(j, c % 3)];
                    diag = (c == r) ? g[c] : diag;
                }
                int its = 0;
                if (JAC) {
                    if (lane < 9) for (int c = 0; c < 9; ++c) jw->A[lane * 9 + c] = g[c];
                    wave_sync();
                    x = wave_jacobi_min_eigvec(jw->A, jw->V, 9, 9, &its);
                } else {
                    double r2;
                    x = wave_min_eigvec_reg<9>(g, diag, w->Lp, EIG_MAXIT, &its, &r2);
                    ok = ok && eig_converged(r2);
                }
                if (dbg && lane == 0) dbg[69 + pair] = (double)its;
                // F = reshape(V(:,9),3,3): F(rr,cc) = v[rr + 3 cc]   (linearF.m:55); stored row-major
                if (lane < 9) w->Fm[9 * pair + 3 * (lane % 3) + lane / 3] = x;
                wave_sync();
            }
            if (!ok) {
                status = ST_RETRY;
            } else {
                double* Ein = w->Minv;
                if (lane < 2) {
                    const int v2 = lane + 1;                                 // second view of this pair
                    Mat3 F;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) F.m[r][c] = w->Fm[9 * lane + 3 * r + c];
                    F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm2, v2)), F), normal_matrix(w->nrm2, 0));   // linearF.m:58
                    double v3[3];                                            // rank 2: F - (F v3) v3'   (linearF.m:61-62)
                    null3(F, v3);
                    double fv[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) F.m[r][c] -= fv[r] * v3[c];
                    F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, v2)), F), normal_matrix(w->nrm, 0));     // LinearFPoseEstimation.m:55-56
                    const Mat3 E = mat3_mul(mat3_mul(mat3_T(load_K(w->calm, v2)), F), load_K(w->calm, 0));      // :86
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) Ein[9 * lane + 3 * r + c] = E.m[r][c];
                }
                wave_sync();
                status = recover_poses(w, Ein, pts, N, dbg);                 // :59-60
                scale_t3(w, pts, N, dbg);                                    // :64-70
                write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
                if (lane == 0) compose_camera_from_pose(load_K(w->calm, 2), w->Rt[1], w->Pfin[2]);   // K3 [R3 | lam t3]
                wave_sync();
                if (a.reconst) tri_pass(w, pts, N, TRI_RECONST, 1, w->Pfin[1], w->Pfin[2], a.reconst + b * 3 * (long)N);   // :75-76
                tft_from_cameras(w, w->T1);                                  // :78
                wave_sync();
                if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
                double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
                const bool bad = !(fabs(chk) <= 1.79e308);
                if (wave_any(bad) && status == ST_OK) status = ST_NONFINITE;
            }
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = 0;                                       // :77
            a.status[b] = status;
        }
    }
}

}  // namespace tff
