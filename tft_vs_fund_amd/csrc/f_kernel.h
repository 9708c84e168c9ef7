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
#include "gh_kernel.h"

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

// ---- optimF: Gauss-Helmert refinement of one fundamental matrix (F_methods/optimF.m:34-109) ----
// One epipolar equation per correspondence, so the blocks of Gauss_Helmert.m are scalars:
//   W_i = B_i B_i' (1x1), pinv(W + 1e-12 I) + 1e-12 I diagonal with pinv's global tolerance N eps(max W),
//   A'WA = sum_i W_i a_i a_i' (45 sums) and A'Ww (9 sums): two 32-accumulator sweeps, 11 x 11 KKT system.
struct OptimFLds {
    double p[10];          // F(:) column-major (9)
    double dt[12];         // KKT solution (9 + 2)
    double H[56];          // 45 + 9 accumulated sums
    double M[11 * 12];     // augmented KKT matrix
};
constexpr int OPTIMF_FIXED_DOUBLES = (int)(sizeof(OptimFLds) / sizeof(double));
__host__ __device__ inline int optimf_lds_doubles(int N) { return OPTIMF_FIXED_DOUBLES + 4 * N + 2; }   // + xi (4N)

// per correspondence: f = x2' F x1, a (1x9), B (1x4)   (optimF.m:101-107); o = [x1 y1 x2 y2], Fl = F(:) column-major
__device__ __forceinline__ void epi_block(const double (&Fl)[9], const double (&o)[4], double& f, double (&a)[9], double (&B)[4]) {
    const double x1 = o[0], y1 = o[1], x2 = o[2], y2 = o[3];
    a[0] = x1 * x2; a[1] = x1 * y2; a[2] = x1; a[3] = y1 * x2; a[4] = y1 * y2; a[5] = y1; a[6] = x2; a[7] = y2; a[8] = 1.0;
    f = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) f += Fl[k] * a[k];
    B[0] = Fl[2] + Fl[0] * x2 + Fl[1] * y2;
    B[1] = Fl[5] + Fl[3] * x2 + Fl[4] * y2;
    B[2] = Fl[6] + Fl[0] * x1 + Fl[3] * y1;
    B[3] = Fl[7] + Fl[1] * x1 + Fl[4] * y1;
}

// x (normalised observations of views 1 and v2) for correspondence i
__device__ __forceinline__ void obs4(const double* pts, int i, const double* nrm, int v2, double (&x)[4]) {
    const Pt6 p = premap(load_pt(pts, i), nrm);
    x[0] = p.v[0]; x[1] = p.v[1]; x[2] = p.v[2 * v2]; x[3] = p.v[2 * v2 + 1];
}

// Gauss_Helmert.m:38-83 specialised to optimF's callback.  g.p holds F(:), xi the initial estimates, nrm the map from pts to the
// normalised observations.  Returns iterations.
// Three passes over the correspondences per iteration (round 4; five before): the two accumulation sweeps and the pass for v.  The latter
// also writes xi = x + v in place -- nothing reads xi again when one of the stopping tests fires (only ti = p is kept then, :82) -- and takes
// the maximum of W = B B' at the NEXT iterate (F + dt, x + v) that pinv's tolerance needs (:52,57), so neither the update pass (:80) nor the
// maximum pass of the next iteration exist, nor the buffer for v.
// ONE_SWEEP (kernels compiled for 256 registers): all 54 sums in one pass, 64 accumulators, instead of two passes with 32 each.
template <bool ONE_SWEEP = false>
__device__ inline int gauss_helmert_f_wave(const double* nrm, OptimFLds* g, double* xi, const double* pts, int N, int v2, int* st) {
    const int lane = lane_id();
    constexpr int u = 9, n = 11, ld = 12;
    double objFunc = 0.0, smax = 0.0;
    {
        double Fl[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Fl[k] = wave_uniform(g->p[k]);
        for (int i = lane; i < N; i += WAVE) {
            double x[4];
            obs4(pts, i, nrm, v2, x);
            double o[4] = {xi[4 * i], xi[4 * i + 1], xi[4 * i + 2], xi[4 * i + 3]}, f, a[9], B[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double d = o[k] - x[k]; objFunc += d * d; }
            epi_block(Fl, o, f, a, B);
            const double wv = B[0] * B[0] + B[1] * B[1] + B[2] * B[2] + B[3] * B[3] + 1e-12;
            smax = (wv > smax) ? wv : smax;
            if (!(wv <= 1.79e308)) smax = wv;                                // NaN/Inf propagates
        }
    }
    objFunc = wave_sum(objFunc);
    int it = 0;
#pragma unroll 1
    for (it = 1; it <= 400; ++it) {
        double Fl[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Fl[k] = wave_uniform(g->p[k]);
        // W = B B' (+1e-12), its maximum -> pinv tolerance   (Gauss_Helmert.m:52,57)
        smax = wave_max(smax);
        if (!(smax <= 1.79e308)) { *st = ST_NONFINITE; break; }              // :53-55
        const double tolW = (double)N * eps_of(smax);
        // sums: H[0..44] = lower triangle of sum W a a', H[45..53] = sum W w a
        if constexpr (ONE_SWEEP) {
            double acc[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[k] = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                double o[4] = {xi[4 * i], xi[4 * i + 1], xi[4 * i + 2], xi[4 * i + 3]}, f, a[9], B[4], x[4];
                epi_block(Fl, o, f, a, B);
                const double wv = B[0] * B[0] + B[1] * B[1] + B[2] * B[2] + B[3] * B[3] + 1e-12;
                const double Wp = ((wv > tolW) ? 1.0 / wv : 0.0) + 1e-12;    // :57
                obs4(pts, i, nrm, v2, x);
                const double wr = -f - (B[0] * (x[0] - o[0]) + B[1] * (x[1] - o[1]) + B[2] * (x[2] - o[2]) + B[3] * (x[3] - o[3]));   // :58
                double wa[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) wa[k] = Wp * a[k];
                int e = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) { acc[e] += wa[r] * a[c]; ++e; }
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[45 + k] += wa[k] * wr;
            }
            const double tot = wave_reduce_scatter64(acc);
            const int idx = reduce64_index(lane);
            if (idx < 54) g->H[idx] = tot;
        } else {
#pragma unroll 1
        for (int sweep = 0; sweep < 2; ++sweep) {
            double acc[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                double o[4] = {xi[4 * i], xi[4 * i + 1], xi[4 * i + 2], xi[4 * i + 3]}, f, a[9], B[4], x[4];
                epi_block(Fl, o, f, a, B);
                const double wv = B[0] * B[0] + B[1] * B[1] + B[2] * B[2] + B[3] * B[3] + 1e-12;
                const double Wp = ((wv > tolW) ? 1.0 / wv : 0.0) + 1e-12;    // :57
                obs4(pts, i, nrm, v2, x);
                const double wr = -f - (B[0] * (x[0] - o[0]) + B[1] * (x[1] - o[1]) + B[2] * (x[2] - o[2]) + B[3] * (x[3] - o[3]));   // :58
                double wa[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) wa[k] = Wp * a[k];
                if (sweep == 0) {
                    int e = 0;
#pragma unroll
                    for (int r = 0; r < 9; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) { if (e < 32) acc[e] += wa[r] * a[c]; ++e; }
                } else {
                    int e = 0;
#pragma unroll
                    for (int r = 0; r < 9; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) { if (e >= 32) acc[e - 32] += wa[r] * a[c]; ++e; }
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc[13 + k] += wa[k] * wr;
                }
            }
            const double tot = wave_reduce_scatter<32>(acc);
            const int idx = reduce32_index(lane);
            if ((lane & 1) == 0 && (sweep == 0 || idx < 22)) g->H[32 * sweep + idx] = tot;
        }
        }
        wave_sync();
        // KKT matrix: [N + 1e-12 I, C'; C, 1e-12 I], b = [A'Ww; -g]   (:59-62), constraints optimF.m:90-95
        for (int e = lane; e < n * ld; e += WAVE) g->M[e] = 0.0;
        wave_sync();
        for (int e = lane; e < 81 + 9; e += WAVE) {
            if (e < 81) {
                const int r = e / 9, c = e % 9, hi = (r > c) ? r : c, lo = (r > c) ? c : r;
                g->M[r * ld + c] = g->H[hi * (hi + 1) / 2 + lo] + ((r == c) ? 1e-12 : 0.0);
            } else {
                g->M[(e - 81) * ld + n] = g->H[45 + e - 81];
            }
        }
        if (lane == 0) {
            const double* F = g->p;                                          // F(k+1) = F[k]
            const double C0[9] = {F[4] * F[8] - F[5] * F[7], F[5] * F[6] - F[3] * F[8], F[3] * F[7] - F[4] * F[6],
                                  F[2] * F[7] - F[1] * F[8], F[0] * F[8] - F[2] * F[6], F[1] * F[6] - F[0] * F[7],
                                  F[1] * F[5] - F[2] * F[4], F[2] * F[3] - F[0] * F[5], F[0] * F[4] - F[1] * F[3]};
            double det = F[0] * C0[0] + F[3] * C0[3] + F[6] * C0[6];          // det(F) by the first row (F(1,1),F(1,2),F(1,3) = F[0],F[3],F[6])
            double nn = 0.0;
            for (int k = 0; k < 9; ++k) {
                g->M[9 * ld + k] = C0[k]; g->M[k * ld + 9] = C0[k];
                g->M[10 * ld + k] = 2.0 * F[k]; g->M[k * ld + 10] = 2.0 * F[k];
                nn += F[k] * F[k];
            }
            g->M[9 * ld + 9] = 1e-12; g->M[10 * ld + 10] = 1e-12;
            g->M[9 * ld + n] = -det; g->M[10 * ld + n] = -(nn - 1.0);
        }
        wave_sync();
        double chk = 0.0;
        for (int e = lane; e < n * ld; e += WAVE) chk += g->M[e];
        if (!(fabs(wave_sum(chk)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        if (!wave_solve_gj<n>(g->M, g->dt)) { *st = ST_RANK; break; }        // :67
        wave_sync();
        double dt[9], Fn[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { dt[k] = wave_uniform(g->dt[k]); Fn[k] = Fl[k] + dt[k]; }   // Fn: ti + dt (:80), should the step be accepted
        // v = -B' W (A dt - w)   (:69); xi = x + v (:80); max W at the next iterate
        double obj = 0.0, diff = 0.0, ndt2 = 0.0, snext = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) ndt2 += dt[k] * dt[k];
        for (int i = lane; i < N; i += WAVE) {
            double o[4] = {xi[4 * i], xi[4 * i + 1], xi[4 * i + 2], xi[4 * i + 3]}, f, a[9], B[4], x[4];
            epi_block(Fl, o, f, a, B);
            const double wv = B[0] * B[0] + B[1] * B[1] + B[2] * B[2] + B[3] * B[3] + 1e-12;
            const double Wp = ((wv > tolW) ? 1.0 / wv : 0.0) + 1e-12;
            obs4(pts, i, nrm, v2, x);
            const double wr = -f - (B[0] * (x[0] - o[0]) + B[1] * (x[1] - o[1]) + B[2] * (x[2] - o[2]) + B[3] * (x[3] - o[3]));
            double adt = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) adt += a[k] * dt[k];
            const double r = Wp * (adt - wr);
            double on[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = -B[k] * r;
                obj += v * v;
                const double d = o[k] - x[k] - v;
                diff += d * d;
                on[k] = x[k] + v;
                xi[4 * i + k] = on[k];
            }
            const double Bn0 = Fn[2] + Fn[0] * on[2] + Fn[1] * on[3], Bn1 = Fn[5] + Fn[3] * on[2] + Fn[4] * on[3];   // epi_block's B at (Fn, on)
            const double Bn2 = Fn[6] + Fn[0] * on[0] + Fn[3] * on[1], Bn3 = Fn[7] + Fn[1] * on[0] + Fn[4] * on[1];
            const double wn = Bn0 * Bn0 + Bn1 * Bn1 + Bn2 * Bn2 + Bn3 * Bn3 + 1e-12;
            snext = (wn > snext) ? wn : snext;
            if (!(wn <= 1.79e308)) snext = wn;
        }
        obj = wave_sum(obj);
        diff = wave_sum(diff);
        if (sqrt(ndt2) < 1e-6 && sqrt(diff) < 1e-6) break;                   // :71-73
        if (obj > objFunc) break;                                            // :75-76
        objFunc = obj;
        smax = snext;
        if (lane < 9) g->p[lane] += g->dt[lane];                             // ti += dt   (:80)
        wave_sync();
    }
    return (it > 400) ? 400 : it;
}

// R of the N x 9 system of linearF.m:48-53 for N <= M correspondences, column layout (wave_qr_cols_append), left in Rl (9 x 9 LDS)
template <int M>
__device__ inline void linear_f_system_cols(PoseLds* w, const double* pts, int N, int pair, double* Rl) {
    const int lane = lane_id();
    for (int e = lane; e < 81; e += WAVE) Rl[e] = 0.0;
    wave_sync();
    const int col = (lane < 9) ? lane : 0, ca = col / 3, cb = col % 3;
    double a[M];
#pragma unroll
    for (int q = 0; q < M; ++q) {
        const Pt6 p = premap(premap(load_pt(pts, (q < N) ? q : 0), w->nrm), w->nrm2);   // wave-uniform address
        const double h1 = (ca == 0) ? p.v[0] : (ca == 1) ? p.v[1] : 1.0;
        const double x2 = pair ? p.v[4] : p.v[2], y2 = pair ? p.v[5] : p.v[3];
        const double h2 = (cb == 0) ? x2 : (cb == 1) ? y2 : 1.0;
        a[q] = (q < N && lane < 9) ? h1 * h2 : 0.0;
    }
    wave_qr_cols_append<9, M>(a, Rl);
}

// linearF(x1,x2) and linearF(x1,x3) (F_methods/linearF.m:45-62) on the points premapped by w->nrm, with linearF's own
// normalisation in w->nrm2: 8-point DLT through the 72 moment sums, inner de-normalisation, rank-2 projection.
// Result: w->Fm[9 pair + 3 r + c] (row-major), optionally scaled to unit Frobenius norm.  false: eigen-solver wants the Jacobi pass.
template <bool JAC>
__device__ inline bool linear_f_wave(PoseLds* w, JacobiLdsF* jw, const double* pts, int N, double* dbg, bool unit_norm) {
    const int lane = lane_id();
    if (!JAC) accumulate_moments_f(w, pts, N);
    bool ok = true;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {                                   // linearF(x1,x2), linearF(x1,x3)
        double x;
        int its = 0;
        if (JAC) {
            // exact tier: streaming QR of the N x 9 system itself (linearF.m:48-53; row = h1 (x) h2, position 3a + b), 55 rows per chunk
            // (minimal samples, N <= 16: one chunk in the column layout -- lane c < 9 owns column c -- at a third of the instructions)
            double g[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) g[c] = 0.0;
            if (N <= 8) {
                linear_f_system_cols<8>(w, pts, N, pair, jw->A);
                wave_qr_rows_from_lds<9>(jw->A, g);
            } else if (N <= 16) {
                linear_f_system_cols<16>(w, pts, N, pair, jw->A);
                wave_qr_rows_from_lds<9>(jw->A, g);
            } else
#pragma unroll 1
            for (int base = 0; base < N; base += 55) {
                const int i = base + lane - 9;
                if (lane >= 9) {
                    const bool have = i < N;
                    const Pt6 p = premap(premap(load_pt(pts, have ? i : 0), w->nrm), w->nrm2);
                    const double h1[3] = {p.v[0], p.v[1], 1.0};
                    const double h2[3] = {pair ? p.v[4] : p.v[2], pair ? p.v[5] : p.v[3], 1.0};
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int bb = 0; bb < 3; ++bb) g[3 * a + bb] = have ? h1[a] * h2[bb] : 0.0;
                }
                wave_qr_append<9>(g);
            }
            x = wave_qr_min_rsv<9>(g, jw->A, jw->V, w->Lp, EIG_MAXIT, &its);
            its += 10000;
        } else {
            double g[9], none[1] = {0.0}, diag = 0.0;                        // every row of 16 lanes holds the matrix (row_eig.h)
            const int rp = lane & 15;
            const bool have = rp < 9;
            const int r = have ? rp : 0, i = r / 3, j = r % 3;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                g[c] = have ? w->mom[36 * pair + 6 * hht_index(i, c / 3) + hht_index(j, c % 3)] : 0.0;
                diag = (c == r) ? g[c] : diag;
            }
            double r2, risk;
            x = row_min_eigvec<9>(g, none, diag, 0.0, w->Lp, EIG_MAXIT, &its, &r2, false, 0.0, 0.0, &risk);
            ok = ok && eig_converged(r2) && risk == 0.0;
        }
        if (dbg && lane == 0) dbg[69 + pair] = (double)its;
        // F = reshape(V(:,9),3,3): F(rr,cc) = v[rr + 3 cc]   (linearF.m:55); stored row-major
        if (lane < 9) w->Fm[9 * pair + 3 * (lane % 3) + lane / 3] = x;
        wave_sync();
    }
    if (!ok) return false;                                                   // wave-uniform so far
    if (lane < 2) {                                                          // linearF.m:58-62: inner de-normalisation, rank 2
        const int v2 = lane + 1;
        Mat3 F;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] = w->Fm[9 * lane + 3 * r + c];
        F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm2, v2)), F), normal_matrix(w->nrm2, 0));
        double v3[3], fv[3];
        ok = null3<JAC>(F, v3);
#pragma unroll
        for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { F.m[r][c] -= fv[r] * v3[c]; nn += F.m[r][c] * F.m[r][c]; }
        const double sc = unit_norm ? rsqrt(nn) : 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) w->Fm[9 * lane + 3 * r + c] = F.m[r][c] * sc;
    }
    wave_sync();
    return !wave_any(!ok);
}

// optimF.m:52-69 for both pairs: initial x_est by two-view triangulation with P1 = [I|0], P2 = [crossM(epi21) F, epi21],
// then Gauss-Helmert on F(:).  w->Fm holds the unit-norm linear F (x-frame) on entry, the refined one on exit.  Returns it1 + it2.
// *ok (EXACT = false): cleared when a fast tier (null vector, DLT point) could not finish.
template <bool EXACT = true, bool ONE_SWEEP = false>
__device__ inline int optim_f_refine(PoseLds* w, OptimFLds* og, double* oxi, const double* pts, int N, int* gst, bool* ok = nullptr) {
    const int lane = lane_id();
    int iters = 0;
    bool fine = true;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {
        if (lane == 0) {
            Mat3 F, Ft;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F.m[r][c] = w->Fm[9 * pair + 3 * r + c];
            Ft = mat3_T(F);
            double e[3];
            fine = null3<EXACT>(Ft, e) && fine;                              // epi21 = U(:,3): left null vector   (optimF.m:53)
            // P1 = [I|0] -> Pfin[0];  P2 = [crossM(epi21)*F, epi21] -> P[0]   (:54-55)
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) w->Pfin[0][4 * r + c] = (r == c) ? 1.0 : 0.0;
            for (int c = 0; c < 3; ++c) {
                w->P[0][c] = -e[2] * F.m[1][c] + e[1] * F.m[2][c];
                w->P[0][4 + c] = e[2] * F.m[0][c] - e[0] * F.m[2][c];
                w->P[0][8 + c] = -e[1] * F.m[0][c] + e[0] * F.m[1][c];
            }
            w->P[0][3] = e[0]; w->P[0][7] = e[1]; w->P[0][11] = e[2];
        }
        if (lane < 9) og->p[lane] = w->Fm[9 * pair + 3 * (lane % 3) + lane / 3];   // p = F(:) column-major   (:61)
        wave_sync();
        fine = tri_pass<EXACT>(w, pts, N, TRI_REPROJECT2, pair + 1, w->P[0], w->P[0], oxi, w->nrm) && fine;   // x_est   (:56-60)
        if (!EXACT && wave_any(!fine)) { if (ok) *ok = false; return iters; }                                  // the exact kernel redoes the triplet
        wave_sync();
        iters += gauss_helmert_f_wave<ONE_SWEEP>(w->nrm, og, oxi, pts, N, pair + 1, gst);   // :66
        wave_sync();
        if (lane < 9) w->Fm[9 * pair + 3 * (lane % 3) + lane / 3] = og->p[lane];   // F = reshape(p_opt,3,3)   (:69)
        wave_sync();
    }
    return iters;
}

// METHOD 0: LinearFPoseEstimation; METHOD 1: OptimFPoseEstimation (F_methods/OptimFPoseEstimation.m:44-73)
template <bool JAC, int METHOD>
__global__ void __launch_bounds__(64, (METHOD == 1 && !JAC) ? 3 : 2) k_f_pose(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLdsF* jw = JAC ? reinterpret_cast<JacobiLdsF*>(smem + base) : nullptr;
    double* extra = smem + base + (JAC ? ((JACOBI_F_LDS_DOUBLES + 1) & ~1) : 0);
    OptimFLds* og = (METHOD == 1) ? reinterpret_cast<OptimFLds*>(extra) : nullptr;
    double* oxi = a.spill ? a.spill + blockIdx.x * a.spill_stride : extra + ((OPTIMF_FIXED_DOUBLES + 1) & ~1);   // METHOD 1: xi (4N); large N: global
    double* lds_pts = (METHOD == 1) ? oxi + 4 * a.N + 2 : extra;             // staged correspondences (METHOD 0, or sampled)
    const int lane = lane_id();
    const long nwork = (a.retry_list && (a.flags & FLAG_ONLY_RETRY)) ? (long)*a.retry_count : a.B;            // (fix-up of a row kernel: the compact list of k_collect_retry)
    for (long wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const long b = (a.retry_list && (a.flags & FLAG_ONLY_RETRY)) ? (long)(a.retry_list[wi] & RETRY_INDEX_MASK) : wi;
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;      // wave-uniform
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        double* dbg = a.dbg ? a.dbg + b * DBG_STRIDE : nullptr;
        const double* src = a.corresp + b * 6 * (long)N;
        const double* pts = src;
        wave_sync();
        bool bad_index = false;
        if (a.sample_idx) {
            bad_index = gather_points(a.corresp, a.sample_idx + b * (long)N, lds_pts, N, a.sample_ns);
            pts = lds_pts;
        } else if (a.flags & FLAG_STAGE_LDS) {
            stage_points(src, lds_pts, N);
            pts = lds_pts;
        }
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        int status = ST_OK, iters = 0;
        if (N < 8 || bad_index) {                                            // linearF.m:35-37, optimF.m:36-38 (or a sample index outside the scene)
            status = ST_TOO_FEW;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        } else {
            normalise3(pts, N, w->nrm);                                      // LinearFPoseEstimation.m:46-48 / optimF.m:46-47
            normalise3(pts, N, w->nrm2, w->nrm);                             // linearF.m:45-46 (on the normalised points)
            const bool ok = linear_f_wave<JAC>(w, jw, pts, N, dbg, METHOD == 1);   // optimF.m:50: F = F / |F|_F
            if (!ok) {
                status = ST_RETRY;
            } else {
                wave_sync();
                int gst = ST_OK;
                bool fine = true;
                if (METHOD == 1) iters = optim_f_refine<JAC>(w, og, oxi, pts, N, &gst, &fine);   // [F21,it1] = optimF(...), [F31,it2] = optimF(...)   (:48-49)
                double* Ein = w->Minv;
                if (lane < 2) {
                    const int v2 = lane + 1;
                    Mat3 F;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) F.m[r][c] = w->Fm[9 * lane + 3 * r + c];
                    // LinearFPoseEstimation.m:55-56 / optimF.m:72: back to pixel coordinates
                    F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, v2)), F), normal_matrix(w->nrm, 0));
                    if (METHOD == 1) {                                       // optimF.m:75-76: rank 2 again
                        double v3[3], fv[3];
                        fine = null3<JAC>(F, v3) && fine;
#pragma unroll
                        for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c) F.m[r][c] -= fv[r] * v3[c];
                    }
                    const Mat3 E = mat3_mul(mat3_mul(mat3_T(load_K(w->calm, v2)), F), load_K(w->calm, 0));      // recover_R_t: E = K2' F K1
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) Ein[9 * lane + 3 * r + c] = E.m[r][c];
                }
                wave_sync();
                fine = !wave_any(!fine);
                if (fine) {
                    status = recover_poses<JAC, 2>(w, Ein, pts, N, dbg, &fine);       // (fused votes, second camera from LDS: pose_common.h::recover_vote)
                    if (gst != ST_OK) status = gst;
                }
                if (fine) fine = scale_t3<JAC>(w, pts, N, dbg);
                if (fine) {
                    if (lane == 0) compose_camera_from_pose(load_K(w->calm, 2), w->Rt[1], w->Pfin[2]);   // K3 [R3 | lam t3]
                    wave_sync();
                    if (a.reconst) fine = tri_pass<JAC>(w, pts, N, TRI_RECONST, 1, w->Pfin[1], w->Pfin[2], a.reconst + b * 3 * (long)N);
                }
                if (!fine) {
                    status = ST_RETRY;                                       // a fast tier gave up: redone by k_f_pose<true, .>
                } else {
                    write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
                    tft_from_cameras(w, w->T1);
                    wave_sync();
                    if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
                    double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
                    const bool bad = !(fabs(chk) <= 1.79e308);
                    if (wave_any(bad)) { if (status == ST_OK) status = ST_NONFINITE; wave_nan_outputs(a.Rt2, a.Rt3, a.T, a.reconst, b, N); }
                }
            }
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = iters;
            a.status[b] = status;
        }
    }
}

// ---- building blocks: linearF / optimF for the view pairs (1,2) and (1,3) of each item ----------------------
struct LinearFOnlyArgs {
    const double* corresp;   // B x (6 x N) pixel (or any) coordinates, rows x1;y1;x2;y2;x3;y3
    long B; int N; int flags;
    double* F21; double* F31;   // B x 9 each, 3x3 column-major: x2' F21 x1 = 0, x3' F31 x1 = 0
    int* iter;               // REFINE: it1 + it2 (or null)
    int* status;
};
// REFINE 0: linearF(p1,p2) (F_methods/linearF.m:32-62); REFINE 1: optimF(p1,p2) (F_methods/optimF.m:34-78)
template <bool JAC, int REFINE>
__global__ void __launch_bounds__(64, 2) k_linear_f(const LinearFOnlyArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLdsF* jw = JAC ? reinterpret_cast<JacobiLdsF*>(smem + base) : nullptr;
    double* extra = smem + base + (JAC ? ((JACOBI_F_LDS_DOUBLES + 1) & ~1) : 0);
    OptimFLds* og = REFINE ? reinterpret_cast<OptimFLds*>(extra) : nullptr;
    double* oxi = extra + ((OPTIMF_FIXED_DOUBLES + 1) & ~1);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;
        wave_sync();
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        int st = ST_OK, iters = 0;
        if (N < 8) {                                                         // linearF.m:35-37, optimF.m:36-38
            st = ST_TOO_FEW;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (lane < 9) { a.F21[b * 9 + lane] = qnan; a.F31[b * 9 + lane] = qnan; }
        } else {
            if (REFINE) {
                normalise3(pts, N, w->nrm);                                  // optimF.m:46-47
            } else {
                if (lane < 9) w->nrm[lane] = (lane % 3 == 0) ? 1.0 : 0.0;    // points used as given
                wave_sync();
            }
            normalise3(pts, N, w->nrm2, w->nrm);                             // linearF.m:45-46
            if (!linear_f_wave<JAC>(w, jw, pts, N, nullptr, REFINE != 0)) {
                st = ST_RETRY;
            } else {
                bool fine = true;
                if (REFINE) iters = optim_f_refine<JAC>(w, og, oxi, pts, N, &st, &fine);
                if (lane < 2) {
                    Mat3 F;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) F.m[r][c] = w->Fm[9 * lane + 3 * r + c];
                    if (REFINE) {                                            // optimF.m:72-76: de-normalise, rank 2
                        F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, lane + 1)), F), normal_matrix(w->nrm, 0));
                        double v3[3], fv[3];
                        fine = null3<JAC>(F, v3) && fine;
#pragma unroll
                        for (int r = 0; r < 3; ++r) fv[r] = F.m[r][0] * v3[0] + F.m[r][1] * v3[1] + F.m[r][2] * v3[2];
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c) F.m[r][c] -= fv[r] * v3[c];
                    }
                    double* out = (lane == 0 ? a.F21 : a.F31) + b * 9;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) out[3 * c + r] = F.m[r][c];
                }
                if (wave_any(!fine)) st = ST_RETRY;
            }
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = iters;
            a.status[b] = st;
        }
    }
}

}  // namespace tff
