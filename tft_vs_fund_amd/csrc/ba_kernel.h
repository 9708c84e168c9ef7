// BundleAdjustment (Optimization/BundleAdjustment.m:49-216) for three views, one wavefront per triplet.  SURVEY 8(f) rank 4.
//
// Variables as the reference orders them (:100): angles of cameras 2, 3 (Rx*Ry*Rz), their translations, the N space points;
// camera 1 is K1 [I|0].  Residual: observed minus projected point in the per-view normalised frame (Normalize2Ddata folded into
// the calibration, :52-56), analytic Jacobian of `bundleadjustment_LM` (:128-204).  Parity is UNPINNED twice over -- MATLAB, and
// its closed-source lsqnonlin: the Levenberg-Marquardt loop here is the one stated in oracle/ba_oracle.py (lsqnonlin's documented
// LM defaults: damping 0.01, x10 / /10, FunctionTolerance = StepTolerance = 1e-6, 400 iterations) and is checked against that
// restatement and against MINPACK at the converged optimum.
//
// The reference builds the dense 6N x (12+3N) Jacobian.  Its normal equations have the usual arrow structure, and with
//     V_i = Jp_i' Jp_i + lambda I (3x3 per point),   Q_i = Jp_i inv(V_i) Jp_i' (6x6),   P_i = I - Q_i
// the point blocks eliminate exactly:
//     (sum_i Jc_i' P_i Jc_i + lambda I) dc = - sum_i Jc_i' P_i r_i,      dX_i = -inv(V_i) Jp_i' (r_i + Jc_i dc),
// 78 + 12 sums accumulated one correspondence per lane in three 30-accumulator sweeps (Jacobians are recomputed from the
// parameters in every sweep: nothing per correspondence is stored except the points and their trial values, 6N doubles of LDS).
#pragma once
#include "gh_kernel.h"

namespace tff {

constexpr int BA_MAX_ITER = 400;
constexpr double BA_TOL_FUN = 1e-6, BA_TOL_X = 1e-6, BA_INIT_DAMPING = 0.01;

struct BaCams {                    // everything a projection and its derivatives need, per parameter set (wave-uniform, LDS)
    double c[12];                  // angles2, angles3, t2, t3   (BundleAdjustment.m:100)
    double KR[2][9];               // K_j R_j, row-major
    double Kt[2][3];               // K_j t_j
    double KdR[2][3][9];           // K_j dR_j/d angle_m   (:148-151, :189-190)
};
struct BaLds {
    double K[3][9];                // normalised calibration Normal_j * K_j, row-major   (:55)
    BaCams cur, trial;
    double H[96];                  // 78 + 12 accumulated sums
    double M[12 * 13];             // augmented Schur system
    double dc[12];
};
constexpr int BA_LDS_DOUBLES = (int)(sizeof(BaLds) / sizeof(double));

struct BaArgs {
    const double* calm; long calm_stride;
    const double* Rt2_in; const double* Rt3_in;      // B x 12 (3x4 column-major): the poses to refine, camera 1 = [I|0]
    const double* corresp; long B; int N;
    const double* reconst0;                          // B x 3N or null: triangulate first (:59-77)
    double* Rt2; double* Rt3; double* reconst;       // outputs (reconst may be null)
    int* iter; double* repr_err; int* status;
};

// lane 0: rotation products of one camera from its three angles
__device__ inline void ba_prepare_camera(const double* K, const double* ang, const double* t, double* KR, double* Kt, double (*KdR)[9]) {
    const double cx = cos(ang[0]), sx = sin(ang[0]), cy = cos(ang[1]), sy = sin(ang[1]), cz = cos(ang[2]), sz = sin(ang[2]);
    Mat3 Rx{{{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}}, Ry{{{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}}, Rz{{{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}};
    Mat3 Dx{{{0, 0, 0}, {0, -sx, -cx}, {0, cx, -sx}}}, Dy{{{-sy, 0, cy}, {0, 0, 0}, {-cy, 0, -sy}}}, Dz{{{-sz, -cz, 0}, {cz, -sz, 0}, {0, 0, 0}}};
    Mat3 Km;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Km.m[r][c] = K[3 * r + c];
    const Mat3 R = mat3_mul(mat3_mul(Rx, Ry), Rz);
    const Mat3 A = mat3_mul(Km, R), A0 = mat3_mul(Km, mat3_mul(mat3_mul(Dx, Ry), Rz)), A1 = mat3_mul(Km, mat3_mul(mat3_mul(Rx, Dy), Rz)),
               A2 = mat3_mul(Km, mat3_mul(mat3_mul(Rx, Ry), Dz));
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { KR[3 * r + c] = A.m[r][c]; KdR[0][3 * r + c] = A0.m[r][c]; KdR[1][3 * r + c] = A1.m[r][c]; KdR[2][3 * r + c] = A2.m[r][c]; }
        Kt[r] = Km.m[r][0] * t[0] + Km.m[r][1] * t[1] + Km.m[r][2] * t[2];
    }
}
__device__ inline void ba_prepare(const BaLds* L, BaCams* cam) {
    if (lane_id() < 2) {
        const int j = lane_id();
        ba_prepare_camera(L->K[j + 1], cam->c + 3 * j, cam->c + 6 + 3 * j, cam->KR[j], cam->Kt[j], cam->KdR[j]);
    }
    wave_sync();
}

// residuals (6) of one correspondence; with JAC: Jp (6x3) and the camera Jacobians of views 2, 3 (2 x 6 each: angles, translation)
template <bool JAC>
__device__ __forceinline__ void ba_point(const BaLds* L, const BaCams* cam, const Pt6& x, const double (&X)[3], double (&r)[6],
                                         double (&Jp)[6][3], double (&Jc)[2][2][6]) {
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double* A = (v == 0) ? L->K[0] : cam->KR[v - 1];
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = A[3 * k] * X[0] + A[3 * k + 1] * X[1] + A[3 * k + 2] * X[2] + ((v == 0) ? 0.0 : cam->Kt[v - 1][k]);
        const double iz = 1.0 / p[2];
        const double gx = p[0] * iz, gy = p[1] * iz;
        r[2 * v] = x.v[2 * v] - gx;                                          // Dist(point, Gamma(P*[Point;1]))   (:176-178)
        r[2 * v + 1] = x.v[2 * v + 1] - gy;
        if (JAC) {
            // -dgamma * M for a 3 x k matrix M: rows (-(M0 - gx M2) iz, -(M1 - gy M2) iz)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Jp[2 * v][k] = -(A[k] - gx * A[6 + k]) * iz;                 // respect 3d point: P(:,1:3)   (:184)
                Jp[2 * v + 1][k] = -(A[3 + k] - gy * A[6 + k]) * iz;
            }
            if (v > 0) {
                const double* Kj = L->K[v];
#pragma unroll
                for (int m = 0; m < 3; ++m) {                                // respect rotation (angles)   (:191-192)
                    const double* D = cam->KdR[v - 1][m];
                    const double d0 = D[0] * X[0] + D[1] * X[1] + D[2] * X[2], d1 = D[3] * X[0] + D[4] * X[1] + D[5] * X[2],
                                 d2 = D[6] * X[0] + D[7] * X[1] + D[8] * X[2];
                    Jc[v - 1][0][m] = -(d0 - gx * d2) * iz;
                    Jc[v - 1][1][m] = -(d1 - gy * d2) * iz;
                    Jc[v - 1][0][3 + m] = -(Kj[m] - gx * Kj[6 + m]) * iz;    // respect translation: K   (:188)
                    Jc[v - 1][1][3 + m] = -(Kj[3 + m] - gy * Kj[6 + m]) * iz;
                }
            }
        }
    }
}
// inv(Jp'Jp + lambda I), packed symmetric 3x3 (xx xy xz yy yz zz)
__device__ __forceinline__ void ba_vinv(const double (&Jp)[6][3], double lambda, double (&Vi)[6]) {
    double V[6] = {lambda, 0, 0, lambda, 0, lambda};
#pragma unroll
    for (int row = 0; row < 6; ++row) {
        V[0] += Jp[row][0] * Jp[row][0]; V[1] += Jp[row][0] * Jp[row][1]; V[2] += Jp[row][0] * Jp[row][2];
        V[3] += Jp[row][1] * Jp[row][1]; V[4] += Jp[row][1] * Jp[row][2]; V[5] += Jp[row][2] * Jp[row][2];
    }
    const double c00 = V[3] * V[5] - V[4] * V[4], c01 = V[2] * V[4] - V[1] * V[5], c02 = V[1] * V[4] - V[2] * V[3];
    const double idet = 1.0 / (V[0] * c00 + V[1] * c01 + V[2] * c02);
    Vi[0] = c00 * idet; Vi[1] = c01 * idet; Vi[2] = c02 * idet;
    Vi[3] = (V[0] * V[5] - V[2] * V[2]) * idet; Vi[4] = (V[1] * V[2] - V[0] * V[4]) * idet; Vi[5] = (V[0] * V[3] - V[1] * V[1]) * idet;
}
__device__ __forceinline__ void sym3_mul(const double (&S)[6], const double (&a)[3], double (&o)[3]) {
    o[0] = S[0] * a[0] + S[1] * a[1] + S[2] * a[2];
    o[1] = S[1] * a[0] + S[3] * a[1] + S[4] * a[2];
    o[2] = S[2] * a[0] + S[4] * a[1] + S[5] * a[2];
}
// parameter index p (0..11: angles2, angles3, t2, t3) -> camera (0: view 2, 1: view 3) and column of its 2 x 6 Jacobian
__host__ __device__ constexpr int ba_cam_of(int p) { return (p / 3) & 1; }
__host__ __device__ constexpr int ba_col_of(int p) { return (p % 3) + 3 * (p / 6); }

// one accumulation sweep: entries [30 SW, 30 SW + 30) of (lower triangle of the 12 x 12 Schur matrix, then the 12 right-hand sides)
template <int SW>
__device__ inline void ba_sweep(BaLds* L, const double* pts, const double* nrm, const double* Xs, int N, double lambda) {
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 x = premap(load_pt(pts, i), nrm);
        const double X[3] = {Xs[3 * i], Xs[3 * i + 1], Xs[3 * i + 2]};
        double r[6], Jp[6][3], Jc[2][2][6], Vi[6];
        ba_point<true>(L, &L->cur, x, X, r, Jp, Jc);
        ba_vinv(Jp, lambda, Vi);
        // P = I - Jp inv(V) Jp' restricted to the rows of views 2, 3 (rows 2..5), and s = P r on those rows (r uses all six rows)
        double G[6][3];                                                      // Jp inv(V)
#pragma unroll
        for (int row = 0; row < 6; ++row) sym3_mul(Vi, Jp[row], G[row]);
        double Pm[4][4], s[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const double q = G[2 + a][0] * Jp[2 + b][0] + G[2 + a][1] * Jp[2 + b][1] + G[2 + a][2] * Jp[2 + b][2];
                Pm[a][b] = Pm[b][a] = ((a == b) ? 1.0 : 0.0) - q;
            }
            double sq = 0.0;
#pragma unroll
            for (int row = 0; row < 6; ++row) sq += (G[2 + a][0] * Jp[row][0] + G[2 + a][1] * Jp[row][1] + G[2 + a][2] * Jp[row][2]) * r[row];
            s[a] = r[2 + a] - sq;
        }
#pragma unroll
        for (int k = 0; k < 30; ++k) {
            constexpr int e0 = 30 * SW;
            const int e = e0 + k;
            if (e < 78) {
                const int p = tri_row_of(e), q = tri_col_of(e);              // constants after unrolling
                const int cp = ba_cam_of(p), cq = ba_cam_of(q), lp = ba_col_of(p), lq = ba_col_of(q);
                double t = 0.0;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) t += Jc[cp][a][lp] * Pm[2 * cp + a][2 * cq + b] * Jc[cq][b][lq];
                acc[k] += t;
            } else if (e < 90) {
                const int p = e - 78;
                const int cp = ba_cam_of(p), lp = ba_col_of(p);
                acc[k] -= Jc[cp][0][lp] * s[2 * cp] + Jc[cp][1][lp] * s[2 * cp + 1];
            }
        }
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0 && idx < 30) L->H[30 * SW + idx] = tot;
}

// sum of squared residuals of parameter set `cam` with points Xs
__device__ inline double ba_cost(const BaLds* L, const BaCams* cam, const double* pts, const double* nrm, const double* Xs, int N) {
    double S = 0.0;
    for (int i = lane_id(); i < N; i += WAVE) {
        const Pt6 x = premap(load_pt(pts, i), nrm);
        const double X[3] = {Xs[3 * i], Xs[3 * i + 1], Xs[3 * i + 2]};
        double r[6], Jp[6][3], Jc[2][2][6];
        ba_point<false>(L, cam, x, X, r, Jp, Jc);
#pragma unroll
        for (int k = 0; k < 6; ++k) S += r[k] * r[k];
    }
    return wave_sum(S);
}

__global__ void __launch_bounds__(64, 1) k_bundle_adjust(const BaArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    BaLds* L = reinterpret_cast<BaLds*>(smem + base);
    double* Xc = smem + base + ((BA_LDS_DOUBLES + 1) & ~1);                  // current points (3N), then trial points (3N)
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        double* Xt = Xc + 3 * N;
        const double* pts = a.corresp + b * 6 * (long)N;
        wave_sync();
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        wave_sync();
        normalise3(pts, N, w->nrm);                                          // :52-54
        if (lane < 27) {                                                     // CalM(3j-2:3j,:) = Normal * CalM(...)   (:55)
            const int v = lane / 9, r = (lane % 9) / 3, c = lane % 3;
            const Mat3 Nm = normal_matrix(w->nrm, v);
            const Mat3 Km = load_K(w->calm, v);
            L->K[v][3 * r + c] = Nm.m[r][0] * Km.m[0][c] + Nm.m[r][1] * Km.m[1][c] + Nm.m[r][2] * Km.m[2][c];
        }
        if (lane < 2) {                                                      // angles (:92-94) and translations (:95) of cameras 2, 3
            const double* Rt = (lane == 0 ? a.Rt2_in : a.Rt3_in) + b * 12;   // column-major 3x4: R(r,c) = Rt[r + 3c]
            const double R12 = Rt[1 + 6], R22 = Rt[2 + 6], R02 = Rt[0 + 6], R01 = Rt[0 + 3], R00 = Rt[0];
            L->cur.c[3 * lane + 0] = -atan2(R12, R22);
            L->cur.c[3 * lane + 1] = -atan2(-R02, sqrt(R12 * R12 + R22 * R22));
            L->cur.c[3 * lane + 2] = -atan2(R01, R00);
            for (int k = 0; k < 3; ++k) L->cur.c[6 + 3 * lane + k] = Rt[9 + k];
        }
        wave_sync();
        if (a.reconst0) {
            for (int e = lane; e < 3 * N; e += WAVE) Xc[e] = a.reconst0[b * 3 * (long)N + e];
        } else {                                                             // initial triangulation with the given poses   (:59-77)
            if (lane < 12) {
                const int r = lane >> 2, c = lane & 3;
                w->Pfin[0][lane] = (c < 3) ? L->K[0][3 * r + c] : 0.0;
            }
            if (lane < 24) {
                const int j = lane / 12, e = lane % 12, r = e >> 2, c = e & 3;
                const double* Rt = (j == 0 ? a.Rt2_in : a.Rt3_in) + b * 12;
                w->P[j][e] = L->K[j + 1][3 * r] * Rt[0 + 3 * c] + L->K[j + 1][3 * r + 1] * Rt[1 + 3 * c] + L->K[j + 1][3 * r + 2] * Rt[2 + 3 * c];
            }
            wave_sync();
            tri_pass(w, pts, N, TRI_RECONST, 1, w->P[0], w->P[1], Xc, w->nrm);
        }
        wave_sync();
        ba_prepare(L, &L->cur);
        // ---- Levenberg-Marquardt (oracle/ba_oracle.py: levenberg_marquardt) ----
        double lambda = BA_INIT_DAMPING;
        double S = ba_cost(L, &L->cur, pts, w->nrm, Xc, N);
        int it = 0, status = ST_OK;
#pragma unroll 1
        while (it < BA_MAX_ITER) {
            ba_sweep<0>(L, pts, w->nrm, Xc, N, lambda);
            ba_sweep<1>(L, pts, w->nrm, Xc, N, lambda);
            ba_sweep<2>(L, pts, w->nrm, Xc, N, lambda);
            wave_sync();
            for (int e = lane; e < 12 * 13; e += WAVE) {                     // (sum Jc' P Jc + lambda I) dc = -sum Jc' P r
                const int r = e / 13, c = e % 13;
                double v;
                if (c == 12) v = L->H[78 + r];
                else { const int hi = (r > c) ? r : c, lo = (r > c) ? c : r; v = L->H[hi * (hi + 1) / 2 + lo] + ((r == c) ? lambda : 0.0); }
                L->M[e] = v;
            }
            wave_sync();
            const bool ok = wave_solve_gj<12>(L->M, L->dc);
            if (!ok) { status = ST_NONFINITE; break; }
            if (lane < 12) L->trial.c[lane] = L->cur.c[lane] + L->dc[lane];
            wave_sync();
            ba_prepare(L, &L->trial);
            double dc[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) dc[k] = wave_uniform(L->dc[k]);
            // dX_i = -inv(V_i) Jp_i' (r_i + Jc_i dc); trial cost; norms for the step test
            double St = 0.0, step2 = 0.0, x2 = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                const Pt6 x = premap(load_pt(pts, i), w->nrm);
                const double X[3] = {Xc[3 * i], Xc[3 * i + 1], Xc[3 * i + 2]};
                double r[6], Jp[6][3], Jc[2][2][6], Vi[6];
                ba_point<true>(L, &L->cur, x, X, r, Jp, Jc);
                ba_vinv(Jp, lambda, Vi);
#pragma unroll
                for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                    for (int row = 0; row < 2; ++row) {
                        double e = 0.0;
#pragma unroll
                        for (int m = 0; m < 3; ++m) e += Jc[cj][row][m] * dc[3 * cj + m] + Jc[cj][row][3 + m] * dc[6 + 3 * cj + m];
                        r[2 + 2 * cj + row] += e;
                    }
                double g3[3] = {0.0, 0.0, 0.0}, dX[3];
#pragma unroll
                for (int row = 0; row < 6; ++row) { g3[0] += Jp[row][0] * r[row]; g3[1] += Jp[row][1] * r[row]; g3[2] += Jp[row][2] * r[row]; }
                sym3_mul(Vi, g3, dX);
                double Xn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { Xn[k] = X[k] - dX[k]; Xt[3 * i + k] = Xn[k]; step2 += dX[k] * dX[k]; x2 += X[k] * X[k]; }
                double rt[6], Jp2[6][3], Jc2[2][2][6];
                ba_point<false>(L, &L->trial, x, Xn, rt, Jp2, Jc2);
#pragma unroll
                for (int k = 0; k < 6; ++k) St += rt[k] * rt[k];
            }
            St = wave_sum(St); step2 = wave_sum(step2); x2 = wave_sum(x2);
#pragma unroll
            for (int k = 0; k < 12; ++k) { step2 += dc[k] * dc[k]; const double ck = wave_uniform(L->cur.c[k]); x2 += ck * ck; }
            const bool small_step = sqrt(step2) < BA_TOL_X * (1.4901161193847656e-08 + sqrt(x2));
            if (St < S) {                                                    // successful step
                ++it;
                const bool done = fabs(St - S) <= BA_TOL_FUN * S || small_step;
                for (int e = lane; e < 3 * N; e += WAVE) Xc[e] = Xt[e];
                if (lane < 12) L->cur.c[lane] = L->trial.c[lane];
                wave_sync();
                ba_prepare(L, &L->cur);
                S = St;
                lambda = lambda / 10.0;
                if (done) break;
            } else {
                lambda = lambda * 10.0;
                if (small_step || lambda > 1e16) break;
            }
        }
        // ---- outputs: R = Rx*Ry*Rz, scale 1/|t2| (:108-122) ----
        wave_sync();
        const double t2x = L->cur.c[6], t2y = L->cur.c[7], t2z = L->cur.c[8];
        const double scale = rsqrt(t2x * t2x + t2y * t2y + t2z * t2z);
        if (lane < 2) {
            const double* ang = L->cur.c + 3 * lane;
            const double cx = cos(ang[0]), sx = sin(ang[0]), cy = cos(ang[1]), sy = sin(ang[1]), cz = cos(ang[2]), sz = sin(ang[2]);
            Mat3 Rx{{{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}}, Ry{{{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}}, Rz{{{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}};
            const Mat3 R = mat3_mul(mat3_mul(Rx, Ry), Rz);
            double* out = (lane == 0 ? a.Rt2 : a.Rt3) + b * 12;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out[r + 3 * c] = R.m[r][c]; out[9 + r] = scale * L->cur.c[6 + 3 * lane + r]; }
        }
        if (a.reconst) for (int e = lane; e < 3 * N; e += WAVE) a.reconst[b * 3 * (long)N + e] = scale * Xc[e];
        const bool bad = !(fabs(S) <= 1.79e308);
        if (bad && status == ST_OK) status = ST_NONFINITE;
        if (lane == 0) {
            if (a.iter) a.iter[b] = it;
            if (a.repr_err) a.repr_err[b] = sqrt(S);                         // norm(func(variables))   (:105)
            if (a.status) a.status[b] = status;
        }
    }
}

inline size_t ba_lds_bytes(int N) {
    return (size_t)(((POSE_LDS_DOUBLES + 1) & ~1) + ((BA_LDS_DOUBLES + 1) & ~1) + 6 * (size_t)N) * sizeof(double);
}

}  // namespace tff
