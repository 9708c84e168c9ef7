// BundleAdjustment (Optimization/BundleAdjustment.m:49-216) for M = 2 .. 6 views, one wavefront per problem.  SURVEY 8(f) rank 4.
//
// ba_kernel.h is the three-view kernel the harness calls (camera 1 = [I|0], every point seen in every view).  This file is the
// reference's function as it is written, for any number of views:
//   * per-view Normalize2Ddata folded into the calibration (:52-56);
//   * optional initial triangulation IN THE FRAME OF THE GIVEN POSES, over the views that see the point (:59-77);
//   * change of coordinates to camera 1 (:80-86), Euler angles of every rotation (:89-96), variables
//     [angles(:,2:M), translations(:,2:M), Reconst] (:100);
//   * residual / Jacobian callback `bundleadjustment_LM` (:128-204), including its `isnan` branch (:165): what that branch
//     can see is a WHOLE view -- Normalize2Ddata.m:34-37 takes `mean` over the view's points, so one NaN entry turns every
//     point of that view (and its calibration) into NaN, and :165 then skips the view for every correspondence.  The view's
//     camera keeps its initial angles and translation (its columns of J are zero), the other views are adjusted;
//   * R = Rx*Ry*Rz, scale 1/|t2|, repr_err = norm(func(variables)) in normalised coordinates (:105-123).
// The optimiser is the Levenberg-Marquardt loop of oracle/ba_oracle.py (lsqnonlin is closed source; see ba_kernel.h).
//
// Normal equations: with V_i = Jp_i'Jp_i + lambda I (3 x 3 per point), W_i = Jc_i'Jp_i (P x 3, P = 6 (M - 1) camera parameters)
//     (sum_i Jc_i'Jc_i - W_i inv(V_i) W_i' + lambda I) dc = - sum_i Jc_i' (r_i - Jp_i inv(V_i) Jp_i' r_i),
//     dX_i = -inv(V_i) Jp_i' (r_i + Jc_i dc)
// (the point blocks of the reference's dense 2MN x (P + 3N) system eliminated exactly).  The P (P + 1) / 2 + P sums are
// accumulated one correspondence per lane, thirty per sweep; the Jacobians are recomputed from the parameters in every sweep,
// nothing per correspondence is stored except the points and their trial values (6N doubles of LDS).
#pragma once
#include "ba_kernel.h"

namespace tff {

constexpr int BAV_MIN_VIEWS = 2, BAV_MAX_VIEWS = 6;

template <int M> struct BavDims {
    static constexpr int C = M - 1;                        // cameras with parameters
    static constexpr int P = 6 * C;                        // angles of cameras 2..M, then their translations   (:100)
    static constexpr int NT = P * (P + 1) / 2;             // lower triangle of the Schur matrix
    static constexpr int NH = NT + P;                      // ... then the right-hand sides
    static constexpr int SWEEPS = (NH + 29) / 30;
};
// parameter p -> camera (0: view 2) and column of its 2 x 6 Jacobian (angles 0..2, translation 3..5)
template <int M> __host__ __device__ constexpr int bav_cam_of(int p) { return (p < 3 * (M - 1)) ? p / 3 : (p - 3 * (M - 1)) / 3; }
template <int M> __host__ __device__ constexpr int bav_col_of(int p) { return (p < 3 * (M - 1)) ? p % 3 : 3 + (p - 3 * (M - 1)) % 3; }

template <int M> struct BavCams {                          // per parameter set (wave-uniform, LDS)
    double c[6 * (M - 1)];
    double KR[M - 1][9];
    double Kt[M - 1][3];
    double KdR[M - 1][3][9];
};
template <int M> struct BavLds {
    double K[M][9];                                        // Normal_j * K_j, row-major   (:55)
    double nrm[3 * M];                                     // s, ox, oy of every view (Normalize2Ddata.m:36-37)
    double seen[M];                                        // 1: the view has no NaN / Inf entry; 0: :165 skips it
    double P0[M][12];                                      // cameras of the initial triangulation, row-major 3 x 4   (:70-71)
    double R1[9], t1[3];                                   // change_coord = R_t_0(1:3,:)   (:80)
    BavCams<M> cur, trial;
    double H[30 * BavDims<M>::SWEEPS + 2];
    double Mx[BavDims<M>::P * (BavDims<M>::P + 1)];
    double dc[BavDims<M>::P + 2];
};
template <int M> constexpr int bav_lds_doubles() { return (int)((sizeof(BavLds<M>) / sizeof(double) + 1) & ~(size_t)1); }
template <int M> inline size_t bav_lds_bytes(int N) { return ((size_t)bav_lds_doubles<M>() + 6 * (size_t)N) * sizeof(double); }

struct BavArgs {
    const double* calm; long calm_stride;                  // 9M doubles per set: MATLAB's 3M x 3 CalM, column-major
    const double* Rt_in;                                   // B x 12M: MATLAB's 3M x 4 R_t_0, column-major (camera 1 included)
    const double* corresp; long B; int N;                  // B x N x 2M: column i of the 2M x N Corresp is contiguous
    const double* reconst0;                                // B x 3N or null: triangulate first (:59-77)
    double* Rt; double* reconst;                           // B x 12M (3M x 4 column-major, R_t(1:3,:) = eye(3,4)), B x 3N or null
    int* iter; double* repr_err; int* status;
};

template <int M> struct PtV { double v[2 * M]; };
template <int M>
__device__ __forceinline__ PtV<M> bav_load(const double* pts, int i, const double* nrm) {       // normalised observation   (:54)
    PtV<M> p;
    const double2* q = reinterpret_cast<const double2*>(pts + 2 * M * (long)i);
#pragma unroll
    for (int v = 0; v < M; ++v) {
        const double2 a = q[v];
        p.v[2 * v] = nrm[3 * v] * a.x + nrm[3 * v + 1];
        p.v[2 * v + 1] = nrm[3 * v] * a.y + nrm[3 * v + 2];
    }
    return p;
}

// Normalize2Ddata.m:33-39 per view (the arithmetic of pose_common.h::normalise3)
template <int M>
__device__ inline void bav_normalise(const double* pts, int N, double* nrm, double* seen) {
    const int lane = lane_id();
    double s[2 * M];
#pragma unroll
    for (int k = 0; k < 2 * M; ++k) s[k] = 0.0;
    for (int i = lane; i < N; i += WAVE) {
#pragma unroll
        for (int k = 0; k < 2 * M; ++k) s[k] += pts[2 * M * (long)i + k];
    }
    double c[2 * M];
#pragma unroll
    for (int k = 0; k < 2 * M; ++k) c[k] = wave_sum(s[k]) / (double)N;      // points0 = mean(points,2)
    double d[M];
#pragma unroll
    for (int v = 0; v < M; ++v) d[v] = 0.0;
    for (int i = lane; i < N; i += WAVE) {
#pragma unroll
        for (int v = 0; v < M; ++v) {
            const double dx = pts[2 * M * (long)i + 2 * v] - c[2 * v], dy = pts[2 * M * (long)i + 2 * v + 1] - c[2 * v + 1];
            d[v] += sqrt(dx * dx + dy * dy);
        }
    }
    const double r2 = sqrt(2.0);
#pragma unroll
    for (int v = 0; v < M; ++v) {
        const double norm0 = wave_sum(d[v]) / (double)N;                   // :35
        if (lane == 0) {
            const double sc = r2 / norm0, ox = -r2 * c[2 * v] / norm0, oy = -r2 * c[2 * v + 1] / norm0;
            nrm[3 * v + 0] = sc; nrm[3 * v + 1] = ox; nrm[3 * v + 2] = oy;
            seen[v] = (fabs(sc) <= 1.79e308 && fabs(ox) <= 1.79e308 && fabs(oy) <= 1.79e308) ? 1.0 : 0.0;
        }
    }
    wave_sync();
}

template <int M>
__device__ inline void bav_prepare(const BavLds<M>* L, BavCams<M>* cam) {
    constexpr int C = M - 1;
    if (lane_id() < C) {
        const int j = lane_id();
        ba_prepare_camera(L->K[j + 1], cam->c + 3 * j, cam->c + 3 * C + 3 * j, cam->KR[j], cam->Kt[j], cam->KdR[j]);
    }
    wave_sync();
}

// residuals (2M) of one correspondence; with JAC: Jp (2M x 3) and the camera Jacobians of views 2..M (2 x 6 each: angles, translation).
// A view that is not seen contributes zeros (:165-167).
template <int M, bool JAC>
__device__ __forceinline__ void bav_point(const BavLds<M>* L, const BavCams<M>* cam, const PtV<M>& x, const double (&X)[3], double (&r)[2 * M],
                                          double (&Jp)[2 * M][3], double (&Jc)[M - 1][2][6]) {
#pragma unroll
    for (int v = 0; v < M; ++v) {
        const bool vis = wave_uniform(L->seen[v]) != 0.0;
        if (!vis) {
            r[2 * v] = 0.0; r[2 * v + 1] = 0.0;
            if (JAC) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { Jp[2 * v][k] = 0.0; Jp[2 * v + 1][k] = 0.0; }
                if (v > 0) {
#pragma unroll
                    for (int m = 0; m < 6; ++m) { Jc[v - 1][0][m] = 0.0; Jc[v - 1][1][m] = 0.0; }
                }
            }
            continue;
        }
        const double* A = (v == 0) ? L->K[0] : cam->KR[v - 1];
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = A[3 * k] * X[0] + A[3 * k + 1] * X[1] + A[3 * k + 2] * X[2] + ((v == 0) ? 0.0 : cam->Kt[v - 1][k]);
        const double iz = 1.0 / p[2];
        const double gx = p[0] * iz, gy = p[1] * iz;
        r[2 * v] = x.v[2 * v] - gx;                                          // Dist(point, Gamma(P*[Point;1]))   (:176-178)
        r[2 * v + 1] = x.v[2 * v + 1] - gy;
        if (JAC) {
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
template <int M>
__device__ __forceinline__ void bav_vinv(const double (&Jp)[2 * M][3], double lambda, double (&Vi)[6]) {
    double V[6] = {lambda, 0, 0, lambda, 0, lambda};
#pragma unroll
    for (int row = 0; row < 2 * M; ++row) {
        V[0] += Jp[row][0] * Jp[row][0]; V[1] += Jp[row][0] * Jp[row][1]; V[2] += Jp[row][0] * Jp[row][2];
        V[3] += Jp[row][1] * Jp[row][1]; V[4] += Jp[row][1] * Jp[row][2]; V[5] += Jp[row][2] * Jp[row][2];
    }
    const double c00 = V[3] * V[5] - V[4] * V[4], c01 = V[2] * V[4] - V[1] * V[5], c02 = V[1] * V[4] - V[2] * V[3];
    const double idet = 1.0 / (V[0] * c00 + V[1] * c01 + V[2] * c02);
    Vi[0] = c00 * idet; Vi[1] = c01 * idet; Vi[2] = c02 * idet;
    Vi[3] = (V[0] * V[5] - V[2] * V[2]) * idet; Vi[4] = (V[1] * V[2] - V[0] * V[4]) * idet; Vi[5] = (V[0] * V[3] - V[1] * V[1]) * idet;
}

// entry E of (lower triangle of the P x P Schur matrix, then the P right-hand sides): the correspondence's term.
// (E is a template argument: the camera / column indices must be constants before the arrays are scalarised.)
template <int M, int E>
__device__ __forceinline__ void bav_entry(const double (&Jp)[2 * M][3], const double (&Jc)[M - 1][2][6], const double (&Vi)[6], const double (&s)[2 * M],
                                          double& acc) {
    using D = BavDims<M>;
    if constexpr (E < D::NT) {
        constexpr int p = tri_row_of(E), q = tri_col_of(E);
        constexpr int cp = bav_cam_of<M>(p), cq = bav_cam_of<M>(q), lp = bav_col_of<M>(p), lq = bav_col_of<M>(q);
        double Wp[3], Wq[3], Y[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Wp[k] = Jc[cp][0][lp] * Jp[2 * (cp + 1)][k] + Jc[cp][1][lp] * Jp[2 * (cp + 1) + 1][k];
            Wq[k] = Jc[cq][0][lq] * Jp[2 * (cq + 1)][k] + Jc[cq][1][lq] * Jp[2 * (cq + 1) + 1][k];
        }
        sym3_mul(Vi, Wp, Y);
        double t = -(Y[0] * Wq[0] + Y[1] * Wq[1] + Y[2] * Wq[2]);
        if constexpr (cp == cq) t += Jc[cp][0][lp] * Jc[cq][0][lq] + Jc[cp][1][lp] * Jc[cq][1][lq];
        acc += t;
    } else if constexpr (E < D::NH) {
        constexpr int p = E - D::NT;
        constexpr int cp = bav_cam_of<M>(p), lp = bav_col_of<M>(p);
        acc -= Jc[cp][0][lp] * s[2 * (cp + 1)] + Jc[cp][1][lp] * s[2 * (cp + 1) + 1];
    }
}
template <int M, int E0, int K = 0>
__device__ __forceinline__ void bav_chunk(const double (&Jp)[2 * M][3], const double (&Jc)[M - 1][2][6], const double (&Vi)[6], const double (&s)[2 * M],
                                          double (&acc)[32]) {
    if constexpr (K < 30) {
        bav_entry<M, E0 + K>(Jp, Jc, Vi, s, acc[K]);
        bav_chunk<M, E0, K + 1>(Jp, Jc, Vi, s, acc);
    }
}

// one accumulation sweep: entries [30 SW, 30 SW + 30)
template <int M, int SW>
__device__ inline void bav_sweep(BavLds<M>* L, const double* pts, const double* Xs, int N, double lambda) {
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const PtV<M> x = bav_load<M>(pts, i, L->nrm);
        const double X[3] = {Xs[3 * i], Xs[3 * i + 1], Xs[3 * i + 2]};
        double r[2 * M], Jp[2 * M][3], Jc[M - 1][2][6], Vi[6];
        bav_point<M, true>(L, &L->cur, x, X, r, Jp, Jc);
        bav_vinv<M>(Jp, lambda, Vi);
        double g3[3] = {0.0, 0.0, 0.0}, u[3], s[2 * M];                      // s = r - Jp inv(V) Jp' r
#pragma unroll
        for (int row = 0; row < 2 * M; ++row) { g3[0] += Jp[row][0] * r[row]; g3[1] += Jp[row][1] * r[row]; g3[2] += Jp[row][2] * r[row]; }
        sym3_mul(Vi, g3, u);
#pragma unroll
        for (int row = 0; row < 2 * M; ++row) s[row] = r[row] - (Jp[row][0] * u[0] + Jp[row][1] * u[1] + Jp[row][2] * u[2]);
        bav_chunk<M, 30 * SW>(Jp, Jc, Vi, s, acc);
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0 && idx < 30) L->H[30 * SW + idx] = tot;
}
template <int M, int SW = 0>
__device__ inline void bav_sweeps(BavLds<M>* L, const double* pts, const double* Xs, int N, double lambda) {
    if constexpr (SW < BavDims<M>::SWEEPS) {
        bav_sweep<M, SW>(L, pts, Xs, N, lambda);
        bav_sweeps<M, SW + 1>(L, pts, Xs, N, lambda);
    }
}

template <int M>
__device__ inline double bav_cost(const BavLds<M>* L, const BavCams<M>* cam, const double* pts, const double* Xs, int N) {
    double S = 0.0;
    for (int i = lane_id(); i < N; i += WAVE) {
        const PtV<M> x = bav_load<M>(pts, i, L->nrm);
        const double X[3] = {Xs[3 * i], Xs[3 * i + 1], Xs[3 * i + 2]};
        double r[2 * M], Jp[2 * M][3], Jc[M - 1][2][6];
        bav_point<M, false>(L, cam, x, X, r, Jp, Jc);
#pragma unroll
        for (int k = 0; k < 2 * M; ++k) S += r[k] * r[k];
    }
    return wave_sum(S);
}

template <int M>
__device__ inline void bav_store_nan(const BavArgs& a, long b, int N) {
    const double qnan = __builtin_nan("");
    for (int e = lane_id(); e < 12 * M; e += WAVE) a.Rt[b * 12 * M + e] = qnan;
    if (a.reconst) for (int e = lane_id(); e < 3 * N; e += WAVE) a.reconst[b * 3 * (long)N + e] = qnan;
    if (lane_id() == 0) {
        if (a.iter) a.iter[b] = 0;
        if (a.repr_err) a.repr_err[b] = qnan;
    }
}

template <int M>
__global__ void __launch_bounds__(64, 1) k_bundle_adjust_views(const BavArgs a) {
    using D = BavDims<M>;
    constexpr int C = D::C, P = D::P;
    TFF_DYNAMIC_LDS(double, smem);
    BavLds<M>* L = reinterpret_cast<BavLds<M>*>(smem);
    double* Xc = smem + bav_lds_doubles<M>();                                // current points (3N), then trial points (3N)
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int N = opaque_int(a.N);
        double* Xt = Xc + 3 * N;
        const double* pts = a.corresp + b * 2 * M * (long)N;
        const double* calm = a.calm + b * a.calm_stride;
        const double* Rt0 = a.Rt_in + b * 12 * M;                            // R_t_0(3j + r, c) = Rt0[3j + r + 3M c]
        wave_sync();
        bav_normalise<M>(pts, N, L->nrm, L->seen);                           // :52-54
        for (int e = lane; e < 9 * M; e += WAVE) {                           // CalM(3j-2:3j,:) = Normal * CalM(...)   (:55)
            const int v = e / 9, r = (e % 9) / 3, c = e % 3;
            const Mat3 Nm = normal_matrix(L->nrm, v);
            const double k0 = calm[(3 * v + 0) + 3 * M * c], k1 = calm[(3 * v + 1) + 3 * M * c], k2 = calm[(3 * v + 2) + 3 * M * c];
            L->K[v][3 * r + c] = Nm.m[r][0] * k0 + Nm.m[r][1] * k1 + Nm.m[r][2] * k2;
        }
        if (lane < 12) {                                                     // change_coord = R_t_0(1:3,:)   (:80)
            const int r = lane % 3, c = lane / 3;
            const double v = Rt0[r + 3 * M * c];
            if (c < 3) L->R1[3 * r + c] = v; else L->t1[r] = v;
        }
        wave_sync();
        int nseen = 0;
#pragma unroll
        for (int v = 0; v < M; ++v) nseen += (wave_uniform(L->seen[v]) != 0.0) ? 1 : 0;
        int status = ST_OK;
        if (a.reconst0) {
            for (int e = lane; e < 3 * N; e += WAVE) Xc[e] = a.reconst0[b * 3 * (long)N + e];
        } else {                                                             // initial triangulation with the given poses   (:59-77)
            if (nseen < 2) {                                                 // triangulation3D.m:36-38 returns nothing for fewer than two cameras: the reference stops
                bav_store_nan<M>(a, b, N);
                if (lane == 0 && a.status) a.status[b] = ST_TOO_FEW;
                continue;
            }
            for (int e = lane; e < 12 * M; e += WAVE) {                      // cameras{j} = CalM(3j-2:3j,:) * R_t_0(3j-2:3j,:)
                const int j = e / 12, r = (e % 12) >> 2, c = e & 3;
                L->P0[j][e % 12] = L->K[j][3 * r] * Rt0[3 * j + 0 + 3 * M * c] + L->K[j][3 * r + 1] * Rt0[3 * j + 1 + 3 * M * c] +
                                   L->K[j][3 * r + 2] * Rt0[3 * j + 2 + 3 * M * c];
            }
            wave_sync();
            bool conv_all = true;
            for (int i = lane; i < N; i += WAVE) {
                const PtV<M> x = bav_load<M>(pts, i, L->nrm);
                double S[4][4], X[4];
                tri_zero(S);
#pragma unroll
                for (int v = 0; v < M; ++v) {
                    if (wave_uniform(L->seen[v]) == 0.0) continue;           // :65-67
                    double Pv[12];
#pragma unroll
                    for (int c = 0; c < 12; ++c) Pv[c] = L->P0[v][c];
                    tri_accum(S, Pv, x.v[2 * v], x.v[2 * v + 1]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = r + 1; c < 4; ++c) S[r][c] = S[c][r];
                bool conv;
                spd_min_eigvec<4>(S, X, 200, &conv);                         // V(:,4) of svd(ls_matrix)   (triangulation3D.m:61-62)
                conv_all = conv_all && conv;
                const double iw = 1.0 / X[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) Xc[3 * i + k] = X[k] * iw;       // Reconst0(:,i) = X(1:3)/X(4)   (:75)
            }
            if (wave_any(!conv_all)) status = ST_NONFINITE;                  // two coincident smallest singular values: no defined start
        }
        wave_sync();
        // ---- change of coordinates so that the first pose is [Id 0] (:80-86), angles (:92-94) and translations (:95) ----
        if (lane < C) {
            const int j = lane + 1;
            double Rj[9], tj[3], Rn[9];
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rj[3 * r + c] = Rt0[3 * j + r + 3 * M * c]; tj[r] = Rt0[3 * j + r + 3 * M * 3]; }
            double w1[3];                                                    // change_coord(:,1:3).' * change_coord(:,4)
            for (int k = 0; k < 3; ++k) w1[k] = L->R1[0 + k] * L->t1[0] + L->R1[3 + k] * L->t1[1] + L->R1[6 + k] * L->t1[2];
            for (int r = 0; r < 3; ++r) {
                tj[r] = tj[r] - (Rj[3 * r] * w1[0] + Rj[3 * r + 1] * w1[1] + Rj[3 * r + 2] * w1[2]);
                for (int c = 0; c < 3; ++c) Rn[3 * r + c] = Rj[3 * r] * L->R1[3 * c] + Rj[3 * r + 1] * L->R1[3 * c + 1] + Rj[3 * r + 2] * L->R1[3 * c + 2];
            }
            const double R12 = Rn[5], R22 = Rn[8], R02 = Rn[2], R01 = Rn[1], R00 = Rn[0];
            L->cur.c[3 * lane + 0] = -atan2(R12, R22);
            L->cur.c[3 * lane + 1] = -atan2(-R02, sqrt(R12 * R12 + R22 * R22));
            L->cur.c[3 * lane + 2] = -atan2(R01, R00);
            for (int k = 0; k < 3; ++k) L->cur.c[3 * C + 3 * lane + k] = tj[k];
        }
        for (int i = lane; i < N; i += WAVE) {                               // Reconst0 = R1 * Reconst0 + t1   (:86)
            const double X0 = Xc[3 * i], X1 = Xc[3 * i + 1], X2 = Xc[3 * i + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) Xc[3 * i + r] = L->R1[3 * r] * X0 + L->R1[3 * r + 1] * X1 + L->R1[3 * r + 2] * X2 + L->t1[r];
        }
        wave_sync();
        bav_prepare<M>(L, &L->cur);
        // ---- Levenberg-Marquardt (oracle/ba_oracle.py: levenberg_marquardt) ----
        double lambda = BA_INIT_DAMPING;
        double S = bav_cost<M>(L, &L->cur, pts, Xc, N);
        int it = 0;
#pragma unroll 1
        while (it < BA_MAX_ITER && status == ST_OK) {
            bav_sweeps<M>(L, pts, Xc, N, lambda);
            wave_sync();
            for (int e = lane; e < P * (P + 1); e += WAVE) {                 // (sum Jc'Jc - W inv(V) W' + lambda I) dc = -sum Jc' s
                const int r = e / (P + 1), c = e % (P + 1);
                double v;
                if (c == P) v = L->H[D::NT + r];
                else {
                    const int hi = (r > c) ? r : c, lo = (r > c) ? c : r;
                    v = L->H[hi * (hi + 1) / 2 + lo];
                    if (r == c) {                                            // a camera nobody sees has an all-zero block: dc = 0 for it, whatever lambda is
                        const int cam = (r < 3 * C) ? r / 3 : (r - 3 * C) / 3;
                        v = (L->seen[cam + 1] != 0.0) ? v + lambda : 1.0;
                    }
                }
                L->Mx[e] = v;
            }
            wave_sync();
            const bool ok = wave_solve_gj<P>(L->Mx, L->dc);
            if (!ok) { status = ST_NONFINITE; break; }
            if (lane < P) L->trial.c[lane] = L->cur.c[lane] + L->dc[lane];
            wave_sync();
            bav_prepare<M>(L, &L->trial);
            // dX_i = -inv(V_i) Jp_i' (r_i + Jc_i dc); trial cost; norms for the step test
            double St = 0.0, step2 = 0.0, x2 = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                const PtV<M> x = bav_load<M>(pts, i, L->nrm);
                const double X[3] = {Xc[3 * i], Xc[3 * i + 1], Xc[3 * i + 2]};
                double r[2 * M], Jp[2 * M][3], Jc[M - 1][2][6], Vi[6];
                bav_point<M, true>(L, &L->cur, x, X, r, Jp, Jc);
                bav_vinv<M>(Jp, lambda, Vi);
#pragma unroll
                for (int cj = 0; cj < C; ++cj)
#pragma unroll
                    for (int row = 0; row < 2; ++row) {
                        double e = 0.0;
#pragma unroll
                        for (int m = 0; m < 3; ++m) e += Jc[cj][row][m] * wave_uniform(L->dc[3 * cj + m]) + Jc[cj][row][3 + m] * wave_uniform(L->dc[3 * C + 3 * cj + m]);
                        r[2 + 2 * cj + row] += e;
                    }
                double g3[3] = {0.0, 0.0, 0.0}, dX[3];
#pragma unroll
                for (int row = 0; row < 2 * M; ++row) { g3[0] += Jp[row][0] * r[row]; g3[1] += Jp[row][1] * r[row]; g3[2] += Jp[row][2] * r[row]; }
                sym3_mul(Vi, g3, dX);
                double Xn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { Xn[k] = X[k] - dX[k]; Xt[3 * i + k] = Xn[k]; step2 += dX[k] * dX[k]; x2 += X[k] * X[k]; }
                double rt[2 * M], Jp2[2 * M][3], Jc2[M - 1][2][6];
                bav_point<M, false>(L, &L->trial, x, Xn, rt, Jp2, Jc2);
#pragma unroll
                for (int k = 0; k < 2 * M; ++k) St += rt[k] * rt[k];
            }
            St = wave_sum(St); step2 = wave_sum(step2); x2 = wave_sum(x2);
            for (int k = 0; k < P; ++k) { const double dk = wave_uniform(L->dc[k]), ck = wave_uniform(L->cur.c[k]); step2 += dk * dk; x2 += ck * ck; }
            const bool small_step = sqrt(step2) < BA_TOL_X * (1.4901161193847656e-08 + sqrt(x2));
            if (St < S) {                                                    // successful step
                ++it;
                const bool done = fabs(St - S) <= BA_TOL_FUN * S || small_step;
                for (int e = lane; e < 3 * N; e += WAVE) Xc[e] = Xt[e];
                if (lane < P) L->cur.c[lane] = L->trial.c[lane];
                wave_sync();
                bav_prepare<M>(L, &L->cur);
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
        const double t2x = L->cur.c[3 * C], t2y = L->cur.c[3 * C + 1], t2z = L->cur.c[3 * C + 2];
        const double scale = rsqrt(t2x * t2x + t2y * t2y + t2z * t2z);
        double* out = a.Rt + b * 12 * M;                                     // R_t(3j + r, c) = out[3j + r + 3M c]
        if (lane < 12) { const int r = lane % 3, c = lane / 3; out[r + 3 * M * c] = (r == c) ? 1.0 : 0.0; }
        if (lane < C) {
            const double* ang = L->cur.c + 3 * lane;
            const double cx = cos(ang[0]), sx = sin(ang[0]), cy = cos(ang[1]), sy = sin(ang[1]), cz = cos(ang[2]), sz = sin(ang[2]);
            Mat3 Rx{{{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}}, Ry{{{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}}, Rz{{{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}};
            const Mat3 R = mat3_mul(mat3_mul(Rx, Ry), Rz);
            const int j = lane + 1;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out[3 * j + r + 3 * M * c] = R.m[r][c]; out[3 * j + r + 3 * M * 3] = scale * L->cur.c[3 * C + 3 * lane + r]; }
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

}  // namespace tff
