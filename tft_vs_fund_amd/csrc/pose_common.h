// Wave-level stages shared by the pose kernels (one wavefront per triplet).
// LDS carve-up, point staging, Normalize2Ddata, epipoles, transform_TFT,
// recover_R_t with the cheirality vote, the t3 scale and the final 3-view
// triangulation.  File:line citations are into the reference tree.
#pragma once
#include "wave.h"
#include "small_la.h"
#include "wave_eig.h"

namespace tff {

// ---- flags of the pose kernels -------------------------------------------
constexpr int FLAG_RECONST = 1;       // also produce Reconst (3-view triangulation of every correspondence)
constexpr int FLAG_JACOBI = 2;        // force the Jacobi eigen-solver for the Gram matrices
constexpr int FLAG_STAGE_LDS = 4;     // correspondences are staged once in LDS (else re-read through L2)
constexpr int FLAG_ONLY_RETRY = 8;    // fix-up pass: process only triplets whose status is ST_RETRY
constexpr int FLAG_GH_EXACT = 16;     // Gauss-Helmert: always take the eigen-decomposition path for pinv(W) (A/B against the Cholesky path)

// ---- status codes (per triplet), mirroring the reference's failure modes --
constexpr int ST_OK = 0;
constexpr int ST_TOO_FEW = 1;         // N < 7 (TFT) / N < 8 (F): experiments.m:99, linearF.m:35
constexpr int ST_NONFINITE = 2;       // NaN/Inf reached the outputs (Gauss_Helmert.m:53,63)
constexpr int ST_NO_POSE = 3;         // no candidate with non-negative cheirality score (R_t_from_TFT.m:91-104)
constexpr int ST_RETRY = 100;         // internal: inverse iteration did not converge; redone by the Jacobi fix-up pass

constexpr int DBG_STRIDE = 128;       // doubles per triplet in the optional debug buffer

// ---- per-wave LDS workspace ------------------------------------------------
struct PoseLds {
    double mom[96];        // 6 x 4 x 4 moment sums of the normalised correspondences: mom[16*h + 4*i3 + i2]
    double Lp[729];        // Cholesky factor workspace (27x27 square), reused for the 15x15 / 9x9 sub-problems and as scratch
    double nrm[9];         // per view: s, ox, oy  (Normal_v = [s 0 ox; 0 s oy; 0 0 1])
    double t[27];          // tensor, vec order j + 3k + 9i  <->  T(j,k,i)   (linearTFT.m:67)
    double T1[27];         // tensor after de-normalisation (output T)
    double T2[27];         // calibrated tensor inside R_t_from_TFT
    double nullv[18];      // six 3-vectors (slice null vectors)
    double epi[6];         // e21[3], e31[3]
    double Q[18];          // orthonormal frames [e21 q q'], [e31 q q'] (row-major)
    double tp[16];
    double calm[27];       // K_v(r,c) = calm[(3v + r) + 9c]  (MATLAB 9x3 column-major)
    double Minv[18];
    double cand[2][21];    // per call: R (9, row-major), Rp (9), t (3)
    double P[4][12];       // candidate cameras K_v [R_c | t], row-major 3x4
    double candRt[4][12];  // candidate poses [R_c | t], row-major 3x4 (same order as P)
    double Rt[2][12];      // chosen poses, row-major 3x4
    double Pfin[3][12];    // final cameras
    double pa[18];         // linearTFT's a (18) -> P2, P3 of the constrained solution
    double nrm2[9];        // linearF's inner normalisation (of the already normalised points)
    double Fm[18];         // F21, F31 (row-major)
};
// extra workspace of the Jacobi kernel variant: full matrix + eigenvector matrix
struct JacobiLds {
    double A[27 * 27];
    double V[27 * 27];
};
constexpr int POSE_LDS_DOUBLES = (int)(sizeof(PoseLds) / sizeof(double));
constexpr int JACOBI_LDS_DOUBLES = (int)(sizeof(JacobiLds) / sizeof(double));

__device__ __forceinline__ Mat3 load_K(const double* calm, int v) {
    Mat3 K;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) K.m[r][c] = calm[(3 * v + r) + 9 * c];
    return K;
}

// Correspondence i of the triplet: 6 contiguous doubles [x1 y1 x2 y2 x3 y3]
// (column i of the reference's 6 x N Corresp).  `pts` may point to LDS or to
// global memory (flat addressing).
struct Pt6 { double v[6]; };
__device__ __forceinline__ Pt6 load_pt(const double* pts, int i) {
    Pt6 p;
    const double2* q = reinterpret_cast<const double2*>(pts + 6 * (long)i);
    const double2 a = q[0], b = q[1], c = q[2];
    p.v[0] = a.x; p.v[1] = a.y; p.v[2] = b.x; p.v[3] = b.y; p.v[4] = c.x; p.v[5] = c.y;
    return p;
}

// Stage the 6N doubles of one triplet into LDS with 16-byte coalesced loads.
__device__ inline void stage_points(const double* __restrict__ src, double* dst, int N) {
    const int lane = lane_id();
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    for (int i = lane; i < 3 * N; i += WAVE) d2[i] = s2[i];
    wave_sync();
}

// Minimal-sample hypotheses (config 4): gather N correspondences of one shared scene by index.
__device__ inline void gather_points(const double* __restrict__ scene, const int* __restrict__ idx, double* dst, int N) {
    const int lane = lane_id();
    for (int e = lane; e < 3 * N; e += WAVE) {
        const int i = e / 3, part = e % 3;
        const double2* s2 = reinterpret_cast<const double2*>(scene + 6 * (long)idx[i]);
        reinterpret_cast<double2*>(dst)[e] = s2[part];
    }
    wave_sync();
}

// Normalize2Ddata.m:33-39 for the three views at once.  nrm[3v..3v+2] = s, ox, oy.
// `pre` (9 doubles or null) is an affine map applied to the raw points first
// (x' = s x + ox): linearF normalises points that LinearFPoseEstimation has
// already normalised (linearF.m:45-46 after LinearFPoseEstimation.m:46-48).
__device__ __forceinline__ Pt6 premap(Pt6 p, const double* pre) {
    if (pre) {
#pragma unroll
        for (int v = 0; v < 3; ++v) { p.v[2 * v] = pre[3 * v] * p.v[2 * v] + pre[3 * v + 1]; p.v[2 * v + 1] = pre[3 * v] * p.v[2 * v + 1] + pre[3 * v + 2]; }
    }
    return p;
}
__device__ inline void normalise3(const double* pts, int N, double* nrm, const double* pre = nullptr) {
    const int lane = lane_id();
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = premap(load_pt(pts, i), pre);
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] += p.v[k];
    }
    double c[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) c[k] = wave_sum(s[k]) / (double)N;       // points0 = mean(points,2)
    double d[3] = {0, 0, 0};
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = premap(load_pt(pts, i), pre);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double dx = p.v[2 * v] - c[2 * v], dy = p.v[2 * v + 1] - c[2 * v + 1];
            d[v] += sqrt(dx * dx + dy * dy);
        }
    }
    const double r2 = sqrt(2.0);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double norm0 = wave_sum(d[v]) / (double)N;                   // :35
        if (lane == 0) {
            nrm[3 * v + 0] = r2 / norm0;                                   // :36
            nrm[3 * v + 1] = -r2 * c[2 * v] / norm0;                       // :37
            nrm[3 * v + 2] = -r2 * c[2 * v + 1] / norm0;
        }
    }
    wave_sync();
}

__device__ __forceinline__ Mat3 normal_matrix(const double* nrm, int v) {
    Mat3 M;
    M.m[0][0] = nrm[3 * v]; M.m[0][1] = 0.0; M.m[0][2] = nrm[3 * v + 1];
    M.m[1][0] = 0.0; M.m[1][1] = nrm[3 * v]; M.m[1][2] = nrm[3 * v + 2];
    M.m[2][0] = 0.0; M.m[2][1] = 0.0; M.m[2][2] = 1.0;
    return M;
}

// Epipoles of a tensor t (27, LDS): linearTFT.m:71-79 / R_t_from_TFT.m:47-55.
// Lanes 0..2 take the right null vectors of the slices, lanes 3..5 the left
// ones, lanes 0/1 then the null vector of each stacked 3x3.  epi[0..2] = e21,
// epi[3..5] = e31.  fix_sign: multiply by sign of the own third component
// (R_t_from_TFT.m:50,55).
template <int G>
__device__ __attribute__((noinline)) void epipoles_from_tensor(const double* t, double* nullv, double* epi, bool fix_sign) {
    const int lane = Group<G>::lane();
    if (lane < 6) {
        const int i = (lane < 3) ? lane : lane - 3;
        Mat3 M;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double v = t[j + 3 * k + 9 * i];
                if (lane < 3) M.m[j][k] = v; else M.m[k][j] = v;             // T(:,:,i) or its transpose
            }
        double x[3];
        null3(M, x);
        nullv[3 * lane + 0] = x[0]; nullv[3 * lane + 1] = x[1]; nullv[3 * lane + 2] = x[2];
    }
    wave_sync();
    if (lane < 2) {
        Mat3 M;                                                              // [v1 v2 v3].'
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) M.m[i][k] = nullv[9 * lane + 3 * i + k];
        double x[3];
        null3(M, x);
        if (fix_sign) { const double sg = sgn(x[2]); x[0] *= sg; x[1] *= sg; x[2] *= sg; }
        double* dst = (lane == 0) ? (epi + 3) : epi;                         // lane 0: right nulls -> e31; lane 1: left -> e21
        dst[0] = x[0]; dst[1] = x[1]; dst[2] = x[2];
    }
    wave_sync();
}

// transform_TFT.m:42-49 with inverse = 1:  Tn(:,:,i) = inv(M2) (sum_j M1(j,i) To(:,:,j)) inv(M3).'
// followed by the Frobenius normalisation.  to/tn are 27-vectors in LDS; `mats`
// is 27 doubles of LDS scratch (M1, inv(M2), inv(M3) row-major) so that the
// lane-dependent indices address memory, not registers.
template <int G = 64, class MatFn>
__device__ inline void transform_tft_inverse(const double* to, double* tn, double* mats, MatFn matrix_of) {
    const int lane = Group<G>::lane();
    if (lane < 3) {
        Mat3 M = matrix_of(lane);
        if (lane > 0) M = mat3_inv(M);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) mats[9 * lane + 3 * r + c] = M.m[r][c];
    }
    wave_sync();
    double val = 0.0;
    if (lane < 27) {
        const int i = lane / 9, k = (lane % 9) / 3, j = lane % 3;           // entry T(j,k,i)
        const double m0 = mats[i], m1 = mats[3 + i], m2 = mats[6 + i];       // M1(:,i)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double mix = m0 * to[c + 3 * d] + m1 * to[c + 3 * d + 9] + m2 * to[c + 3 * d + 18];
                val += mats[9 + 3 * j + c] * mix * mats[18 + 3 * k + d];
            }
    }
    const double nn = Group<G>::sum(val * val);
    wave_sync();
    if (lane < 27) tn[lane] = val * rsqrt(nn);
    wave_sync();
}

// Wave-uniform operands of the per-correspondence loops (camera matrices, poses)
// are read from LDS once per pass and pinned to scalar registers: three 3x4
// cameras held in VGPRs would cost 72 registers and a wave per SIMD.
__device__ __forceinline__ void load_uniform12(const double* p, double (&u)[12]) {
#pragma unroll
    for (int c = 0; c < 12; ++c) u[c] = wave_uniform(p[c]);
}

// 2-view / 3-view DLT triangulation of one correspondence (triangulation3D.m:51-63):
// accumulate S = ls' * ls for the two rows [0 -1 y; 1 0 -x] * P of one view.
__device__ __forceinline__ void tri_accum(double (&S)[4][4], const double (&P)[12], double x, double y) {
    double r0[4], r1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { r0[c] = y * P[8 + c] - P[4 + c]; r1[c] = P[c] - x * P[8 + c]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) S[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
}
__device__ __forceinline__ void tri_zero(double (&S)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i][j] = 0.0;
}

// P = K [R | t]   (R row-major 9, t 3) -> row-major 3x4
__device__ __forceinline__ void compose_camera(const Mat3& K, const double* R, const double* t, double* P) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) P[4 * r + c] = K.m[r][0] * R[c] + K.m[r][1] * R[3 + c] + K.m[r][2] * R[6 + c];
        P[4 * r + 3] = K.m[r][0] * t[0] + K.m[r][1] * t[1] + K.m[r][2] * t[2];
    }
}

// P = K_v [R | t] from a row-major 3x4 pose held in LDS
__device__ __forceinline__ void compose_camera_from_pose(const Mat3& K, const double* Rt, double* P) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) P[4 * r + c] = K.m[r][0] * Rt[c] + K.m[r][1] * Rt[4 + c] + K.m[r][2] * Rt[8 + c];
}

// Cheirality vote of one pose candidate (R_t_from_TFT.m:96-101): sum_n sign(X1(3)) + sign(X2(3)), X1 the two-view DLT point of
// correspondence n (cameras Pfin[0] = K1 [I|0] and camB), X2 = [R t] X1.  Only the two SIGNS are consumed, so this is a slimmed copy
// of tri_pass: the first camera's fourth column is zero by construction; the point is the least-squares solution with X(4) = 1,
//     X(1:3) = -inv(S(1:3,1:3)) S(1:3,4)
// read off the Cholesky factor of S = A'A (its last row is the forward solve, one 3 x 3 back substitution finishes it) -- the
// inhomogeneous form of the same DLT system, within lambda_4/lambda_3 (~1e-6 for correspondences consistent with the candidate)
// of the singular vector the reference takes, which cannot move the sign of a depth that is O(1) in these units.  No fourth pivot,
// no normalisation, no iteration: 60 % of the instructions of the general pass.  `Rt`: candidate pose, row-major 3x4.
__device__ __attribute__((noinline)) int tri_vote(PoseLds* w, const double* pts, int N, int view, const double* camB, const double* Rt) {
    const int lane = lane_id();
    double PA[9], PB[12], R3[4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) PA[3 * r + c] = wave_uniform(w->Pfin[0][4 * r + c]);
    load_uniform12(camB, PB);
#pragma unroll
    for (int c = 0; c < 4; ++c) R3[c] = wave_uniform(Rt[8 + c]);
    int score = 0;
    Pt6 pnext = load_pt(pts, (lane < N) ? lane : 0);
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = pnext;
        if (i + WAVE < N) pnext = load_pt(pts, i + WAVE);
        const double x1 = p.v[0], y1 = p.v[1], x2 = (view == 1) ? p.v[2] : p.v[4], y2 = (view == 1) ? p.v[3] : p.v[5];
        double a0[3], a1[3], b0[4], b1[4];                                  // rows [0 -1 y; 1 0 -x] * P   (triangulation3D.m:58-59)
#pragma unroll
        for (int c = 0; c < 3; ++c) { a0[c] = y1 * PA[6 + c] - PA[3 + c]; a1[c] = PA[c] - x1 * PA[6 + c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) { b0[c] = y2 * PB[8 + c] - PB[4 + c]; b1[c] = PB[c] - x2 * PB[8 + c]; }
        double S[4][3];                                                      // lower triangle of A'A, columns 0..2 (S[3][3] is not needed)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c > r) continue;
                double v = b0[r] * b0[c] + b1[r] * b1[c];
                if (r < 3) v += a0[r] * a0[c] + a1[r] * a1[c];
                S[r][c] = v;
            }
        // Cholesky of S + delta I as in spd_min_eigvec (the shift only conditions the factorisation)
        const double tr = S[0][0] + S[1][1] + S[2][2] + (b0[3] * b0[3] + b1[3] * b1[3]);
        const double delta = 1e-14 * tr, pfloor = 1e-3 * delta + 1e-300;
        double L[4][3], inv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double d = S[j][j] + delta;
#pragma unroll
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            d = (d > pfloor) ? d : pfloor;
            inv[j] = rsqrt(d);
#pragma unroll
            for (int r = j + 1; r < 4; ++r) {
                double sv = S[r][j];
#pragma unroll
                for (int k = 0; k < j; ++k) sv -= L[r][k] * L[j][k];
                L[r][j] = sv * inv[j];
            }
        }
        double z[3];                                                         // L11' z = -L(4,1:3)'
#pragma unroll
        for (int r = 2; r >= 0; --r) {
            double sum = -L[3][r];
#pragma unroll
            for (int k = r + 1; k < 3; ++k) sum -= L[k][r] * z[k];
            z[r] = sum * inv[r];
        }
        const double z2 = R3[0] * z[0] + R3[1] * z[1] + R3[2] * z[2] + R3[3];
        score += (int)sgn(z[2]) + (int)sgn(z2);                             // X(4) = 1 > 0
    }
    return wave_sum_i(score);
}

// One pass over the correspondences of the triplet, one lane per correspondence:
// DLT triangulation (triangulation3D.m:51-63) from camera Pfin[0] = K1 [I|0], camera
// `camB` and (mode TRI_RECONST) camera `aux`, then a mode-specific epilogue.
// A single non-inlined copy serves the cheirality vote, the t3 scale and Reconst
// (code size: see the instruction-cache note in wave_eig.h).
//   TRI_SCALE   : aux = [K3*R3 | u3 = K3*t3]; num/den of R_t_from_TFT.m:72-73 -> w->pa[0..1]
//   TRI_RECONST : aux = third camera; dehomogenised points -> `out` (3 x N column-major)
//   TRI_REPROJECT : aux = third camera; the three reprojections of the homogeneous point ->
//                 `out` (6 x N), the initial observations of the Gauss-Helmert methods
//                 (ResslTFTPoseEstimation.m:72-75)
//   TRI_REPROJECT2 : two views only (Pfin[0], camB; `view` picks the second view's coordinates); the two
//                 reprojections -> `out` (4 x N), the initial observations of optimF (optimF.m:56-60)
// `pre` (9 doubles or null): affine map applied to the raw correspondences first (normalised points).
constexpr int TRI_VOTE = 0, TRI_SCALE = 1, TRI_RECONST = 2, TRI_REPROJECT = 3, TRI_REPROJECT2 = 4;
__device__ __attribute__((noinline)) int tri_pass(PoseLds* w, const double* pts, int N, int mode, int view, const double* camB,
                                                   const double* aux, double* out, const double* pre = nullptr) {
    const int lane = lane_id();
    double PA[12], PB[12], AX[12];
    load_uniform12(w->Pfin[0], PA);
    load_uniform12(camB, PB);
    load_uniform12(aux, AX);
    int score = 0;
    double num = 0.0, den = 0.0;
    Pt6 pnext = load_pt(pts, (lane < N) ? lane : 0);                             // software-pipelined: next point in flight
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = premap(pnext, pre);
        if (i + WAVE < N) pnext = load_pt(pts, i + WAVE);
        double S[4][4];
        tri_zero(S);
        tri_accum(S, PA, p.v[0], p.v[1]);
        tri_accum(S, PB, (view == 1) ? p.v[2] : p.v[4], (view == 1) ? p.v[3] : p.v[5]);
        if (mode == TRI_RECONST || mode == TRI_REPROJECT) tri_accum(S, AX, p.v[4], p.v[5]);
        double X[4];
        spd_min_eigvec<4>(S, X, 40);
        if (mode == TRI_REPROJECT2) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const double (&P)[12] = (v == 0) ? PA : PB;
                const double a = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
                const double b = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
                const double c = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
                out[4 * (long)i + 2 * v] = a / c;
                out[4 * (long)i + 2 * v + 1] = b / c;
            }
            continue;
        }
        if (mode == TRI_REPROJECT) {                                               // p_est = P*X; p(1:2)./p(3)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double (&P)[12] = (v == 0) ? PA : ((v == 1) ? PB : AX);
                const double a = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
                const double b = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
                const double c = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
                out[6 * (long)i + 2 * v] = a / c;
                out[6 * (long)i + 2 * v + 1] = b / c;
            }
            continue;
        }
        const double iw = 1.0 / X[3];
        const double X0 = X[0] * iw, X1 = X[1] * iw, X2 = X[2] * iw;                 // X./X(4)
        if (mode == TRI_SCALE) {
            double X3[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) X3[r] = AX[4 * r] * X0 + AX[4 * r + 1] * X1 + AX[4 * r + 2] * X2;   // X3 = K3*R3*X  (:71)
            const double u3[3] = {AX[3], AX[7], AX[11]};
            const double p3[3] = {p.v[4], p.v[5], 1.0};
            double c1[3], c2[3];
            cross3(p3, X3, c1);
            cross3(p3, u3, c2);
            num += c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
            den += c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
        } else {
            out[3 * (long)i + 0] = X0;
            out[3 * (long)i + 1] = X1;
            out[3 * (long)i + 2] = X2;
        }
    }
    if (mode == TRI_SCALE) {
        num = wave_sum(num);
        den = wave_sum(den);
        wave_sync();
        if (lane == 0) { w->pa[0] = num; w->pa[1] = den; }
        wave_sync();
    }
    return 0;
}

// recover_R_t (R_t_from_TFT.m:82-106 == LinearFPoseEstimation.m:84-109):
// decompose the two essential matrices (lanes 0,1), then vote.
// E[call] row-major in LDS scratch `Ein` (18 doubles).  Results: w->Rt[call].
// Candidate scores obey score(R,-t) = -score(R,t) exactly (the DLT system of the
// mirrored camera is the original with its 4th column negated), so only two of
// the four triangulation passes are evaluated; the selection loop below replays
// the reference's order and its `>=` rule on all four scores.
template <int G>
__device__ inline void recover_prepare(PoseLds* w, const double* Ein) {
    const int lane = Group<G>::lane();
    if (lane < 2) {
        Mat3 E, U, V;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) E.m[r][c] = Ein[9 * lane + 3 * r + c];
        double sv[3];
        svd3(E, U, V, sv);
        // U*W and U*W' with W = [0 -1 0; 1 0 0; 0 0 1]   (:84-86)
        Mat3 UW, UWt, Vt = mat3_T(V);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            UW.m[r][0] = U.m[r][1];  UW.m[r][1] = -U.m[r][0]; UW.m[r][2] = U.m[r][2];
            UWt.m[r][0] = -U.m[r][1]; UWt.m[r][1] = U.m[r][0]; UWt.m[r][2] = U.m[r][2];
        }
        Mat3 R = mat3_mul(UW, Vt), Rp = mat3_mul(UWt, Vt);
        const double sR = sgn(mat3_det(R)), sRp = sgn(mat3_det(Rp));        // :87
        double* c = w->cand[lane];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { c[3 * r + cc] = R.m[r][cc] * sR; c[9 + 3 * r + cc] = Rp.m[r][cc] * sRp; }
        c[18] = U.m[0][2]; c[19] = U.m[1][2]; c[20] = U.m[2][2];            // t = U(:,3)   (:88)
    }
    wave_sync();
    if (lane < 4) {                                                          // cameras of (R,t) and (Rp,t) for both calls
        const int call = lane >> 1, cd = lane & 1;
        const Mat3 K = load_K(w->calm, call + 1);
        compose_camera(K, w->cand[call] + 9 * cd, w->cand[call] + 18, w->P[lane]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w->candRt[lane][4 * r + c] = w->cand[call][9 * cd + 3 * r + c];
            w->candRt[lane][4 * r + 3] = w->cand[call][18 + r];
        }
    }
    if (lane == 4) {                                                         // P1 = K1 [I | 0]
        const Mat3 K1 = load_K(w->calm, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) { w->Pfin[0][4 * r] = K1.m[r][0]; w->Pfin[0][4 * r + 1] = K1.m[r][1]; w->Pfin[0][4 * r + 2] = K1.m[r][2]; w->Pfin[0][4 * r + 3] = 0.0; }
    }
    wave_sync();
}

__device__ inline int recover_vote(PoseLds* w, const double* pts, int N, double* dbg) {
    const int lane = lane_id();
    phase_stamp(dbg, 10);
    int status = ST_OK;
#pragma unroll 1
    for (int call = 0; call < 2; ++call) {
        const int sR = tri_vote(w, pts, N, call + 1, w->P[2 * call], w->candRt[2 * call]);
        const int sRp = tri_vote(w, pts, N, call + 1, w->P[2 * call + 1], w->candRt[2 * call + 1]);
        // reference order: k=1 (R,t), k=2 (R,-t), k=3 (Rp,-t), k=4 (Rp,t)   (:92-104)
        const int score[4] = {sR, -sR, -sRp, sRp};
        int seen = 0, pick = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (score[k] >= seen) { pick = k; seen = score[k]; }
        if (pick < 0) status = ST_NO_POSE;
        if (dbg && lane == 0) { for (int k = 0; k < 4; ++k) dbg[60 + 4 * call + k] = (double)score[k]; }
        if (lane < 12) {
            const int r = lane >> 2, c = lane & 3;
            const double* R = w->cand[call] + ((pick >= 2) ? 9 : 0);
            const double tsign = (pick == 1 || pick == 2) ? -1.0 : 1.0;
            w->Rt[call][lane] = (c < 3) ? R[3 * r + c] : tsign * w->cand[call][18 + r];
        }
        wave_sync();
    }
    return status;
}

__device__ inline int recover_poses(PoseLds* w, const double* Ein, const double* pts, int N, double* dbg) {
    recover_prepare<64>(w, Ein);
    return recover_vote(w, pts, N, dbg);
}

// t3 scale, R_t_from_TFT.m:68-74 == LinearFPoseEstimation.m:64-70.  Scales w->Rt[1](:,4) in place.
__device__ inline void scale_t3(PoseLds* w, const double* pts, int N, double* dbg) {
    const int lane = lane_id();
    if (lane < 2)                                                            // Pfin[1] = K2 [R2|t2];  Pfin[2] = [K3*R3 | u3 = K3*t3]   (:68,:71)
        compose_camera_from_pose(load_K(w->calm, lane + 1), w->Rt[lane], w->Pfin[lane + 1]);
    wave_sync();
    tri_pass(w, pts, N, TRI_SCALE, 1, w->Pfin[1], w->Pfin[2], nullptr);      // X from views 1,2 (:69-70)
    const double lam = -w->pa[0] / w->pa[1];                                 // :72-73
    if (dbg && lane == 0) dbg[68] = lam;
    wave_sync();
    if (lane < 3) w->Rt[1][4 * lane + 3] *= lam;                             // :74
    wave_sync();
}

// Final reconstruction (LinearTFTPoseEstimation.m:59-60): 3-view DLT with the
// recovered poses, dehomogenised, written as 3 x N column-major.
__device__ inline void final_reconst(PoseLds* w, const double* pts, int N, double* __restrict__ out) {
    const int lane = lane_id();
    if (lane == 0) compose_camera_from_pose(load_K(w->calm, 2), w->Rt[1], w->Pfin[2]);
    wave_sync();
    tri_pass(w, pts, N, TRI_RECONST, 1, w->Pfin[1], w->Pfin[2], out);
}

// write the chosen poses (row-major in LDS) as MATLAB column-major 3x4 arrays
__device__ inline void write_poses(const PoseLds* w, double* __restrict__ Rt2, double* __restrict__ Rt3) {
    const int lane = lane_id();
    if (lane < 24) {
        const int which = lane / 12, e = lane % 12, c = e / 3, r = e % 3;
        double* dst = which ? Rt3 : Rt2;
        dst[e] = w->Rt[which][4 * r + c];
    }
}

}  // namespace tff
