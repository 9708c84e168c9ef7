// Wave-level stages shared by the pose kernels (one wavefront per triplet).
// LDS carve-up, point staging, Normalize2Ddata, epipoles, transform_TFT,
// recover_R_t with the cheirality vote, the t3 scale and the final 3-view
// triangulation.  File:line citations are into the reference tree.
#pragma once
#include "wave.h"
#include "small_la.h"
#include "wave_eig.h"
#include "row_eig.h"
#include "wave_qr.h"

namespace tff {

// ---- flags of the pose kernels -------------------------------------------
constexpr int FLAG_RECONST = 1;       // also produce Reconst (3-view triangulation of every correspondence)
constexpr int FLAG_JACOBI = 2;        // force the Jacobi eigen-solver for the Gram matrices
constexpr int FLAG_STAGE_LDS = 4;     // correspondences are staged once in LDS (else re-read through L2)
constexpr int FLAG_ONLY_RETRY = 8;    // fix-up pass: process only triplets whose status is ST_RETRY
constexpr int FLAG_DBG_FP_HANDOVER = 64; // FaugPapa block kernel, test hook (TFF_OPT_DEBUG_FP_HANDOVER): hand every third triplet back to the generic kernel as if its pseudo-inverse had failed
constexpr int FLAG_DBG_ADAPTIVE = 32; // rows kernel, debug entry points only: keep the adaptive cheirality votes (the default under debug is all four scores)
constexpr int FLAG_XI_IN_LDS = 128;   // workgroup Gauss-Helmert kernels with the per-correspondence state in global slices: xi (6 N) stays in LDS after all, only W+ goes out
constexpr int FLAG_GH_EXACT = 16;     // Gauss-Helmert: always take the eigen-decomposition path for pinv(W) (A/B against the Cholesky path)

// ---- status codes (per triplet), mirroring the reference's failure modes --
constexpr int ST_OK = 0;
constexpr int ST_TOO_FEW = 1;         // N < 7 (TFT) / N < 8 (F): experiments.m:99, linearF.m:35
constexpr int ST_NONFINITE = 2;       // NaN/Inf reached the outputs (Gauss_Helmert.m:53,63)
constexpr int ST_NO_POSE = 3;         // no candidate with non-negative cheirality score (R_t_from_TFT.m:91-104)
constexpr int ST_RETRY = 100;         // internal: a fast tier could not finish or certify its result; the triplet is redone by the exact kernel

constexpr int DBG_STRIDE = 128;       // doubles per triplet in the optional debug buffer

// A triplet whose outputs hold a NaN / Inf anywhere reports it in its status and gets NaN in EVERY output (T, R_t_2, R_t_3, Reconst): the
// one-triplet kernels call this after their stores (whole wavefront; the fence puts the rewrite behind the stores other lanes made).
__device__ __forceinline__ void wave_nan_outputs(double* Rt2, double* Rt3, double* T, double* reconst, const long b, const int N) {
    const int lane = (int)(threadIdx.x & 63u);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    store_fence();
    if (lane < 12) { Rt2[b * 12 + lane] = qnan; Rt3[b * 12 + lane] = qnan; }
    if (lane < 27) T[b * 27 + lane] = qnan;
    if (reconst) for (int i = lane; i < 3 * N; i += 64) reconst[b * 3 * (long)N + i] = qnan;
}

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
// extra workspace of the exact kernel variant: R of the 4N x 27 system (kept for the second solve of linearTFT.m:84; the rotations
// of a Hestenes fall-back go to PoseLds::Lp).  One 27 x 27 array, not two: 16 KB per wavefront with PoseLds -- eight wavefronts per CU
struct JacobiLds {
    double A[27 * 27];
};
// the same for the fundamental-matrix kernels (9 x 9): 1.3 KB instead of 11.7 KB per wavefront
struct JacobiLdsF {
    double A[9 * 9];
    double V[9 * 9];
};
constexpr int POSE_LDS_DOUBLES = (int)(sizeof(PoseLds) / sizeof(double));
constexpr int JACOBI_LDS_DOUBLES = (int)(sizeof(JacobiLds) / sizeof(double));
constexpr int JACOBI_F_LDS_DOUBLES = (int)(sizeof(JacobiLdsF) / sizeof(double));

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

// Minimal-sample hypotheses (config 4): gather N correspondences of one shared scene by index.  Returns true (wave-uniform)
// when an index falls outside [0, Ns): that correspondence is not read (zeros) and the caller reports ST_TOO_FEW.
__device__ inline bool gather_points(const double* __restrict__ scene, const int* __restrict__ idx, double* dst, int N, int Ns) {
    const int lane = lane_id();
    bool bad = false;
    for (int e = lane; e < 3 * N; e += WAVE) {
        const int i = e / 3, part = e % 3;
        const int k = idx[i];
        const bool ok = k >= 0 && k < Ns;
        bad = bad || !ok;
        double2 v; v.x = 0.0; v.y = 0.0;
        if (ok) v = reinterpret_cast<const double2*>(scene + 6 * (long)k)[part];
        reinterpret_cast<double2*>(dst)[e] = v;
    }
    wave_sync();
    return wave_any(bad);
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
// Returns (per lane) false when a null-vector iteration hit its cap and EXACT = false left it unfinished.
template <int G, bool EXACT = true>
__device__ __attribute__((noinline)) bool epipoles_from_tensor(const double* t, double* nullv, double* epi, bool fix_sign) {
    const int lane = Group<G>::lane();
    bool ok = true;
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
        ok = null3<EXACT, true>(M, x);
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
        ok = null3<EXACT, true>(M, x) && ok;
        if (fix_sign) { const double sg = sgn(x[2]); x[0] *= sg; x[1] *= sg; x[2] *= sg; }
        double* dst = (lane == 0) ? (epi + 3) : epi;                         // lane 0: right nulls -> e31; lane 1: left -> e21
        dst[0] = x[0]; dst[1] = x[1]; dst[2] = x[2];
    }
    wave_sync();
    return ok;
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

// Rows [0 -1 y; 1 0 -x] * P of one view's DLT block (triangulation3D.m:58-59), P row-major 3x4.
__device__ __forceinline__ void dlt_rows(double* r0, double* r1, const double (&P)[12], double x, double y) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { r0[c] = y * P[8 + c] - P[4 + c]; r1[c] = P[c] - x * P[8 + c]; }
}

// Exact tier of the DLT point (see dlt_point): one-sided Jacobi (Hestenes) on the 2M x 4 matrix itself -- the reference's
// [~,~,V] = svd(ls_matrix); V(:,4) at SVD accuracy, whatever the gap between the two smallest singular values.
// Out of line and deliberately ROLLED (the matrices live in per-lane scratch memory, indexed dynamically): it runs for the rare
// correspondences whose inverse iteration hits its cap, and a caller's register budget is the maximum over its callees --
// unrolled into 40 live doubles it would cost every kernel that can reach it its occupancy.
// cam*: row-major 3x4 cameras in LDS (camC ignored unless three).
struct Vec4 { double v[4]; };
__device__ __attribute__((noinline)) Vec4 dlt_point_exact(const double* camA, const double* camB, const double* camC, const int three,
                                                          const double xa, const double ya, const double xb, const double yb,
                                                          const double xc, const double yc) {
    double M[6][4], V[4][4];
    const int rows = three ? 6 : 4;
#pragma unroll 1
    for (int v = 0; v < 3; ++v) {
        const double* P = (v == 0) ? camA : ((v == 1) ? camB : camC);
        const double x = (v == 0) ? xa : ((v == 1) ? xb : xc), y = (v == 0) ? ya : ((v == 1) ? yb : yc);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            const bool have = v < 2 || three;
            M[2 * v][c] = have ? y * P[8 + c] - P[4 + c] : 0.0;              // [0 -1 y; 1 0 -x] * P   (triangulation3D.m:58-59)
            M[2 * v + 1][c] = have ? P[c] - x * P[8 + c] : 0.0;
        }
    }
#pragma unroll 1
    for (int i = 0; i < 4; ++i)
#pragma unroll 1
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll 1
        for (int p = 0; p < 3; ++p)
#pragma unroll 1
            for (int q = p + 1; q < 4; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll 1
                for (int r = 0; r < rows; ++r) { const double mp = M[r][p], mq = M[r][q]; al += mp * mp; be += mq * mq; ga += mp * mq; }
                if (!(fabs(ga) > 1e-15 * sqrt(al * be))) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll 1
                for (int r = 0; r < rows; ++r) { const double mp = M[r][p], mq = M[r][q]; M[r][p] = c * mp - s * mq; M[r][q] = s * mp + c * mq; }
#pragma unroll 1
                for (int r = 0; r < 4; ++r) { const double vp = V[r][p], vq = V[r][q]; V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq; }
            }
        if (!rotated) break;
    }
    int best = 0;
    double bv = 0.0;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        double nn = 0.0;
#pragma unroll 1
        for (int r = 0; r < rows; ++r) nn += M[r][c] * M[r][c];
        if (c == 0 || nn < bv) { bv = nn; best = c; }
    }
    Vec4 X;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) X.v[r] = V[r][best];
    return X;
}

// Steps B - D of spd_min_eigvec_cert for the DLT system (see small_la.h), from the iterate `start` of its step A.  Out of line
// like dlt_point_exact (a caller's register budget is the maximum over its callees); the rows of the system are re-derived from
// the cameras in LDS wherever they are needed instead of being kept.  *ok = 0: not certified, the caller goes on to dlt_point_exact.
__device__ __attribute__((noinline)) Vec4 dlt_point_cert(const double* camA, const double* camB, const double* camC, const int three,
                                                         const double xa, const double ya, const double xb, const double yb,
                                                         const double xc, const double yc, const Vec4 start, int* ok) {
    const int views = three ? 3 : 2;
    auto rows = [&](const int v, double (&r0)[4], double (&r1)[4]) {
        const double* P = (v == 0) ? camA : ((v == 1) ? camB : camC);
        const double x = (v == 0) ? xa : ((v == 1) ? xb : xc), y = (v == 0) ? ya : ((v == 1) ? yb : yc);
#pragma unroll
        for (int c = 0; c < 4; ++c) { r0[c] = y * P[8 + c] - P[4 + c]; r1[c] = P[c] - x * P[8 + c]; }
    };
    double S[4][4];
    tri_zero(S);
#pragma unroll 1
    for (int v = 0; v < views; ++v) {
        double r0[4], r1[4];
        rows(v, r0, r1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) S[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
    }
    double x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = start.v[k];
    const bool fine = spd_min_eigvec_cert_tail<4>(S, x, [&](const double (&u)[4], double (&out)[4], double& rho) {
        rho = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = 0.0;
#pragma unroll 1
        for (int v = 0; v < views; ++v) {
            double r0[4], r1[4];
            rows(v, r0, r1);
            const double m0 = r0[0] * u[0] + r0[1] * u[1] + r0[2] * u[2] + r0[3] * u[3];
            const double m1 = r1[0] * u[0] + r1[1] * u[1] + r1[2] * u[2] + r1[3] * u[3];
            rho += m0 * m0 + m1 * m1;
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] += r0[c] * m0 + r1[c] * m1;
        }
    });
    *ok = fine ? 1 : 0;
    Vec4 X;
#pragma unroll
    for (int k = 0; k < 4; ++k) X.v[k] = x[k];
    return X;
}

// The DLT point of one correspondence (triangulation3D.m:51-63): V(:,4) of the 2M x 4 system of cameras PA, PB and (three) PC,
// unit norm, sign free.  Fast tier: Cholesky + inverse iteration on the 4 x 4 normal matrix, run until the iterate stops
// moving.  When the two smallest singular values nearly coincide (inconsistent systems: minimal samples, wrong candidates) the
// iteration cap is hit; then
//   EXACT = true : the one-sided Jacobi on the matrix itself takes over (gap-independent, SVD accuracy), out of line;
//   EXACT = false: the function only reports it (returns false) and the CALLER repeats its pass with EXACT = true -- the hot
//                  per-correspondence loops then contain no call, which would make their (non-inlined) functions save and
//                  restore ~35 registers on every entry.
// PA.. hold the cameras as (wave-uniform) values, camA.. point to the same cameras in LDS (for the exact tier).
template <bool EXACT, bool CERT = false>
__device__ __forceinline__ bool dlt_point_solve(const double (&S)[4][4], const double* camA, const double* camB, const double* camC, const bool three,
                                                const double xa, const double ya, const double xb, const double yb, const double xc, const double yc,
                                                double (&X)[4]);
template <bool EXACT, bool CERT = false>
__device__ __forceinline__ bool dlt_point(const double (&PA)[12], const double (&PB)[12], const double (&PC)[12],
                                          const double* camA, const double* camB, const double* camC, const bool three,
                                          const double xa, const double ya, const double xb, const double yb, const double xc, const double yc,
                                          double (&X)[4]) {
    double S[4][4];
    tri_zero(S);
    tri_accum(S, PA, xa, ya);
    tri_accum(S, PB, xb, yb);
    if (three) tri_accum(S, PC, xc, yc);
    if constexpr (EXACT && CERT) {
        return dlt_point_solve<true, true>(S, camA, camB, camC, three, xa, ya, xb, yb, xc, yc, X);
    } else {
        bool conv;
        spd_min_eigvec<4>(S, X, 40, &conv);
        if (EXACT && !conv) {
            Vec4 E; E = dlt_point_exact(camA, camB, camC, three ? 1 : 0, xa, ya, xb, yb, xc, yc);
#pragma unroll
            for (int k = 0; k < 4; ++k) X[k] = E.v[k];
            conv = true;
        }
        return conv;
    }
}
// ... from the normal matrix S = M'M of the system (lower triangle), for callers that need S themselves.
// CERT (with EXACT): the gap-independent certified tier of small_la.h between the three-iteration fast tier and the one-sided Jacobi
// -- for the kernels that triangulate many correspondences of unknown consistency (k_repr_error, k_triangulate) and, since round 4, for
// the exact passes over a triplet's own correspondences (tri_vote_exact, tri_pass_exact): on minimal samples a tenth of those triangulations
// used to fall through to the one-sided Jacobi, whose rolled copy lives in scratch memory -- 57 GB of scratch traffic per million eight-point
// hypotheses (profiles/r3_config4f_summary.json: 129 x the algorithmic bytes).  The certified tier settles all but ~7e-4 of them in registers:
// eight-point hypotheses 33.9 -> 29.3 ms per million, seven-point 60.7 -> 58.4 ms, identical inlier counts.  (The fast kernels never reach these
// functions -- EXACT = false only reports -- so their register budgets are untouched; the exact kernels stay at two wavefronts per SIMD.)
template <bool EXACT, bool CERT>
__device__ __forceinline__ bool dlt_point_solve(const double (&S)[4][4], const double* camA, const double* camB, const double* camC, const bool three,
                                                const double xa, const double ya, const double xb, const double yb, const double xc, const double yc,
                                                double (&X)[4]) {
    bool conv;
    spd_min_eigvec<4>(S, X, (EXACT && CERT) ? opaque_int(3) : 40, &conv);
    if constexpr (EXACT && CERT) {
        if (!conv) {                                                         // gap-independent tier (small_la.h), out of line
            Vec4 E;
#pragma unroll
            for (int k = 0; k < 4; ++k) E.v[k] = X[k];
            int ok = 0;
            E = dlt_point_cert(camA, camB, camC, three ? 1 : 0, xa, ya, xb, yb, xc, yc, E, &ok);
#pragma unroll
            for (int k = 0; k < 4; ++k) X[k] = E.v[k];
            conv = ok != 0;
        }
    }
    if (EXACT && !conv) {
        Vec4 E; E = dlt_point_exact(camA, camB, camC, three ? 1 : 0, xa, ya, xb, yb, xc, yc);
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = E.v[k];
        conv = true;
    }
    return conv;
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

// Cheirality vote of one pose candidate (R_t_from_TFT.m:96-101): sum_n sign(X1(3)) + sign(X2(3)), X1 = X / X(4) with X the
// homogeneous two-view DLT point of correspondence n (cameras Pfin[0] = K1 [I|0] and camB; V(:,4) of the 4 x 4 system,
// triangulation3D.m:61-62), X2 = [R t] X1.  Only the two SIGNS are consumed, so the vote is two-tiered:
//   * fast tier (tri_vote_fast): the inhomogeneous least-squares point z = -inv(S11) s (S = M'M = [S11 s; s' sigma]; X(4) = 1),
//     read off the Cholesky factor with one 3 x 3 back substitution -- no fourth column, no normalisation, no iteration;
//   * certificate: the singular vector satisfies X(1:3)/X(4) = -inv(S11 - mu I) s with mu = lambda_min(S) <= nu = lambda_min(S11)
//     (interlacing), a continuous path from z (mu' = 0) along which every eigen-component of z grows by at most
//     mu / (nu - mu); so |X(1:3)/X(4) - z| <= rho |z|, rho = m / (nu_low - m), with the computable bounds
//     mu <= m = (Schur complement of S11) / (1 + |z|^2)  (Rayleigh quotient of (z,1)) and nu >= nu_low = 4 det(S11) / tr(S11)^2.
//     Both depths are then sign-certain iff they exceed (2 rho + rounding) |z|  (the pose row R(3,:) has unit norm);
//   * exact tier (tri_vote_exact), run for the whole candidate when ANY of its correspondences is not certified (points near
//     infinity or near a camera plane, inconsistent systems of minimal samples): the converged homogeneous solution --
//     inverse iteration on S, one-sided Jacobi on the 4 x 4 matrix itself when its two smallest singular values nearly coincide.
// The scores therefore equal the reference's for every candidate; on well-posed triplets only the fast tier runs (+8 % on it
// for the certificate).  `Rt`: candidate pose, row-major 3x4.  tri_vote_fast returns 2 * score + (1 if any correspondence
// was not certified, in which case the score is not valid).
__device__ __attribute__((noinline)) int tri_vote_fast(PoseLds* w, const double* pts, int N, int view, const double* camB, const double* Rt) {
    const int lane = lane_id();
    double PA[12], PB[12], R3[4];
    load_uniform12(w->Pfin[0], PA);                                          // PA[3] = PA[7] = PA[11] = 0
    load_uniform12(camB, PB);
#pragma unroll
    for (int c = 0; c < 4; ++c) R3[c] = wave_uniform(Rt[8 + c]);
    int score = 0;
    bool all_certain = true;
    Pt6 pnext = load_pt(pts, (lane < N) ? lane : 0);
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = pnext;
        if (i + WAVE < N) pnext = load_pt(pts, i + WAVE);
        const double x1 = p.v[0], y1 = p.v[1], x2 = (view == 1) ? p.v[2] : p.v[4], y2 = (view == 1) ? p.v[3] : p.v[5];
        double a0[3], a1[3], b0[4], b1[4];                                  // rows [0 -1 y; 1 0 -x] * P   (triangulation3D.m:58-59)
#pragma unroll
        for (int c = 0; c < 3; ++c) { a0[c] = y1 * PA[8 + c] - PA[4 + c]; a1[c] = PA[c] - x1 * PA[8 + c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) { b0[c] = y2 * PB[8 + c] - PB[4 + c]; b1[c] = PB[c] - x2 * PB[8 + c]; }
        double S[4][3];                                                      // lower triangle of A'A, columns 0..2
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
        const double S33 = b0[3] * b0[3] + b1[3] * b1[3];
        const double tr3 = S[0][0] + S[1][1] + S[2][2];
        const double delta = 1e-14 * (tr3 + S33), pfloor = 1e-3 * delta + 1e-300;
        double L[4][3], inv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double d = S[j][j] + delta;
#pragma unroll
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
            inv[j] = rsqrt_pos(d);
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
        const double d1 = z[2], d2 = R3[0] * z[0] + R3[1] * z[1] + R3[2] * z[2] + R3[3];   // depths with X(4) = 1 > 0
        // certificate (division-free form of  min|d| > (2 rho + 1e-13 cond(S11)) |z|, everything scaled by tr^2 / det(S11))
        const double d4 = S33 + delta - (L[3][0] * L[3][0] + L[3][1] * L[3][1] + L[3][2] * L[3][2]);   // Schur complement (>= 0)
        const double zz = z[0] * z[0] + z[1] * z[1] + z[2] * z[2];
        const double idet = (inv[0] * inv[1]) * (inv[0] * inv[1]) * (inv[2] * inv[2]);                 // 1 / det(S11 + delta I)
        const double trs = tr3 + 3.0 * delta;
        const double e = trs * trs * idet;                                   // 4 / nu_low
        const double Gp = 4.0 * (1.0 + zz) - d4 * e;                         // (nu_low - m), scaled; > 0 required
        const double rhs = 2.0 * d4 * e + 2.5e-14 * trs * e * Gp;
        const double dmin2 = fmin(d1 * d1, d2 * d2);
        const bool certain = Gp > 0.0 && dmin2 * Gp * Gp > rhs * rhs * zz;  // false for NaN / inf
        all_certain = all_certain && certain;
        score += (int)sgn(d1) + (int)sgn(d2);
    }
    return 2 * wave_sum_i(score) + (wave_any(!all_certain) ? 1 : 0);
}
// The fast tier for the TWO rotation candidates of one essential matrix in one pass over the correspondences (same second view, same
// camera K1 [I|0]): the point is loaded once and camera 1's two rows and their share of S11 are formed once -- a quarter of the per-point
// instructions of the second candidate.  Same arithmetic per candidate as tri_vote_fast.  Returns tri_vote_fast's value for candidate 0 in
// the low and for candidate 1 in the high 16 bits (|score| <= 2 N <= 2^14 is the caller's business: it falls back to two single passes beyond).
struct VoteCam { double PB[12], R3[4]; };
// what vote_one knows about the 4 x 4 system when it is done: the Cholesky factor of S + delta I short of its last pivot (d4, unfloored) and the
// inhomogeneous point z -- exactly the state of spd_min_eigvec<4> before its first iteration (dlt_from_vote goes on from there)
struct VoteFactor { double L[4][3], inv[3], d4, z[3], pfloor; };
// CERT = false (with KEEP): factor and z only -- the caller takes the two signs from the converged point dlt_from_vote gives it anyway (the exact
// tier of the vote by definition), so the certificate of the fast tier is not needed; score / all_certain are left alone.
template <bool KEEP = false, bool CERT = true>
__device__ __forceinline__ void vote_one(const double (&SA)[6], const VoteCam& cam, const double x2, const double y2, int& score, bool& all_certain,
                                         VoteFactor* keep = nullptr) {
    double b0[4], b1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { b0[c] = y2 * cam.PB[8 + c] - cam.PB[4 + c]; b1[c] = cam.PB[c] - x2 * cam.PB[8 + c]; }
    double S[4][3];                                                          // lower triangle of A'A, columns 0..2
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c > r) continue;
            double v = b0[r] * b0[c] + b1[r] * b1[c];
            if (r < 3) v += SA[r * (r + 1) / 2 + c];
            S[r][c] = v;
        }
    const double S33 = b0[3] * b0[3] + b1[3] * b1[3];
    const double tr3 = S[0][0] + S[1][1] + S[2][2];
    const double delta = 1e-14 * (tr3 + S33), pfloor = 1e-3 * delta + 1e-300;
    double L[4][3], inv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double d = S[j][j] + delta;
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
        inv[j] = rsqrt_pos(d);
#pragma unroll
        for (int r = j + 1; r < 4; ++r) {
            double sv = S[r][j];
#pragma unroll
            for (int k = 0; k < j; ++k) sv -= L[r][k] * L[j][k];
            L[r][j] = sv * inv[j];
        }
    }
    double z[3];
#pragma unroll
    for (int r = 2; r >= 0; --r) {
        double sum = -L[3][r];
#pragma unroll
        for (int k = r + 1; k < 3; ++k) sum -= L[k][r] * z[k];
        z[r] = sum * inv[r];
    }
    const double d4 = S33 + delta - (L[3][0] * L[3][0] + L[3][1] * L[3][1] + L[3][2] * L[3][2]);
    if constexpr (CERT) {
        const double d1 = z[2], d2 = cam.R3[0] * z[0] + cam.R3[1] * z[1] + cam.R3[2] * z[2] + cam.R3[3];
        const double zz = z[0] * z[0] + z[1] * z[1] + z[2] * z[2];
        const double idet = (inv[0] * inv[1]) * (inv[0] * inv[1]) * (inv[2] * inv[2]);
        const double trs = tr3 + 3.0 * delta;
        const double e = trs * trs * idet;
        const double Gp = 4.0 * (1.0 + zz) - d4 * e;
        const double rhs = 2.0 * d4 * e + 2.5e-14 * trs * e * Gp;
        const double dmin2 = fmin(d1 * d1, d2 * d2);
        const bool certain = Gp > 0.0 && dmin2 * Gp * Gp > rhs * rhs * zz;  // false for NaN / inf
        all_certain = all_certain && certain;
        score += (int)sgn(d1) + (int)sgn(d2);
    }
    if constexpr (KEEP) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) keep->L[r][c] = (c < r) ? L[r][c] : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { keep->inv[c] = inv[c]; keep->z[c] = z[c]; }
        keep->d4 = d4;
        keep->pfloor = pfloor;
    }
}
// The homogeneous two-view DLT point (triangulation3D.m:61-62) of the system vote_one has just factored: what dlt_point's fast tier
// (spd_min_eigvec<4>) computes from scratch -- same factor, same start, same loop.  Returns false when the iteration hit its cap.
__device__ __forceinline__ bool dlt_from_vote(const VoteFactor& f, double (&X)[4]) {
    double L[4][4], inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) L[r][c] = (c < r) ? f.L[r][c] : 0.0;    // (the loop reads the strict lower triangle and inv[])
    inv[0] = f.inv[0]; inv[1] = f.inv[1]; inv[2] = f.inv[2];
    inv[3] = rsqrt_pos(fmax(f.d4, f.pfloor));
    const double nn0 = 1.0 + (f.z[0] * f.z[0] + f.z[1] * f.z[1] + f.z[2] * f.z[2]);
    const double r0 = rsqrt(nn0);
    const bool fin = nn0 <= 1e300;
    X[0] = fin ? f.z[0] * r0 : 0.5; X[1] = fin ? f.z[1] * r0 : 0.5; X[2] = fin ? f.z[2] * r0 : 0.5; X[3] = fin ? r0 : 0.5;
    bool conv;
    chol_invit<4>(L, inv, X, 40, &conv);
    return conv;
}
// CAM1_IN_REGISTERS: the second camera in vector registers (194 registers: for kernels that run two wavefronts per SIMD anyway) or re-read from
// LDS per trip (132: for the fundamental-matrix kernels, three wavefronts per SIMD at <= 168; 1 % slower inside the trifocal kernel).
template <bool CAM1_IN_REGISTERS>
__device__ __attribute__((noinline)) int tri_vote_fast2(PoseLds* w, const double* pts, int N, int view, const double* camB0, const double* Rt0,
                                                        const double* camB1, const double* Rt1) {
    const int lane = lane_id();
    double PA[12];
    VoteCam c0, c1r;
    load_uniform12(w->Pfin[0], PA);                                          // PA[3] = PA[7] = PA[11] = 0
    load_uniform12(camB0, c0.PB);
#pragma unroll
    for (int c = 0; c < 4; ++c) c0.R3[c] = wave_uniform(Rt0[8 + c]);
    if constexpr (CAM1_IN_REGISTERS) {                                       // (vector registers: three cameras do not fit the scalar file)
#pragma unroll
        for (int c = 0; c < 12; ++c) c1r.PB[c] = camB1[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) c1r.R3[c] = Rt1[8 + c];
    }
    int score0 = 0, score1 = 0;
    bool certain0 = true, certain1 = true;
    Pt6 pnext = load_pt(pts, (lane < N) ? lane : 0);
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = pnext;
        if (i + WAVE < N) pnext = load_pt(pts, i + WAVE);
        const double x1 = p.v[0], y1 = p.v[1], x2 = (view == 1) ? p.v[2] : p.v[4], y2 = (view == 1) ? p.v[3] : p.v[5];
        double a0[3], a1[3], SA[6];                                          // rows [0 -1 y; 1 0 -x] * P1 and their A'A   (triangulation3D.m:58-59)
#pragma unroll
        for (int c = 0; c < 3; ++c) { a0[c] = y1 * PA[8 + c] - PA[4 + c]; a1[c] = PA[c] - x1 * PA[8 + c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) SA[r * (r + 1) / 2 + c] = a0[r] * a0[c] + a1[r] * a1[c];
        vote_one(SA, c0, x2, y2, score0, certain0);
        if constexpr (CAM1_IN_REGISTERS) {
            vote_one(SA, c1r, x2, y2, score1, certain1);
        } else {
            VoteCam c1;                                                      // the second camera is re-read from LDS per trip (uniform address: one broadcast read each)
            const double* pb = camB1 + opaque_int(0);
            const double* pr = Rt1 + opaque_int(0);
#pragma unroll
            for (int c = 0; c < 12; ++c) c1.PB[c] = pb[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) c1.R3[c] = pr[8 + c];
            vote_one(SA, c1, x2, y2, score1, certain1);
        }
    }
    const int r0 = 2 * wave_sum_i(score0) + (wave_any(!certain0) ? 1 : 0), r1 = 2 * wave_sum_i(score1) + (wave_any(!certain1) ? 1 : 0);
    return (int)(((unsigned)r1 << 16) | ((unsigned)r0 & 0xffffu));
}
// exact tier: every correspondence from its converged homogeneous point  (R_t_from_TFT.m:98-99)
__device__ __attribute__((noinline)) int tri_vote_exact(PoseLds* w, const double* pts, int N, int view, const double* camB, const double* Rt) {
    const int lane = lane_id();
    typedef const double (&cam_ref)[12];                                     // cameras read from LDS where used: see tri_pass_impl
    cam_ref PA = *reinterpret_cast<const double(*)[12]>(w->Pfin[0]);
    cam_ref PB = *reinterpret_cast<const double(*)[12]>(camB);
    double R3[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) R3[c] = wave_uniform(Rt[8 + c]);
    int score = 0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = load_pt(pts, i);
        double X[4];
        dlt_point<true, true>(PA, PB, PB, w->Pfin[0], camB, camB, false, p.v[0], p.v[1], (view == 1) ? p.v[2] : p.v[4],
                        (view == 1) ? p.v[3] : p.v[5], 0.0, 0.0, X);
        const double s4 = sgn(X[3]);                                         // X1 = X ./ X(4)
        const double d1 = X[2] * s4, d2 = (R3[0] * X[0] + R3[1] * X[1] + R3[2] * X[2] + R3[3] * X[3]) * s4;
        score += (int)sgn(d1) + (int)sgn(d2);
    }
    return wave_sum_i(score);
}
// EXACT = true: the score (the exact tier runs when needed).  EXACT = false: fast tier only; *ok = false when the score is not
// certified (the caller hands the triplet to the exact kernel).
template <bool EXACT = true>
__device__ __forceinline__ int tri_vote(PoseLds* w, const double* pts, int N, int view, const double* camB, const double* Rt, bool* ok) {
    const int r = tri_vote_fast(w, pts, N, view, camB, Rt);                   // wave-uniform
    if (r & 1) {
        if (EXACT) return tri_vote_exact(w, pts, N, view, camB, Rt);
        *ok = false;
    }
    return r >> 1;
}

// One pass over the correspondences of the triplet, one lane per correspondence:
// DLT triangulation (triangulation3D.m:51-63) from camera Pfin[0] = K1 [I|0], camera
// `camB` and (mode TRI_RECONST) camera `aux`, then a mode-specific epilogue.
// A single non-inlined copy serves the t3 scale, Reconst and the initial observations
// (code size: see the instruction-cache note in wave_eig.h).
//   TRI_SCALE   : aux = [K3*R3 | u3 = K3*t3]; num/den of R_t_from_TFT.m:72-73 -> w->pa[0..1]
//   TRI_RECONST : aux = third camera; dehomogenised points -> `out` (3 x N column-major)
//   TRI_REPROJECT : aux = third camera; the three reprojections of the homogeneous point ->
//                 `out` (6 x N), the initial observations of the Gauss-Helmert methods
//                 (ResslTFTPoseEstimation.m:72-75)
//   TRI_REPROJECT2 : two views only (Pfin[0], camB; `view` picks the second view's coordinates); the two
//                 reprojections -> `out` (4 x N), the initial observations of optimF (optimF.m:56-60)
// `pre` (9 doubles or null): affine map applied to the raw correspondences first (normalised points).
// Two copies (see dlt_point): tri_pass_fast returns 1 when some correspondence's inverse iteration hit its cap, and
// tri_pass then repeats the pass with the exact tier enabled (rare: inconsistent systems of minimal samples).
constexpr int TRI_VOTE = 0, TRI_SCALE = 1, TRI_RECONST = 2, TRI_REPROJECT = 3, TRI_REPROJECT2 = 4;
template <bool EXACT>
__device__ __forceinline__ int tri_pass_impl(PoseLds* w, const double* pts, int N, int mode, int view, const double* camB,
                                             const double* aux, double* out, const double* pre) {
    const int lane = lane_id();
    // EXACT: the cameras are read from LDS where they are used instead of being held in registers, so that none of their 36
    // values stays live across the out-of-line tiers of dlt_point (a function's register budget is what its callers must keep free)
    double PAr[12], PBr[12], AXr[12];
    if (!EXACT) {
        load_uniform12(w->Pfin[0], PAr);
        load_uniform12(camB, PBr);
        load_uniform12(aux, AXr);
    }
    typedef const double (&cam_ref)[12];
    cam_ref PA = EXACT ? *reinterpret_cast<const double(*)[12]>(w->Pfin[0]) : PAr;
    cam_ref PB = EXACT ? *reinterpret_cast<const double(*)[12]>(camB) : PBr;
    cam_ref AX = EXACT ? *reinterpret_cast<const double(*)[12]>(aux) : AXr;
    bool all_conv = true;
    double num = 0.0, den = 0.0;
    Pt6 pnext = load_pt(pts, (lane < N) ? lane : 0);                             // software-pipelined: next point in flight
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 p = premap(pnext, pre);
        if (i + WAVE < N) pnext = load_pt(pts, i + WAVE);
        double X[4];
        const bool conv = dlt_point<EXACT, EXACT>(PA, PB, AX, w->Pfin[0], camB, aux, mode == TRI_RECONST || mode == TRI_REPROJECT, p.v[0], p.v[1],
                                           (view == 1) ? p.v[2] : p.v[4], (view == 1) ? p.v[3] : p.v[5], p.v[4], p.v[5], X);
        all_conv = all_conv && conv;
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
    return wave_any(!all_conv) ? 1 : 0;
}
__device__ __attribute__((noinline)) int tri_pass_fast(PoseLds* w, const double* pts, int N, int mode, int view, const double* camB,
                                                        const double* aux, double* out, const double* pre) {
    return tri_pass_impl<false>(w, pts, N, mode, view, camB, aux, out, pre);
}
__device__ __attribute__((noinline)) int tri_pass_exact(PoseLds* w, const double* pts, int N, int mode, int view, const double* camB,
                                                         const double* aux, double* out, const double* pre) {
    return tri_pass_impl<true>(w, pts, N, mode, view, camB, aux, out, pre);
}
// Returns true when every correspondence's point is converged (always, with EXACT = true).
template <bool EXACT = true>
__device__ __forceinline__ bool tri_pass(PoseLds* w, const double* pts, int N, int mode, int view, const double* camB,
                                         const double* aux, double* out, const double* pre = nullptr) {
    if (tri_pass_fast(w, pts, N, mode, view, camB, aux, out, pre)) {
        if (!EXACT) return false;
        tri_pass_exact(w, pts, N, mode, view, camB, aux, out, pre);
    }
    return true;
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

// ok (optional, EXACT = false): set to false when a score could not be certified by the fast tier.
// FUSE: both rotation candidates of a call in one pass (tri_vote_fast2) -- 1: second camera in registers, 2: re-read from LDS; 0: two single
// passes (A/B).  A callee's register count is its callers': the register form needs 194 and cost the fundamental-matrix kernels (three wavefronts
// per SIMD at <= 168) a wavefront -- LinearF 291 -> 314 us; the LDS form needs 132 and serves them (LinearF 34.3 -> 35.7 M/s).
template <bool EXACT = true, int FUSE = 1>
__device__ inline int recover_vote(PoseLds* w, const double* pts, int N, double* dbg, bool* ok = nullptr) {
    const int lane = lane_id();
    phase_stamp(dbg, 10);
    int status = ST_OK;
    bool certified = true;
#pragma unroll 1
    for (int call = 0; call < 2; ++call) {
        int sR, sRp;
        if (FUSE != 0 && N <= 4096) {                                             // both candidates in one pass (|2 score + 1| < 2^15)
            const int both = tri_vote_fast2<FUSE == 1>(w, pts, N, call + 1, w->P[2 * call], w->candRt[2 * call], w->P[2 * call + 1], w->candRt[2 * call + 1]);
            const int r0 = (int)(short)(both & 0xffff), r1 = both >> 16;     // wave-uniform
            sR = r0 >> 1; sRp = r1 >> 1;
            if (r0 & 1) { if (EXACT) sR = tri_vote_exact(w, pts, N, call + 1, w->P[2 * call], w->candRt[2 * call]); else certified = false; }
            if (r1 & 1) { if (EXACT) sRp = tri_vote_exact(w, pts, N, call + 1, w->P[2 * call + 1], w->candRt[2 * call + 1]); else certified = false; }
        } else {
            sR = tri_vote<EXACT>(w, pts, N, call + 1, w->P[2 * call], w->candRt[2 * call], &certified);
            sRp = tri_vote<EXACT>(w, pts, N, call + 1, w->P[2 * call + 1], w->candRt[2 * call + 1], &certified);
        }
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
    if (ok && !certified) *ok = false;
    return status;
}

template <bool EXACT = true, int FUSE = 1>
__device__ inline int recover_poses(PoseLds* w, const double* Ein, const double* pts, int N, double* dbg, bool* ok = nullptr) {
    recover_prepare<64>(w, Ein);
    return recover_vote<EXACT, FUSE>(w, pts, N, dbg, ok);
}

// t3 scale, R_t_from_TFT.m:68-74 == LinearFPoseEstimation.m:64-70.  Scales w->Rt[1](:,4) in place.
template <bool EXACT = true>
__device__ inline bool scale_t3(PoseLds* w, const double* pts, int N, double* dbg) {
    const int lane = lane_id();
    if (lane < 2)                                                            // Pfin[1] = K2 [R2|t2];  Pfin[2] = [K3*R3 | u3 = K3*t3]   (:68,:71)
        compose_camera_from_pose(load_K(w->calm, lane + 1), w->Rt[lane], w->Pfin[lane + 1]);
    wave_sync();
    const bool conv = tri_pass<EXACT>(w, pts, N, TRI_SCALE, 1, w->Pfin[1], w->Pfin[2], nullptr);      // X from views 1,2 (:69-70)
    const double lam = -w->pa[0] / w->pa[1];                                 // :72-73
    if (dbg && lane == 0) dbg[68] = lam;
    wave_sync();
    if (lane < 3) w->Rt[1][4 * lane + 3] *= lam;                             // :74
    wave_sync();
    return conv;
}

// Final reconstruction (LinearTFTPoseEstimation.m:59-60): 3-view DLT with the
// recovered poses, dehomogenised, written as 3 x N column-major.
template <bool EXACT = true>
__device__ inline bool final_reconst(PoseLds* w, const double* pts, int N, double* __restrict__ out) {
    const int lane = lane_id();
    if (lane == 0) compose_camera_from_pose(load_K(w->calm, 2), w->Rt[1], w->Pfin[2]);
    wave_sync();
    return tri_pass<EXACT>(w, pts, N, TRI_RECONST, 1, w->Pfin[1], w->Pfin[2], out);
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
