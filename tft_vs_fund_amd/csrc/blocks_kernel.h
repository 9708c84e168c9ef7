// Building-block kernels behind the C ABI: the reference's L1/L2 functions as
// batched device entry points, one wavefront per batch item, plus the minimal-sample /
// inlier-count pair of BASELINE.json's config 4.
//
//   k_triangulate     auxiliar_functions/triangulation3D.m:32-64
//   k_repr_error      auxiliar_functions/ReprError.m:39-65 (+ project3Dpoints.m:28-35), and the
//                     1-px inlier rule of experiments_real.m:94-98 (int32 count per item)
//   k_transform_tft   TFT_methods/transform_TFT.m:36-49 (both directions)
//   k_rt_from_tft     TFT_methods/R_t_from_TFT.m:40-106
//   k_linear_tft      TFT_methods/linearTFT.m:33-91 (points used as given: no normalisation)
#pragma once
#include "tft_kernel.h"

namespace tff {

// MATLAB 3x4 column-major -> row-major 12
__device__ __forceinline__ void load_camera_cm(const double* src, double* dst_rowmajor, int lane) {
    if (lane < 12) dst_rowmajor[4 * (lane % 3) + lane / 3] = src[lane];
}

// ---- triangulation3D ---------------------------------------------------------------------
struct TriangulateArgs {
    const double* cams;      // B x (M x 12) column-major 3x4 each, or M x 12 shared (cam_stride 0)
    long cam_stride;         // 12 M or 0
    const double* pts;       // B x (2M x N) column-major: point n = 2M contiguous doubles
    long B;
    int M, N;
    double* X;               // B x (4 x N): unit-norm homogeneous points (sign free), as triangulation3D returns
};
__global__ void __launch_bounds__(64, 4) k_triangulate(const TriangulateArgs a) {
    __shared__ double cam[3][12];
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        wave_sync();
        for (int v = 0; v < a.M; ++v) load_camera_cm(a.cams + b * a.cam_stride + 12 * v, cam[v], lane);
        wave_sync();
        double P0[12], P1[12], P2[12];
        load_uniform12(cam[0], P0);
        load_uniform12(cam[1], P1);
        load_uniform12(cam[(a.M > 2) ? 2 : 1], P2);
        const double* p = a.pts + b * 2 * (long)a.M * a.N;
        double* out = a.X + b * 4 * (long)a.N;
#pragma unroll 1
        for (int i = lane; i < a.N; i += WAVE) {
            const double* q = p + 2 * (long)a.M * i;
            double X[4];
            dlt_point<true, true>(P0, P1, P2, cam[0], cam[1], cam[2], a.M > 2, q[0], q[1], q[2], q[3], (a.M > 2) ? q[4] : 0.0, (a.M > 2) ? q[5] : 0.0, X);
#pragma unroll
            for (int k = 0; k < 4; ++k) out[4 * (long)i + k] = X[k];
        }
    }
}

// ---- ReprError / inlier count ---------------------------------------------------------------
struct ReprErrorArgs {
    const double* cams;      // B x (3 x 12) or 3 x 12 shared (M = 3 views), column-major 3x4; or null:
    long cam_stride;         // 36 or 0
    const double* calm;      // (cams == null) 27 doubles shared: cameras are K1 [I|0], K2 Rt2[b], K3 Rt3[b]
    const double* Rt2; const double* Rt3;    // (cams == null) B x 12 column-major poses
    const double* corresp;   // B x (6 x N) or one shared 6 x N scene (corresp_stride 0)
    long corresp_stride;     // 6 N or 0
    const double* pts3d;     // B x (3 x N) or null: triangulate first (ReprError.m:43-44)
    long B;
    int N;
    double thr;              // inlier threshold in pixels (per coordinate)
    double* err;             // B or null: RMS reprojection error (ReprError.m:65)
    int* inliers;            // B or null: #{n : all six |residuals| <= thr}   (experiments_real.m:98)
};
// true only if S - Z (lower triangles) is positive definite with a margin that covers the approximate reciprocals (v_rcp_f64,
// ~1e-7 relative) of this division-free pivot test: every pivot of the LDL' factorisation above 1e-4 of its diagonal entry
__device__ __forceinline__ bool certainly_positive_definite(const double (&S)[4][4], const double (&Z)[4][4]) {
    double A[4][4], L[4][4], d[4], rd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) A[i][j] = S[i][j] - Z[i][j];
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double dj = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * d[k];
        pd = pd && (dj > 1e-4 * S[j][j]);
        d[j] = dj;
        rd[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * rd[j];
        }
    }
    return pd;
}
__global__ void __launch_bounds__(64, 4) k_repr_error(const ReprErrorArgs a) {
    __shared__ double cam[3][12];
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        wave_sync();
        if (a.cams) {
            for (int v = 0; v < 3; ++v) load_camera_cm(a.cams + b * a.cam_stride + 12 * v, cam[v], lane);
        } else if (lane < 3) {
            const Mat3 K = load_K(a.calm, lane);
            double Rt[12];                                                   // row-major pose of view `lane`
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int r = e >> 2, c = e & 3;
                Rt[e] = (lane == 0) ? ((r == c) ? 1.0 : 0.0) : ((lane == 1) ? a.Rt2[b * 12 + r + 3 * c] : a.Rt3[b * 12 + r + 3 * c]);
            }
            compose_camera_from_pose(K, Rt, cam[lane]);
        }
        wave_sync();
        double P[3][12];
        load_uniform12(cam[0], P[0]);
        load_uniform12(cam[1], P[1]);
        load_uniform12(cam[2], P[2]);
        const double* c = a.corresp + b * a.corresp_stride;
        // Inlier count only: a correspondence whose DLT point X passes the rule has |M X|^2 = sum_v z_v^2 (dx_v^2 + dy_v^2) <=
        // thr^2 X'Z X with z_v = P_v(3,:) X and Z = 2 sum_v P_v(3,:)' P_v(3,:) (each DLT row is the depth times one residual).
        // So S - thr^2 Z positive definite  =>  NO X passes, whatever the eigen-solve would return: a certain outlier for the
        // price of one 4 x 4 pivot test.  Against a wrong hypothesis that is nearly every correspondence of the scene, and the
        // (divergent, gap-dependent) eigen-solves are left to the correspondences that could be inliers.
        const bool count_only = a.inliers && !a.err && !a.pts3d;
        double Zt[4][4];
        if (count_only) {
            const double k2 = 2.0 * a.thr * a.thr;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) Zt[i][j] = k2 * (P[0][8 + i] * P[0][8 + j] + P[1][8 + i] * P[1][8 + j] + P[2][8 + i] * P[2][8 + j]);
        }
        double ss = 0.0;
        int cnt = 0;
#pragma unroll 1
        for (int i = lane; i < a.N; i += WAVE) {
            const Pt6 p = load_pt(c, i);
            double X[4];
            if (a.pts3d) {
                const double* q = a.pts3d + (b * a.N + i) * 3;
                X[0] = q[0]; X[1] = q[1]; X[2] = q[2]; X[3] = 1.0;
            } else {
                double S[4][4];
                tri_zero(S);
                tri_accum(S, P[0], p.v[0], p.v[1]);
                tri_accum(S, P[1], p.v[2], p.v[3]);
                tri_accum(S, P[2], p.v[4], p.v[5]);
                if (count_only && certainly_positive_definite(S, Zt)) continue;
                dlt_point_solve<true, true>(S, cam[0], cam[1], cam[2], true, p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], p.v[5], X);
            }
            bool in = true;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double u = P[v][0] * X[0] + P[v][1] * X[1] + P[v][2] * X[2] + P[v][3] * X[3];
                const double w2 = P[v][4] * X[0] + P[v][5] * X[1] + P[v][6] * X[2] + P[v][7] * X[3];
                const double z = P[v][8] * X[0] + P[v][9] * X[1] + P[v][10] * X[2] + P[v][11] * X[3];
                const double dx = u / z - p.v[2 * v], dy = w2 / z - p.v[2 * v + 1];
                ss += dx * dx + dy * dy;
                in = in && (fabs(dx) <= a.thr) && (fabs(dy) <= a.thr);     // sum(abs(residuals) > th, 1) == 0
            }
            cnt += in ? 1 : 0;
        }
        ss = wave_sum(ss);
        cnt = wave_sum_i(cnt);
        if (lane == 0) {
            if (a.err) a.err[b] = sqrt(ss / (3.0 * (double)a.N));
            if (a.inliers) a.inliers[b] = cnt;
        }
    }
}

// The int32 inlier counts of many hypotheses against ONE shared scene (config 4: experiments_real.m:94-98's rule per hypothesis) with the scene
// staged in LDS once per workgroup: k_repr_error re-reads the scene through L1 / L2 for every hypothesis (a million hypotheses x 19 KB = 3.8 TB/s
// of cache traffic at 5 ms per launch, more than twice what the arithmetic needs).  Four wavefronts share one copy of the scene and walk the
// hypotheses with a grid stride; same arithmetic per correspondence as k_repr_error's count-only path (certain-outlier pivot test, then the
// certified DLT ladder for the correspondences that could be inliers), so the counts are identical.
constexpr int INLIER_WG_WAVES = 4;
__global__ void __launch_bounds__(64 * INLIER_WG_WAVES, 4) k_inlier_count_staged(const ReprErrorArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    double* scene = smem;                                                    // 6 N doubles
    double* camw = smem + 6 * (size_t)a.N + 36 * wave_in_block();            // the wavefront's three cameras (row-major 3 x 4)
    const int lane = lane_id();
    {
        const double2* s2 = reinterpret_cast<const double2*>(a.corresp);
        double2* d2 = reinterpret_cast<double2*>(scene);
        for (int i = thread_in_block(); i < 3 * a.N; i += 64 * INLIER_WG_WAVES) d2[i] = s2[i];
    }
    __syncthreads();
    for (long b = (long)blockIdx.x * INLIER_WG_WAVES + wave_in_block(); b < a.B; b += (long)gridDim.x * INLIER_WG_WAVES) {
        wave_sync();
        if (lane < 3) {
            const Mat3 K = load_K(a.calm, lane);
            double Rt[12];                                                   // row-major pose of view `lane`
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int r = e >> 2, c = e & 3;
                Rt[e] = (lane == 0) ? ((r == c) ? 1.0 : 0.0) : ((lane == 1) ? a.Rt2[b * 12 + r + 3 * c] : a.Rt3[b * 12 + r + 3 * c]);
            }
            compose_camera_from_pose(K, Rt, camw + 12 * lane);
        }
        wave_sync();
        double P[3][12];
        load_uniform12(camw, P[0]);
        load_uniform12(camw + 12, P[1]);
        load_uniform12(camw + 24, P[2]);
        double Zt[4][4];
        const double k2 = 2.0 * a.thr * a.thr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Zt[i][j] = k2 * (P[0][8 + i] * P[0][8 + j] + P[1][8 + i] * P[1][8 + j] + P[2][8 + i] * P[2][8 + j]);
        int cnt = 0;
#pragma unroll 1
        for (int i = lane; i < a.N; i += WAVE) {
            const Pt6 p = load_pt(scene, i);
            double S[4][4], X[4];
            tri_zero(S);
            tri_accum(S, P[0], p.v[0], p.v[1]);
            tri_accum(S, P[1], p.v[2], p.v[3]);
            tri_accum(S, P[2], p.v[4], p.v[5]);
            if (certainly_positive_definite(S, Zt)) continue;
            dlt_point_solve<true, true>(S, camw, camw + 12, camw + 24, true, p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], p.v[5], X);
            bool in = true;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double u = P[v][0] * X[0] + P[v][1] * X[1] + P[v][2] * X[2] + P[v][3] * X[3];
                const double w2 = P[v][4] * X[0] + P[v][5] * X[1] + P[v][6] * X[2] + P[v][7] * X[3];
                const double z = P[v][8] * X[0] + P[v][9] * X[1] + P[v][10] * X[2] + P[v][11] * X[3];
                const double dx = u / z - p.v[2 * v], dy = w2 / z - p.v[2 * v + 1];
                in = in && (fabs(dx) <= a.thr) && (fabs(dy) <= a.thr);     // sum(abs(residuals) > th, 1) == 0
            }
            cnt += in ? 1 : 0;
        }
        cnt = wave_sum_i(cnt);
        if (lane == 0) a.inliers[b] = cnt;
    }
}

// The same counts with FOUR hypotheses per wavefront, one per row of 16 lanes (round 5): a wavefront per hypothesis spends a fifth of its instructions on
// work that is the same for all its lanes -- composing three cameras on three lanes, pinning 36 camera entries and the ten entries of Z to scalar
// registers -- and its seventh trip over a 400-correspondence scene runs 16 lanes of 64.  Here a row composes its hypothesis's cameras once into its own
// 36 doubles of LDS, every position keeps them in vector registers, and 25 trips of 16 cover the scene exactly.  Per correspondence the same expressions
// as above (certain-outlier pivot test, then the certified DLT ladder): identical counts (tests/test_gpu_blocks.py).
__global__ void __launch_bounds__(64 * INLIER_WG_WAVES, 2) k_inlier_count_rows(const ReprErrorArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    double* scene = smem;                                                    // 6 N doubles
    const int lane = lane_id(), p = lane & 15, row = lane >> 4;
    double* camw = smem + 6 * (size_t)a.N + 36 * (4 * wave_in_block() + row);   // the row's three cameras (row-major 3 x 4)
    {
        const double2* s2 = reinterpret_cast<const double2*>(a.corresp);
        double2* d2 = reinterpret_cast<double2*>(scene);
        for (int i = thread_in_block(); i < 3 * a.N; i += 64 * INLIER_WG_WAVES) d2[i] = s2[i];
    }
    __syncthreads();
    for (long b0 = ((long)blockIdx.x * INLIER_WG_WAVES + wave_in_block()) * 4; b0 < a.B; b0 += (long)gridDim.x * INLIER_WG_WAVES * 4) {
        const bool valid = b0 + row < a.B;
        const long b = valid ? b0 + row : a.B - 1;                           // (a tail row repeats the last hypothesis and does not store)
        wave_sync();
        if (p < 3) {
            const Mat3 K = load_K(a.calm, p);
            double Rt[12];                                                   // row-major pose of view p
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int r = e >> 2, c = e & 3;
                Rt[e] = (p == 0) ? ((r == c) ? 1.0 : 0.0) : ((p == 1) ? a.Rt2[b * 12 + r + 3 * c] : a.Rt3[b * 12 + r + 3 * c]);
            }
            compose_camera_from_pose(K, Rt, camw + 12 * p);
        }
        wave_sync();
        double P[3][12];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int c = 0; c < 12; ++c) P[v][c] = camw[12 * v + c];
        double Zt[4][4];
        const double k2 = 2.0 * a.thr * a.thr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Zt[i][j] = k2 * (P[0][8 + i] * P[0][8 + j] + P[1][8 + i] * P[1][8 + j] + P[2][8 + i] * P[2][8 + j]);
        int cnt = 0;
#pragma unroll 1
        for (int i = p; i < a.N; i += 16) {
            const Pt6 q = load_pt(scene, i);
            double S[4][4], X[4];
            tri_zero(S);
            tri_accum(S, P[0], q.v[0], q.v[1]);
            tri_accum(S, P[1], q.v[2], q.v[3]);
            tri_accum(S, P[2], q.v[4], q.v[5]);
            if (certainly_positive_definite(S, Zt)) continue;
            dlt_point_solve<true, true>(S, camw, camw + 12, camw + 24, true, q.v[0], q.v[1], q.v[2], q.v[3], q.v[4], q.v[5], X);
            bool in = true;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double u = P[v][0] * X[0] + P[v][1] * X[1] + P[v][2] * X[2] + P[v][3] * X[3];
                const double w2 = P[v][4] * X[0] + P[v][5] * X[1] + P[v][6] * X[2] + P[v][7] * X[3];
                const double z = P[v][8] * X[0] + P[v][9] * X[1] + P[v][10] * X[2] + P[v][11] * X[3];
                const double dx = u / z - q.v[2 * v], dy = w2 / z - q.v[2 * v + 1];
                in = in && (fabs(dx) <= a.thr) && (fabs(dy) <= a.thr);     // sum(abs(residuals) > th, 1) == 0
            }
            cnt += in ? 1 : 0;
        }
        const double tot = row_sum16((double)cnt);
        if (p == 0 && valid) a.inliers[b] = (int)tot;
    }
}

// ---- transform_TFT ----------------------------------------------------------------------------
struct TransformArgs {
    const double* T;         // B x 27
    const double* M1; const double* M2; const double* M3;    // B x 9 (3x3 column-major) each, or shared (m_stride 0)
    long m_stride;           // 9 or 0
    long B;
    int inverse;             // 0 or 1 (transform_TFT.m:36,42)
    double* Tout;            // B x 27
};
__global__ void __launch_bounds__(64, 4) k_transform_tft(const TransformArgs a) {
    __shared__ double t[27], tn[27], mats[27], raw[27];
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        wave_sync();
        if (lane < 27) {
            t[lane] = a.T[b * 27 + lane];
            const double* src = (lane < 9) ? a.M1 : ((lane < 18) ? a.M2 : a.M3);
            raw[lane] = src[b * a.m_stride + lane % 9];                      // column-major: M(r,c) at r + 3c
        }
        wave_sync();
        auto mat = [&](int v) { Mat3 M; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M.m[r][c] = raw[9 * v + r + 3 * c]; return M; };
        if (a.inverse) {
            transform_tft_inverse(t, tn, mats, mat);
        } else {
            // T_new(:,:,i) = M2 (sum_j M1i(j,i) T_old(:,:,j)) M3.'   (:37-40)
            if (lane < 3) {
                Mat3 M = mat(lane);
                if (lane == 0) M = mat3_inv(M);
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) mats[9 * lane + 3 * r + c] = M.m[r][c];
            }
            wave_sync();
            double val = 0.0;
            if (lane < 27) {
                const int i = lane / 9, k = (lane % 9) / 3, j = lane % 3;
                for (int c = 0; c < 3; ++c)
                    for (int d = 0; d < 3; ++d) {
                        const double mix = mats[i] * t[c + 3 * d] + mats[3 + i] * t[c + 3 * d + 9] + mats[6 + i] * t[c + 3 * d + 18];
                        val += mats[9 + 3 * j + c] * mix * mats[18 + 3 * k + d];
                    }
            }
            const double nn = wave_sum(val * val);
            if (lane < 27) tn[lane] = val * rsqrt(nn);                       // :49
            wave_sync();
        }
        if (lane < 27) a.Tout[b * 27 + lane] = tn[lane];
    }
}

// ---- R_t_from_TFT -----------------------------------------------------------------------------
struct RtFromTftArgs {
    const double* T;         // B x 27 (pixel-coordinate tensor)
    const double* calm; long calm_stride;
    const double* corresp;   // B x (6 x N)
    long B; int N;
    double* Rt2; double* Rt3; int* status;
};
__global__ void __launch_bounds__(64, 2) k_rt_from_tft(const RtFromTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        wave_sync();
        if (lane < 27) { w->calm[lane] = a.calm[b * a.calm_stride + lane]; w->T1[lane] = a.T[b * 27 + lane]; }
        wave_sync();
        const double* pts = a.corresp + b * 6 * (long)a.N;
        int st = rt_from_tft_wave(w, pts, a.N, nullptr);
        write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
        if (lane == 0 && a.status) a.status[b] = st;
    }
}

// ---- linearTFT ----------------------------------------------------------------------------------
struct LinearTftOnlyArgs {
    const double* corresp;   // B x (6 x N): the points handed to linearTFT (rows x1;y1;x2;y2;x3;y3), used as given
    long B; int N; int flags;
    double* T;               // B x 27, unit norm (t = Up * tp)
    double* P2; double* P3;  // B x 12 each (3x4 column-major) or null; P1 = [I|0]
    int* status;
};
template <bool JAC>
__global__ void __launch_bounds__(64, 2) k_linear_tft(const LinearTftOnlyArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLds* jw = JAC ? reinterpret_cast<JacobiLds*>(smem + base) : nullptr;
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;
        wave_sync();
        const double* pts = a.corresp + b * 6 * (long)a.N;
        if (lane < 9) w->nrm[lane] = (lane % 3 == 0) ? 1.0 : 0.0;            // identity "normalisation"
        wave_sync();
        int st = ST_OK;
        if (a.N < 7) st = ST_TOO_FEW;
        else {
            const bool ok = linear_tft_wave<JAC>(w, jw, pts, a.N, true, nullptr);
            if (!ok) st = ST_RETRY;
            else {
                if (lane < 27) a.T[b * 27 + lane] = w->t[lane];
                if (lane < 12 && a.P2) {
                    const int r = lane % 3, c = lane / 3;
                    a.P2[b * 12 + lane] = (c < 3) ? w->pa[3 * c + r] : w->epi[r];
                    a.P3[b * 12 + lane] = (c < 3) ? w->pa[9 + 3 * c + r] : w->epi[3 + r];
                }
            }
        }
        if (lane == 0) a.status[b] = st;
    }
}

}  // namespace tff
