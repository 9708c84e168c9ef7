// OptimFPoseEstimation as three launches (the arrangement of the iterative trifocal methods, gh_rows_kernel.h / gh_wg_kernel.h):
//
//   k_optimf_linear_rows   four triplets per wavefront: Normalize2Ddata x3, linearF for both view pairs (f_rows_kernel.h), F / |F|_F
//                          (OptimFPoseEstimation.m:46-50 via optimF.m:46-50) -> a record per triplet (F21, F31 in the normalised frame | normalisations)
//   k_optimf_refine        one wavefront per triplet: initial x_est by two-view triangulation, Gauss-Helmert on F(:) for both pairs
//                          (optimF.m:52-69, f_kernel.h::optim_f_refine) -> refined F21, F31 in the record, iter = it1 + it2
//   k_optimf_finish_rows   four triplets per wavefront: back to pixels, rank 2 (optimF.m:72-76), E = K' F K, recover_R_t x2, t3 scale, Reconst,
//                          T = TFT_from_P (OptimFPoseEstimation.m:53-73; the pose tail of tft_rows_kernel.h)
//
// Why: the fused one-triplet kernel k_f_pose<false, 1> spent a quarter of its time in its linear stage and a sixth in the pose tail, both of
// which are lane-sparse or make eight to ten passes over the correspondences when a wavefront has one triplet; the Gauss-Helmert iteration
// itself keeps per-correspondence state and gains nothing from the row layout (built, measured, not kept).  With the
// stages split each runs in the layout that suits it.  A triplet a fast tier cannot finish anywhere is marked ST_RETRY and redone whole by
// k_f_pose<true, 1>.
// Reference: F_methods/OptimFPoseEstimation.m:44-73, F_methods/optimF.m:34-109.
#pragma once
#include "f_rows_kernel.h"
#include "f_kernel.h"

namespace tff {

// wavefronts per SIMD k_optimf_refine is compiled for.  Measured (10 000 x 200, whole method): 2 -> 0.69 ms (256 registers: no spills, all 54 sums
// of an iteration in one sweep), 3 -> 0.84 ms (168 registers, 19 spilled, two sweeps), 4 -> 1.32 ms (81 spilled)
constexpr int OPTIMF_REFINE_WAVES = 2;
constexpr int OPTIMF_REC_DOUBLES = 32;    // F21 9 | F31 9 (row-major, normalised frame, unit Frobenius norm) | nrm 9 | pad

struct OptimFStageArgs {
    LinearTftArgs la;
    double* rec;             // B x OPTIMF_REC_DOUBLES
    double* spill; long spill_stride;   // k_optimf_refine: per-correspondence state in global slices (large N), see LinearTftArgs
};

__global__ void __launch_bounds__(64, 2) k_optimf_linear_rows(const OptimFStageArgs sa) {
    TFF_DYNAMIC_LDS(double, smem);
    const LinearTftArgs& a = sa.la;
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const RowJob j = rows_begin(a, w, blk, N);
        int status = ST_OK;
        if (N < 8) {                                                         // optimF.m:36-38 (wave-uniform: N is the batch's)
            status = ST_TOO_FEW;
        } else {
            {
                double cen[6];
                rows_centroids(j.src, N, cen);                               // optimF.m:46-47
                rows_distances_moments_f(j.src, N, cen, w->nrm, w->pa, w->mom);
            }
            wave_sync();
            const bool ok = rows_linear_f_middle<true>(w, nullptr, w->t, w->pa, nullptr);   // optimF.m:49-50: F = linearF(...); F = F / |F|_F
            if (!ok) status = ST_RETRY;
            if (j.valid && ok) {
                double* r = sa.rec + j.b * OPTIMF_REC_DOUBLES;
                r[p] = w->t[p];
                if (p < 2) r[16 + p] = w->t[16 + p];
                if (p < 9) r[18 + p] = w->nrm[p];
            }
        }
        if (p == 0 && j.valid) { a.status[j.b] = status; if (a.iter) a.iter[j.b] = 0; }
    }
}

// k_optimf_refine keeps what the iteration touches in every pass in LDS: xi (4 N) and -- STAGE_X -- the NORMALISED observations (6 N, the
// correspondences mapped once: every pass of the fused kernel re-read them from L2 and mapped them again, a dependent global load per trip that
// two wavefronts per SIMD cannot hide).  18 KB at N = 200: eight wavefronts per CU, as the 256-register build allows.
struct OptimFRefineLds {
    double nrm[10];        // map from the points the passes read to the normalised frame (STAGE_X: the identity)
    double Fm[18];         // F21, F31 (row-major, normalised frame)
    double PA[12], PB[12]; // cameras of the initial triangulation   (optimF.m:54-55)
};
constexpr int OPTIMF_REFINE_FIXED_DOUBLES = (int)(sizeof(OptimFRefineLds) / sizeof(double)) + ((OPTIMF_FIXED_DOUBLES + 1) & ~1);
__host__ __device__ inline size_t optimf_refine_lds_bytes(int N, bool stage_x) {
    return (size_t)(OPTIMF_REFINE_FIXED_DOUBLES + 4 * N + 2 + (stage_x ? 6 * N : 0)) * sizeof(double);
}

template <int WAVES_PER_SIMD, bool STAGE_X>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) k_optimf_refine(const OptimFStageArgs sa) {
    TFF_DYNAMIC_LDS(double, smem);
    const LinearTftArgs& a = sa.la;
    OptimFRefineLds* w = reinterpret_cast<OptimFRefineLds*>(smem);
    OptimFLds* og = reinterpret_cast<OptimFLds*>(smem + sizeof(OptimFRefineLds) / sizeof(double));
    double* var = smem + OPTIMF_REFINE_FIXED_DOUBLES;
    double* xn = var;                                                        // STAGE_X: 6 N
    double* oxi = sa.spill ? sa.spill + blockIdx.x * sa.spill_stride : var + (STAGE_X ? 6 * a.N : 0);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (a.status[b] != ST_OK) continue;                                  // wave-uniform: too few points, or left to the exact kernel
        const int N = opaque_int(a.N);
        const double* src = a.corresp + b * 6 * (long)N;
        wave_sync();
        double* r = sa.rec + b * OPTIMF_REC_DOUBLES;
        if (lane < 18) w->Fm[lane] = r[lane];
        if (lane < 9) w->nrm[lane] = r[18 + lane];
        wave_sync();
        const double* pts = src;
        if constexpr (STAGE_X) {
            for (int i = lane; i < N; i += WAVE) {
                const Pt6 q = premap(load_pt(src, i), w->nrm);
#pragma unroll
                for (int k = 0; k < 6; ++k) xn[6 * i + k] = q.v[k];
            }
            wave_sync();
            if (lane < 9) w->nrm[lane] = (lane % 3 == 0) ? 1.0 : 0.0;        // the passes below read normalised points
            wave_sync();
            pts = xn;
        }
        int gst = ST_OK, iters = 0;
        bool fine = true;
#pragma unroll 1
        for (int pair = 0; pair < 2; ++pair) {                               // [F21,it1] = optimF(...), [F31,it2] = optimF(...)   (OptimFPoseEstimation.m:48-49)
            if (lane == 0) {                                                 // f_kernel.h::optim_f_refine
                Mat3 F, Ft;
                for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) F.m[rr][c] = w->Fm[9 * pair + 3 * rr + c];
                Ft = mat3_T(F);
                double e[3];
                fine = null3<false>(Ft, e) && fine;                          // epi21 = U(:,3): left null vector   (optimF.m:53)
                for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 4; ++c) w->PA[4 * rr + c] = (rr == c) ? 1.0 : 0.0;   // P1 = [I|0]   (:54)
                for (int c = 0; c < 3; ++c) {                                // P2 = [crossM(epi21)*F, epi21]   (:55)
                    w->PB[c] = -e[2] * F.m[1][c] + e[1] * F.m[2][c];
                    w->PB[4 + c] = e[2] * F.m[0][c] - e[0] * F.m[2][c];
                    w->PB[8 + c] = -e[1] * F.m[0][c] + e[0] * F.m[1][c];
                }
                w->PB[3] = e[0]; w->PB[7] = e[1]; w->PB[11] = e[2];
            }
            if (lane < 9) og->p[lane] = w->Fm[9 * pair + 3 * (lane % 3) + lane / 3];   // p = F(:) column-major   (:61)
            wave_sync();
            {                                                                // x_est: the two reprojections of the two-view DLT point   (:56-60)
                double PA[12], PB[12];
                load_uniform12(w->PA, PA);
                load_uniform12(w->PB, PB);
                for (int i = lane; i < N; i += WAVE) {
                    const Pt6 q = premap(load_pt(pts, i), w->nrm);
                    double X[4];
                    const bool conv = dlt_point<false, false>(PA, PB, PB, w->PA, w->PB, w->PB, false, q.v[0], q.v[1], pair ? q.v[4] : q.v[2],
                                                              pair ? q.v[5] : q.v[3], 0.0, 0.0, X);
                    fine = fine && conv;
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const double (&P)[12] = (v == 0) ? PA : PB;
                        const double pa = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
                        const double pb = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
                        const double pc = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
                        oxi[4 * (long)i + 2 * v] = pa / pc;
                        oxi[4 * (long)i + 2 * v + 1] = pb / pc;
                    }
                }
            }
            if (wave_any(!fine)) break;                                      // a fast tier gave up: the exact kernel redoes the triplet
            wave_sync();
            iters += gauss_helmert_f_wave<WAVES_PER_SIMD == 2>(w->nrm, og, oxi, pts, N, pair + 1, &gst);   // :66
            wave_sync();
            if (lane < 9) w->Fm[9 * pair + 3 * (lane % 3) + lane / 3] = og->p[lane];   // F = reshape(p_opt,3,3)   (:69)
            wave_sync();
        }
        fine = !wave_any(!fine);
        if (fine && lane < 18) r[lane] = w->Fm[lane];
        if (lane == 0) {
            if (a.iter) a.iter[b] = iters;
            a.status[b] = !fine ? ST_RETRY : ((gst != ST_OK) ? -gst : ST_OK);   // negative: reported after k_optimf_finish_rows has produced the outputs
        }
    }
}

__global__ void __launch_bounds__(64, 2) k_optimf_finish_rows(const OptimFStageArgs sa) {
    TFF_DYNAMIC_LDS(double, smem);
    LinearTftArgs la = sa.la;
    la.iter = nullptr; la.dbg = nullptr;
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < la.B; blk += gridDim.x) {
        const int N = opaque_int(la.N);
        RowJob j = rows_begin(la, w, blk, N);                                // (calibration -> w->calm)
        const int s0 = la.status[j.b];
        const bool dead = s0 > 0;                                            // ST_TOO_FEW: no outputs; ST_RETRY: the exact kernel's (it stores them all)
        if (s0 == ST_TOO_FEW) rows_store_nan(la, j, N);
        const double* r = sa.rec + j.b * OPTIMF_REC_DOUBLES;
        w->t[p] = dead ? (((p % 9) % 4 == 0) ? 1.0 : 0.0) : r[p];                  // (a dead row works on a harmless matrix and stores nothing)
        if (p < 2) w->t[16 + p] = dead ? ((p == 1) ? 1.0 : 0.0) : r[16 + p];
        if (p < 9) w->nrm[p] = dead ? ((p % 3 == 0) ? 1.0 : 0.0) : r[18 + p];
        wave_sync();
        j.valid = j.valid && !dead;
        bool nok = true;
        if (p < 2) {
            const int v2 = p + 1;
            Mat3 F;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) F.m[rr][c] = w->t[9 * p + 3 * rr + c];
            F = mat3_mul(mat3_mul(mat3_T(normal_matrix(w->nrm, v2)), F), normal_matrix(w->nrm, 0));     // optimF.m:72: back to pixel coordinates
            double v3[3], fv[3];
            nok = null3<false>(F, v3);                                       // :75-76: rank 2 again
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) fv[rr] = F.m[rr][0] * v3[0] + F.m[rr][1] * v3[1] + F.m[rr][2] * v3[2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) F.m[rr][c] -= fv[rr] * v3[c];
            const Mat3 E = mat3_mul(mat3_mul(mat3_T(load_K(w->calm, v2)), F), load_K(w->calm, 0));      // recover_R_t: E = K2' F K1
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) rt->Ein[9 * p + 3 * rr + c] = E.m[rr][c];
        }
        const bool ok = !row_any(!nok);
        wave_sync();
        rows_recover_prepare(w, rt);
        int status = rows_pose_tail<true>(la, w, rt, j, N, ok);
        if (s0 < 0 && status != ST_RETRY) status = -s0;                         // OptimFPoseEstimation: the Gauss-Helmert loop's NaN / rank break
        if (p == 0 && j.valid) la.status[j.b] = status;
    }
}

}  // namespace tff
