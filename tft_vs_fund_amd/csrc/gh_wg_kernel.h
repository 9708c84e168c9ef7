// Gauss-Helmert methods (Ressl, Nordberg, FaugPapa) as three launches, with a WORKGROUP of four wavefronts per
// triplet for the iteration itself:
//
//   k_gh_linear<JAC>   one wavefront per triplet (2 waves / SIMD): Normalize2Ddata x3 + linearTFT
//                      (ResslTFTPoseEstimation.m:48-53) -> a 64-double record per triplet (t, a, epipoles, normalisations)
//   k_gh_block<Model>  256 threads per triplet: model set-up, initial observations (projective triangulation +
//                      reprojection, :72-75), Gauss_Helmert.m:38-83 -> optimised tensor (27 doubles)
//   k_gh_finish        one wavefront per triplet: transform_TFT, R_t_from_TFT, Reconst (:96-103)
//
// Why: the per-correspondence state of the iteration (xi, W+, W+w: 20 N doubles) plus the normal-equation workspace
// take ~58 KB of LDS at N = 200.  With one wavefront per triplet (k_gh_tft_pose in gh_kernel.h) that allows two
// wavefronts per CU -- two of the four SIMDs idle, the other two running a lone wave at ~8 cycles per dependent
// instruction.  Four wavefronts sharing one triplet's LDS put 8 waves on a CU (2 per SIMD) and split the
// per-correspondence sweeps four ways; the wave-serial steps (parameter Jacobian, KKT solve) run on one wave of
// the group, chosen by blockIdx so that they spread over the SIMDs.
//
// Same arithmetic per correspondence as gh_kernel.h (shared device functions); sums over correspondences are
// grouped per wavefront and then added, so results differ from the single-wave kernel by rounding only
// (TFF_OPT_KERNEL = 1 selects the single-wave kernel for A/B runs).
#pragma once
#include "gh_kernel.h"

namespace tff {

constexpr int GH_WG_WAVES = 4;
constexpr int GH_WG_THREADS = 64 * GH_WG_WAVES;
constexpr int GH_REC_DOUBLES = 64;        // t 27 | pa 18 | epi 6 | nrm 9 | pad
// record strides of the per-correspondence state xi / W+.  Even (16-byte aligned records): the loads are ds_read_b128.  Padding them to
// odd strides (7 / 11 doubles: 32 distinct bank pairs for consecutive lanes instead of 16) was measured SLOWER -- Ressl 3.0 -> 3.5 ms,
// the sweeps 17 k -> 26 k cycles -- because the records lose their alignment and every access becomes two ds_read_b64.
// A component-major layout (xi[k * N + i]: conflict-free ds_read_b64) was measured slower too -- the ten sweeps 17.3 k -> 22.5 k cycles,
// twice the LDS instructions and 20 more spilled registers: the SQ_LDS_BANK_CONFLICT share of profiles/r2_ressl_* (0.75 of the LDS
// instruction cycles) is not what bounds these passes, LDS instruction issue is.
constexpr int GH_XI = 6, GH_PP = 10, GH_SN = 6;         // GH_SN: per correspondence n (4), cs, pad -- N records behind the N W+ records (pp + GH_PP N)

// LDS of k_gh_block after the PoseLds header: p, dt, Tc, dT, D, H, Y, M, V (112 doubles of scratch), xi (6N), W+ (10N), reduction slots.
// `pinv` (FaugPapa: D is the identity and never stored): the eigenvectors + scratch of the pseudo-inverse (n^2 + 2n) overlay D | H | Y,
// all dead during the solve, which keeps the fixed part under 40 KB -> four workgroups per CU once xi / W+ are spilled.
// Ghat (729) lives in PoseLds::Lp (the linear stage's Cholesky factor is not needed here); the ten sweeps write their sums straight
// into H (each sweep runs on one wavefront over all correspondences); W+ w is recomputed where needed.
// Ressl at N = 200: 50.4 KB with the header -> three workgroups per CU.
__host__ __device__ inline int gh_wg_lds_doubles(int u, int c, int N, bool pinv) {
    const int n = u + c;
    const int strong = pinv ? 0 : ((u * (u + 1) / 2 + u + 1) & ~1);          // sums of the factored strong-direction terms (minimal parameterisations)
    return 2 * ((u + 1) & ~1) + 2 * c + 28 + 28 + 27 * u + 298 + ((27 * u > 298) ? 27 * u : 298) + n * (n + 1) + 112 + strong + GH_XI * N + (GH_PP + GH_SN) * N + 16 + 8;
}
__device__ inline GhWork gh_wg_carve(double* base, PoseLds* w, int u, int c, int N, bool pinv, double** red) {
    GhWork g;
    const int n = u + c;
    double* q = base;
    g.p = q; q += (u + 1) & ~1;
    g.dt = q; q += ((u + 1) & ~1) + 2 * c;
    g.Tc = q; q += 28;
    g.dT = q; q += 28;
    g.D = q; q += 27 * u;
    g.G = w->Lp;
    g.H = q; q += 298;
    g.Y = q; q += (27 * u > 298) ? 27 * u : 298;
    g.M = q; q += n * (n + 1);
    g.V = pinv ? g.D : q; q += 112;                                                // 108 doubles of scratch (Nordberg's rotations)
    g.S = pinv ? nullptr : q; q += pinv ? 0 : ((u * (u + 1) / 2 + u + 1) & ~1);
    g.xi = q; q += GH_XI * N;
    g.pp = q; q += (GH_PP + GH_SN) * N;
    *red = q;
    g.u = u; g.c = c;
    return g;
}
// w = -f - B (x - xi)   (Gauss_Helmert.m:58)
__device__ __forceinline__ void gh_w_vector(const PoseLds* w, const double* pts, int i, const double (&o)[6], const double (&f)[4],
                                            const double (&B)[4][6], double (&wv)[4]) {
    const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = -f[a];
#pragma unroll
        for (int k = 0; k < 6; ++k) s -= B[a][k] * (x.v[k] - o[k]);
        wv[a] = s;
    }
}

struct GhWgArgs {
    const double* corresp; const double* calm; long calm_stride; long B; int N; int flags;
    double* rec;             // B x GH_REC_DOUBLES (k_gh_linear out, k_gh_block / k_gh_finish in)
    double* topt;            // B x 27 (k_gh_block out, k_gh_finish in): optimised tensor in the normalised frame
    double* Rt2; double* Rt3; double* T; double* reconst; int* iter; int* status; double* dbg;
    double* spill; long spill_stride;   // see LinearTftArgs
    double* init_rec;        // B x Model::PRE_DOUBLES (k_nordberg_init out, k_gh_block<NordbergModel> in) or null
    const double* pre;       // null, or B x tff::PRE_DOUBLES from k_tft_moments (tft_moments_kernel.h): what k_gh_linear_rows<true> starts from
};

template <bool JAC>
__global__ void __launch_bounds__(64, 2) k_gh_linear(const GhWgArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLds* jw = JAC ? reinterpret_cast<JacobiLds*>(smem + base) : nullptr;
    double* lds_pts = smem + base + (JAC ? ((JACOBI_LDS_DOUBLES + 1) & ~1) : 0);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        wave_sync();
        if (a.flags & FLAG_STAGE_LDS) { stage_points(pts, lds_pts, N); pts = lds_pts; }
        int status = ST_OK;
        if (N < 7) {
            status = ST_TOO_FEW;
        } else {
            normalise3(pts, N, w->nrm);
            if (!linear_tft_wave<JAC>(w, jw, pts, N, true, nullptr)) {
                status = ST_RETRY;
            } else {
                double* r = a.rec + b * GH_REC_DOUBLES;
                if (lane < 27) r[lane] = w->t[lane];
                if (lane < 18) r[27 + lane] = w->pa[lane];
                if (lane < 6) r[45 + lane] = w->epi[lane];
                if (lane < 9) r[51 + lane] = w->nrm[lane];
            }
        }
        if (lane == 0) { a.status[b] = status; if (a.iter) a.iter[b] = 0; }
    }
}

// Models whose initial parameters have a serial part that runs ahead of the block kernel, one triplet per LANE (NordbergModel::init_serial)
template <class M, class = void> struct gh_has_preinit { static constexpr bool value = false; };
template <class M> struct gh_has_preinit<M, decltype((void)M::PREINIT)> { static constexpr bool value = M::PREINIT; };

// NordbergTFTPoseEstimation.m:56-78 for 64 triplets per wavefront, one per lane: cameras of the linear solution (record of k_gh_linear) ->
// P2, P3 after the projective fix-up | U, V, W | their axis-angle vectors | deficient flag.  The block kernel's owner wavefront used to run
// this on its lane 0 (58 k cycles per triplet, 255 threads waiting: an eighth of the method's time).
__global__ void __launch_bounds__(64) k_nordberg_init(const GhWgArgs a) {
    const long b = (long)blockIdx.x * WAVE + lane_id();
    if (b >= a.B || a.status[b] != ST_OK) return;
    const double* r = a.rec + b * GH_REC_DOUBLES;
    double P2[12], P3[12], rot[27], p9[9];
#pragma unroll
    for (int e = 0; e < 12; ++e) {                                           // gh_linear_cameras: P2 = [reshape(a(1:9),3,3) e21], P3 = [reshape(a(10:18),3,3) e31]
        const int rr = e >> 2, c = e & 3;
        P2[e] = (c < 3) ? r[27 + 3 * c + rr] : r[45 + rr];
        P3[e] = (c < 3) ? r[27 + 9 + 3 * c + rr] : r[45 + 3 + rr];
    }
    int deficient = 0;
    NordbergModel::init_serial(P2, P3, rot, p9, &deficient);
    double* o = a.init_rec + b * NordbergModel::PRE_DOUBLES;
#pragma unroll
    for (int e = 0; e < 12; ++e) { o[e] = P2[e]; o[12 + e] = P3[e]; }
#pragma unroll
    for (int e = 0; e < 27; ++e) o[24 + e] = rot[e];
#pragma unroll
    for (int e = 0; e < 9; ++e) o[51 + e] = p9[e];
    o[60] = (double)deficient;
}

// ---- block-level helpers (256 threads) -------------------------------------------------------------------------------
// (WV: wavefronts of the workgroup -- four everywhere but in k_pi_block<PiColModel>, which runs two per triplet so that four workgroups per CU
// leave each thread 256 registers)
template <int WV>
__device__ __forceinline__ double block_sum_w(double v, double* red) {
    static_assert(WV == 4 || WV == 2, "two or four wavefronts");
    v = wave_sum(v);
    if (lane_id() == 0) red[wave_in_block()] = v;
    __syncthreads();
    const double r = (WV == 4) ? (red[0] + red[1]) + (red[2] + red[3]) : red[0] + red[1];
    __syncthreads();
    return r;
}
template <int WV>
__device__ __forceinline__ double block_max_w(double v, double* red) {
    v = wave_max(v);
    if (lane_id() == 0) red[wave_in_block()] = v;
    __syncthreads();
    const double a = (red[0] > red[1]) ? red[0] : red[1];
    double r = a;
    if constexpr (WV == 4) { const double b = (red[2] > red[3]) ? red[2] : red[3]; r = (a > b) ? a : b; }
    __syncthreads();
    return r;
}
template <int WV> __device__ __forceinline__ bool block_any_w(bool p, double* red) { return block_sum_w<WV>(p ? 1.0 : 0.0, red) != 0.0; }
// three maxima and one "any" in ONE barrier pair (two-wavefront workgroups: red[0 .. 7]; four wavefronts: the single reductions)
template <int WV>
__device__ __forceinline__ void block_max3_any_w(double& a, double& b, double& c, bool& flag, double* red) {
    if constexpr (WV == 2) {
        a = wave_max(a); b = wave_max(b); c = wave_max(c);
        const double f = wave_any(flag) ? 1.0 : 0.0;
        if (lane_id() == 0) { const int w = wave_in_block(); red[w] = a; red[2 + w] = b; red[4 + w] = c; red[6 + w] = f; }
        __syncthreads();
        a = (red[0] > red[1]) ? red[0] : red[1];
        b = (red[2] > red[3]) ? red[2] : red[3];
        c = (red[4] > red[5]) ? red[4] : red[5];
        flag = (red[6] + red[7]) != 0.0;
        __syncthreads();
    } else {
        a = block_max_w<WV>(a, red); b = block_max_w<WV>(b, red); c = block_max_w<WV>(c, red);
        flag = block_any_w<WV>(flag, red);
    }
}
__device__ __forceinline__ double block_sum(double v, double* red) { return block_sum_w<GH_WG_WAVES>(v, red); }
__device__ __forceinline__ double block_max(double v, double* red) { return block_max_w<GH_WG_WAVES>(v, red); }
__device__ __forceinline__ bool block_any(bool p, double* red) { return block_any_w<GH_WG_WAVES>(p, red); }

// The wavefront of the workgroup that runs the wave-serial steps (parameter Jacobian, KKT solve, pseudo-inverse).  The workgroups that
// share a CU should run theirs on DIFFERENT SIMDs.  `blockIdx & 3` does not achieve that: consecutive workgroups go round-robin over the
// eight XCDs, so the workgroups resident on one CU all have the same blockIdx mod 4, hence the same serial wave index, hence (waves are
// dealt to the SIMDs in order) the same SIMD -- four serial waves time-slicing one SIMD while three idle, measured as ~30 cycles per
// instruction in those steps.  The hardware knows better: the workgroup's slot number on its CU picks the SIMD, the wave that actually
// runs there takes the job.  red: 4 doubles of scratch at red[12 ..].
template <int WV>
__device__ inline int pick_serial_wave_w(double* red) {
    if (lane_id() == 0) red[12 + wave_in_block()] = (double)hw_simd_id();
    __syncthreads();
    const int target = hw_workgroup_slot() & 3;                              // the SIMD this workgroup's serial steps should run on
    int own = -1;
#pragma unroll
    for (int w = WV - 1; w >= 0; --w) own = ((int)red[12 + w] == target) ? w : own;
    own = (own < 0) ? (target & (WV - 1)) : own;                             // no wave of this workgroup on that SIMD
    __syncthreads();
    return wave_uniform_i(own);
}
__device__ inline int pick_serial_wave(double* red) { return pick_serial_wave_w<GH_WG_WAVES>(red); }

// x = pinv(M) b through the eigen-decomposition (wave_pinv_solve_sym) on the wavefront `own` of the workgroup; the others wait.
// (A workgroup-parallel cyclic Jacobi was measured here first: 11 sweeps x 39 rounds x 2 barriers on the 39 x 39 matrix, 1.5 ms per
// solve; tridiagonalisation + QL on one wavefront does a tenth of the arithmetic and leaves the CU to the other workgroups.)
// scratch: 2 n doubles.
template <int n>
__device__ inline void block_pinv_solve_sym(double* M, double* V, double* sol, double* scratch, const int own) {
    if (wave_in_block() == own) wave_pinv_solve_sym<true>(M, V, n, sol, scratch);
    __syncthreads();
}

// one accumulation sweep (30 of the 297 sums) over ALL correspondences by the calling wavefront -> H
template <int CH>
__device__ inline void gh_sweep_part(const GhWork& g, double* Hp, int N, const double (&T)[27], const PoseLds* w, const double* pts) {
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        GhPoint pt;
#pragma unroll
        for (int k = 0; k < 6; ++k) pt.o[k] = g.xi[GH_XI * i + k];
#pragma unroll
        for (int k = 0; k < 10; ++k) pt.Wp[k] = g.pp[GH_PP * i + k];
        if constexpr (CH == 9) {                                             // W+ w is not stored: recompute w from the observations
            double f[4], B[4][6], wv[4];
            tril_block(T, pt.o, f, B);
            gh_w_vector(w, pts, i, pt.o, f, B, wv);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                pt.ww[a] = wp_at(pt.Wp, a, 0) * wv[0] + wp_at(pt.Wp, a, 1) * wv[1] + wp_at(pt.Wp, a, 2) * wv[2] + wp_at(pt.Wp, a, 3) * wv[3];
        }
        if constexpr (CH < 9) {
            const double hh[6] = {pt.o[0] * pt.o[0], pt.o[0] * pt.o[1], pt.o[0], pt.o[1] * pt.o[1], pt.o[1], 1.0};
            gh_accum_chunk<CH>(pt, hh, acc);
        } else {
            gh_accum_rhs(pt, acc);
        }
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0) {
        if (CH < 9) { if (idx < 30) Hp[30 * CH + idx] = tot; }
        else if (idx < 27) Hp[270 + idx] = tot;
    }
}

// x_est: reprojection of the projective triangulation with P1 (Pfin[0]), P2 (P[0]), P3 (P[1])   (ResslTFT...m:72-75)
template <int WV = GH_WG_WAVES>
__device__ inline void gh_block_reproject(PoseLds* w, const double* pts, int N, double* xi) {
    double PA[12], PB[12], PC[12];
    load_uniform12(w->Pfin[0], PA);
    load_uniform12(w->P[0], PB);
    load_uniform12(w->P[1], PC);
#pragma unroll 1
    for (int i = thread_in_block(); i < N; i += WV * WAVE) {
        const Pt6 p = premap(load_pt(pts, i), w->nrm);
        double X[4];
        dlt_point<true>(PA, PB, PC, w->Pfin[0], w->P[0], w->P[1], true, p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], p.v[5], X);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double (&P)[12] = (v == 0) ? PA : ((v == 1) ? PB : PC);
            const double aa = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
            const double bb = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
            const double cc = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
            xi[GH_XI * (long)i + 2 * v] = aa / cc;
            xi[GH_XI * (long)i + 2 * v + 1] = bb / cc;
        }
    }
}

// Gauss_Helmert.m:38-83, one workgroup per problem.  `own`: the wavefront that runs the wave-serial steps.
// Round 5 measured six changes to THIS iteration against the round-4 build on the same box (tools/ab_libs.py, profiles/r5_ab_libs.txt; Ressl /
// Nordberg, ms per 10 000 x 200).  KEPT (run 5): the finite check riding on a speculative weight pass + the strong direction (n, cs) kept in a
// record of its own BEHIND the packed W+ records, so that the v pass does not deflate every block a second time: 2.367 -> 2.268 / 2.673 -> 2.579
// (the speculation alone: 2.340 / 2.669).  NOT kept:
//   * finite check riding on a speculative weight pass (+ the strong direction n, cs kept in a 16-double W+ record for the v pass): 2.371 -> 2.365 /
//     2.659 -> 2.669; the v pass fell from 19 k to 12 k cycles but the 128-byte record stride cost the ten sweeps as much (W+ is in global slices);
//   * check and tolerance bounds in one pass: 2.369 -> 2.430 / 2.672 -> 2.719 (the blocks stay under the truncation limit here, so round 4 never ran
//     the bounds pass: this only added work);
//   * the strong-direction sums on the matrix core (gh_kernel.h::StrongGram; kept for the Pi kernels, U = 27: -12 %): 2.449 -> 2.432 / 2.742 -> 2.769;
//   * Ghat D and D'Y on the matrix core (14 matrix instructions per wavefront and product): 2.364 -> 2.412 / 2.657 -> 2.711 with the speculative pass.
// What the phase stamps attribute to a phase is mostly the wait for the SIMD shared with three other workgroups' wavefronts: shortening one
// phase of one workgroup moves the wait, not the kernel's 0.74 VALU-busy total.
template <class Model, int WV>
__device__ inline int gauss_helmert_block(PoseLds* w, GhWork& g, double* red, Model& model, int own, const double* pts, int N,
                                          int* st, bool exact_pinv, double* dbg) {
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    constexpr int THREADS = WV * WAVE;
    constexpr int u = Model::U, c = Model::C, n = u + c, ld = n + 1;
    const bool owner = wave == own;
    double* sdbg = owner ? dbg : nullptr;                                    // phase stamps of the first iteration (debug entry point)
    double objFunc = 0.0;                                                    // v0' v0, v0 = x0 - x   (:45-46)
    for (int i = tid; i < N; i += THREADS) {
        const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = g.xi[GH_XI * i + k] - x.v[k]; objFunc += d * d; }
    }
    objFunc = block_sum_w<WV>(objFunc, red);
    int it = 0;
    bool expect_truncate = false;                                            // what the previous iteration found (the blocks change little between iterations)
#pragma unroll 1
    for (it = 1; it <= GH_IT_MAX; ++it) {
        if (it == 1) phase_stamp(sdbg, 40);
        if (owner) model.eval(g);                                            // func(xi, ti, yi)   (:50): Tc, D, constraint rows
        __syncthreads();
        if (it == 1) phase_stamp(sdbg, 41);
        double T[27];
        load_uniform27(g.Tc, T);
        // ---- W = B B' (:52): finite check, bound on the largest eigenvalue; see gh_kernel.h for the two pinv paths ----
        double f2max = 0.0;                                                  // max_i |W_i|_F^2
        // Round 5: pinv's tolerance matters only when it can reach the 1e-12 shift, and whether it can is known from max_i |W_i|_F -- a by-product of
        // the blocks the weight pass builds anyway.  So the deflated weight pass runs first, without tolerance, and takes the finite check and the
        // Frobenius maximum along (`spec`); it is repeated with the tolerance only if the maximum says pinv could truncate, and a problem whose
        // previous iteration found that keeps the old order.  The trifocal blocks of normalised image data at N = 200 stay under the limit.
        bool spec = !exact_pinv && !expect_truncate;                         // the check rides on the weight pass
        bool nonfinite = false;
        if (!spec) {
            bool finite = true;
            for (int i = tid; i < N; i += THREADS) {
                double o[6], f[4], B[4][6], W[4][4];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
                tril_block(T, o, f, B);
                block_W(B, W);
                double chk = 0.0, fro2 = 0.0;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) { chk += W[a][b]; fro2 += W[a][b] * W[a][b]; }
                }
                finite = finite && (fabs(chk) <= 1.79e308);
                f2max = (fro2 > f2max) ? fro2 : f2max;
            }
            f2max = block_max_w<WV>(f2max, red);
            if (block_any_w<WV>(!finite, red) || !(f2max <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :53-55
        }
        // blocks in the deflated form (gh_kernel.h, pinv_block_deflated); pinv's tolerance is needed only when it can truncate
        bool may_truncate = spec ? false : !(4.0 * (double)N * eps_of(sqrt(f2max)) < 0.9e-12);
        if (it == 1) phase_stamp(sdbg, 42);
        double tolW = 0.0;
        bool have_tol = false, jacobi = exact_pinv;
        // minimal parameterisations: regular part of pinv(W) in pp, strong direction factored (pinv_block_deflated)
        const bool want_factored = !Model::IDENTITY_D && g.S != nullptr;
        bool factored = false;
#pragma unroll 1
        for (int attempt = 0; attempt < 3; ++attempt) {
            if ((may_truncate || jacobi) && !have_tol) {
                // pinv's tolerance 4N eps(max_i lambda_max(W_i)) needs only the binade of that maximum: when cheap bounds agree on it,
                // the eigenvalue pass that would find the maximum is skipped
                double umax = 0.0, lmax = 0.0;                               // upper / lower bound on max_i lambda_max(W_i)
                for (int i = tid; i < N; i += THREADS) {
                    double o[6], f[4], B[4][6], W[4][4];
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
                    tril_block(T, o, f, B);
                    block_W(B, W);
                    double up, lo;
                    psd_lambda_max_bounds(W, up, lo);
                    umax = (up > umax) ? up : umax;
                    lmax = (lo > lmax) ? lo : lmax;
                }
                umax = block_max_w<WV>(umax, red);
                lmax = block_max_w<WV>(lmax, red);
                double smax = umax;
                if (eps_of(lmax) != eps_of(umax)) {
                    smax = 0.0;
                    for (int i = tid; i < N; i += THREADS) {
                        double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                        for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
                        tril_block(T, o, f, B);
                        block_W(B, W);
                        jacobi4<false>(W, V);
#pragma unroll
                        for (int a = 0; a < 4; ++a) smax = (fabs(W[a][a]) > smax) ? fabs(W[a][a]) : smax;
                    }
                    smax = block_max_w<WV>(smax, red);
                }
                tolW = 4.0 * (double)N * eps_of(smax);
                have_tol = true;
            }
            if (!jacobi) {
                bool bad = false, finite = true;
                constexpr int SLOT = (Model::U * (Model::U + 1) / 2 + Model::U + 1) & ~1;
                double* slot = nullptr;
                if (want_factored) {                                         // per-wavefront partial sums of the strong-direction terms
                    slot = (wave < WV - 1) ? g.G + wave * SLOT : g.S;        // Ghat's slot is not yet in use; 3 SLOT <= 729
                    for (int e = lane; e < SLOT; e += WAVE) slot[e] = 0.0;
                }
#pragma unroll 1
                for (int base = 0; base < N; base += THREADS) {        // block-uniform trip count (the butterflies need whole wavefronts)
                    const int i = base + tid;
                    double bv[Model::U], tv = 0.0;
#pragma unroll
                    for (int k = 0; k < Model::U; ++k) bv[k] = 0.0;
                    if (i < N) {
                        double o[6], f[4], B[4][6], W[4][4], Wp[10];
#pragma unroll
                        for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
                        tril_block(T, o, f, B);
                        block_W(B, W);
                        if (spec) {
                            double chk = 0.0, fro2 = 0.0;
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
#pragma unroll
                                for (int b = 0; b < 4; ++b) { chk += W[a][b]; fro2 += W[a][b] * W[a][b]; }
                            }
                            finite = finite && (fabs(chk) <= 1.79e308);
                            f2max = (fro2 > f2max) ? fro2 : f2max;
                        }
                        double nn[4], cs = 0.0;
                        const bool ok = want_factored ? pinv_block_deflated<true>(B, W, tolW, Wp, nn, &cs) : pinv_block_deflated<false>(B, W, tolW, Wp, nn, &cs);
                        bad = !ok || bad;
#pragma unroll
                        for (int a = 0; a < 4; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
#pragma unroll
                        for (int k = 0; k < 10; ++k) g.pp[GH_PP * i + k] = Wp[k];
                        if constexpr (!Model::IDENTITY_D) if (want_factored && ok) {
                            double* sn = g.pp + GH_PP * (long)N + GH_SN * (long)i;
#pragma unroll
                            for (int k = 0; k < 4; ++k) sn[k] = nn[k];
                            sn[4] = cs;
                            // b = sqrt(cs) D' (Ap' n),  t = sqrt(cs) n'w,  n'w = -n'f - (B'n) . (x - xi)
                            double gm[3][3];
                            tril_grad_n(o, nn, gm);
                            const double h1[3] = {o[0], o[1], 1.0};
                            strong_apply_Dt<Model>(model, g, h1, gm, bv);
                            const double sc = sqrt(cs);
#pragma unroll
                            for (int k = 0; k < Model::U; ++k) bv[k] *= sc;
                            const Pt6 x = premap(load_pt(pts, i), w->nrm);
                            double nw = -(nn[0] * f[0] + nn[1] * f[1] + nn[2] * f[2] + nn[3] * f[3]);
#pragma unroll
                            for (int k = 0; k < 6; ++k) nw -= (B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3]) * (x.v[k] - o[k]);
                            tv = sc * nw;
                        }
                    }
                    if constexpr (!Model::IDENTITY_D) { if (want_factored) strong_accumulate<Model::U>(bv, tv, slot); }
                }
                if (want_factored) {
                    __syncthreads();
                    constexpr int total = Model::U * (Model::U + 1) / 2 + Model::U;
                    for (int e = tid; e < total; e += THREADS)
                        g.S[e] = (WV == 4) ? (g.G[e] + g.G[SLOT + e]) + (g.G[2 * SLOT + e] + g.S[e]) : g.G[e] + g.S[e];
                }
                if (spec) {
                    f2max = block_max_w<WV>(f2max, red);
                    if (block_any_w<WV>(!finite, red) || !(f2max <= 1.79e308)) { nonfinite = true; break; }   // :53-55
                    spec = false;
                    may_truncate = !(4.0 * (double)N * eps_of(sqrt(f2max)) < 0.9e-12);
                    if (may_truncate) continue;
                }
                if (!block_any_w<WV>(bad, red)) { factored = want_factored; break; }
                jacobi = true;                                               // a block without the structure: eigen-decompositions for all
                continue;
            }
            for (int i = tid; i < N; i += THREADS) {
                double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
                tril_block(T, o, f, B);
                block_W(B, W);
                jacobi4<true>(W, V);
                double inv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) inv[a] = (W[a][a] > tolW) ? 1.0 / W[a][a] : 0.0;
                double Wp[10];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b)
                        Wp[a * (a + 1) / 2 + b] = V[a][0] * inv[0] * V[b][0] + V[a][1] * inv[1] * V[b][1] + V[a][2] * inv[2] * V[b][2]
                                                  + V[a][3] * inv[3] * V[b][3] + ((a == b) ? 1e-12 : 0.0);
#pragma unroll
                for (int k = 0; k < 10; ++k) g.pp[GH_PP * i + k] = Wp[k];
            }
            break;
        }
        if (nonfinite) { *st = ST_NONFINITE; break; }
        expect_truncate = may_truncate;
        if (it == 1) phase_stamp(sdbg, 43);
        // ---- Ghat, ghat: the ten sweeps are dealt to the four wavefronts, each sweep runs over ALL correspondences on one wavefront
        //      (4 per lane at N = 200) and ends in one reduce-scatter: a quarter of the reductions of the per-wavefront-partial layout
        //      and no combine step ----
        __syncthreads();                                                     // xi, W+ of every correspondence are in place
        if constexpr (WV == 4) {
            if (wave == 0) { gh_sweep_part<0>(g, g.H, N, T, w, pts); gh_sweep_part<4>(g, g.H, N, T, w, pts); gh_sweep_part<8>(g, g.H, N, T, w, pts); }
            else if (wave == 1) { gh_sweep_part<1>(g, g.H, N, T, w, pts); gh_sweep_part<5>(g, g.H, N, T, w, pts); gh_sweep_part<7>(g, g.H, N, T, w, pts); }
            else if (wave == 2) { gh_sweep_part<2>(g, g.H, N, T, w, pts); gh_sweep_part<6>(g, g.H, N, T, w, pts); }
            else { gh_sweep_part<3>(g, g.H, N, T, w, pts); gh_sweep_part<9>(g, g.H, N, T, w, pts); }   // 9: the right-hand side, recomputes w
        } else {                                                             // two wavefronts: five sweeps each
            if (wave == 0) { gh_sweep_part<0>(g, g.H, N, T, w, pts); gh_sweep_part<4>(g, g.H, N, T, w, pts); gh_sweep_part<8>(g, g.H, N, T, w, pts); gh_sweep_part<2>(g, g.H, N, T, w, pts); gh_sweep_part<6>(g, g.H, N, T, w, pts); }
            else { gh_sweep_part<1>(g, g.H, N, T, w, pts); gh_sweep_part<5>(g, g.H, N, T, w, pts); gh_sweep_part<7>(g, g.H, N, T, w, pts); gh_sweep_part<3>(g, g.H, N, T, w, pts); gh_sweep_part<9>(g, g.H, N, T, w, pts); }
        }
        __syncthreads();
        if (it == 1) phase_stamp(sdbg, 44);
        for (int e = tid; e < 729; e += THREADS) {                     // Ghat[(q,i1),(q',i1')] = H[6 tri(q,q') + hht(i1,i1')]
            const int r = e / 27, cc = e % 27;
            const int q = r % 9, i1 = r / 9, qq = cc % 9, i1p = cc / 9;
            const int hi = (q > qq) ? q : qq, lo = (q > qq) ? qq : q;
            g.G[e] = g.H[6 * (hi * (hi + 1) / 2 + lo) + hht_index(i1, i1p)];
        }
        __syncthreads();
        if (Model::IDENTITY_D) {                                             // A = Ap: A'WA = Ghat, A'Ww = ghat
            for (int e = tid; e < 729 + 27; e += THREADS) {
                if (e < 729) g.M[(e / 27) * ld + e % 27] = g.G[e] + ((e / 27 == e % 27) ? 1e-12 : 0.0);
                else g.M[(e - 729) * ld + n] = g.H[270 + e - 729];
            }
        } else {
            // (Summing over the <= 9 non-zeros per column of Ressl's D instead -- 5 k multiply-adds for the two products, not 25 k -- was
            // measured SLOWER, 14.2 k -> 23.8 k cycles: the index arithmetic and the irregular LDS addresses cost more than the dense,
            // perfectly regular loops save.)
            if constexpr (u % 4 == 0) {                                      // (Ressl, u = 20; Nordberg's u = 19 loses the 16-byte alignment: measured 2.5 % slower)
                // Four adjacent columns per thread: one read of Ghat(r,k) (D(k,pr)) and two 16-byte reads of D (Y) feed four multiply-adds, where a
                // thread per entry made two LDS reads per multiply-add -- these two products were 14 % (N = 200) to 22 % (N = 60) of an iteration,
                // LDS-issue-bound.  Every entry is still the same 27-term sum in the same order.
                constexpr int UQ = (u + 3) / 4;
                for (int e = tid; e < 27 * UQ; e += THREADS) {             // Y = Ghat D
                    const int r = e / UQ, c0 = 4 * (e % UQ);
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll 3
                    for (int k = 0; k < 27; ++k) {                         // (fully unrolled the 135 loads are hoisted and spill 80 registers)
                        const double gv = g.G[r * 27 + k];
    #pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] += gv * g.D[k * u + ((c0 + j < u) ? c0 + j : 0)];
                    }
    #pragma unroll
                    for (int j = 0; j < 4; ++j) if (c0 + j < u) g.Y[r * u + c0 + j] = acc[j];
                }
                __syncthreads();
                for (int e = tid; e < u * UQ + u; e += THREADS) {          // M = [D'Y + 1e-12 I ...], b = [D' ghat; -g]
                    if (e < u * UQ) {
                        const int pr = e / UQ, c0 = 4 * (e % UQ);
                        double acc[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll 3
                        for (int k = 0; k < 27; ++k) {
                            const double dv = g.D[k * u + pr];
    #pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] += dv * g.Y[k * u + ((c0 + j < u) ? c0 + j : 0)];
                        }
    #pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int pc = c0 + j;
                            if (pc < u) {
                                double v = acc[j];
                                if (factored) v += g.S[(pr >= pc) ? tri_index(pr, pc) : tri_index(pc, pr)];
                                g.M[pr * ld + pc] = v + ((pr == pc) ? 1e-12 : 0.0);
                            }
                        }
                    } else {
                        const int pc = e - u * UQ;
                        double acc = 0.0;
                        for (int k = 0; k < 27; ++k) acc += g.D[k * u + pc] * g.H[270 + k];
                        if (factored) acc += g.S[u * (u + 1) / 2 + pc];
                        g.M[pc * ld + n] = acc;
                    }
                }
            } else {
                for (int e = tid; e < 27 * u; e += THREADS) {              // Y = Ghat D
                    const int r = e / u, pcol = e % u;
                    double acc = 0.0;
                    for (int k = 0; k < 27; ++k) acc += g.G[r * 27 + k] * g.D[k * u + pcol];
                    g.Y[e] = acc;
                }
                __syncthreads();
                for (int e = tid; e < u * u + u; e += THREADS) {           // M = [D'Y + 1e-12 I ...], b = [D' ghat; -g]
                    const int pr = e / u, pc = e % u;
                    double acc = 0.0;
                    if (e < u * u) {
                        for (int k = 0; k < 27; ++k) acc += g.D[k * u + pr] * g.Y[k * u + pc];
                        if (factored) acc += g.S[(pr >= pc) ? tri_index(pr, pc) : tri_index(pc, pr)];
                        g.M[pr * ld + pc] = acc + ((pr == pc) ? 1e-12 : 0.0);
                    } else {
                        for (int k = 0; k < 27; ++k) acc += g.D[k * u + pc] * g.H[270 + k];
                        if (factored) acc += g.S[u * (u + 1) / 2 + pc];
                        g.M[pc * ld + n] = acc;
                    }
                }
            }
        }
        if (tid < c) g.M[(u + tid) * ld + u + tid] = 1e-12;
        __syncthreads();
        double chkM = 0.0;
        for (int e = tid; e < n * ld; e += THREADS) chkM += g.M[e];
        if (!(fabs(block_sum_w<WV>(chkM, red)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        if (it == 1) phase_stamp(sdbg, 45);
        // aux = pinv(M + 1e-12 I) * b   (:67): truncated pseudo-inverse by the workgroup, or Gauss-Jordan on the owner wavefront
        if (Model::REDUNDANT_CONSTRAINTS) block_pinv_solve_sym<n>(g.M, g.V, g.dt, g.V + n * n, own);
        if (owner) {
            bool ok = true;
            if (!Model::REDUNDANT_CONSTRAINTS) ok = wave_solve_gj<n>(g.M, g.dt);
            wave_sync();
            if (lane < 27) {                                                 // dT = D dt
                double acc = 0.0;
                if (Model::IDENTITY_D) acc = g.dt[lane];
                else for (int k = 0; k < u; ++k) acc += g.D[lane * u + k] * g.dt[k];
                g.dT[lane] = acc;
            }
            if (lane == 0) red[8] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        if (!Model::REDUNDANT_CONSTRAINTS && red[8] == 0.0) {
            // numerically singular KKT matrix (degenerate geometry): pinv truncates, so does the eigen-decomposition path.
            // Eigenvectors go to Y and the scratch vectors to H, both dead by now.
            block_pinv_solve_sym<n>(g.M, g.Y, g.dt, g.H, own);
            if (owner && lane < 27) {
                double acc = 0.0;
                if (Model::IDENTITY_D) acc = g.dt[lane];
                else for (int k = 0; k < u; ++k) acc += g.D[lane * u + k] * g.dt[k];
                g.dT[lane] = acc;
            }
            __syncthreads();
        }
        double dTr[27];
        load_uniform27(g.dT, dTr);
        if (it == 1) phase_stamp(sdbg, 46);
        // ---- v = -B' W+ (A dt - w)   (:69) ----
        double obj = 0.0, diff = 0.0;
        for (int i = tid; i < N; i += THREADS) {
            double o[6], f[4], B[4][6], Ad[4];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[GH_XI * i + k];
            tril_block(T, o, f, B);
            {
                double m[3][3], t1[3][3], t2[3][3];
                tril_slices(dTr, o, m, t1, t2);
                tril_quad(m, o[2], o[3], o[4], o[5], Ad);                    // Ap_i (D dt)
            }
            double Wp[10], r[4], wv[4];
#pragma unroll
            for (int k = 0; k < 10; ++k) Wp[k] = g.pp[GH_PP * i + k];
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double sw = -f[a];
#pragma unroll
                for (int k = 0; k < 6; ++k) sw -= B[a][k] * (x.v[k] - o[k]);
                wv[a] = Ad[a] - sw;                                          // A dt - w
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
                r[a] = wp_at(Wp, a, 0) * wv[0] + wp_at(Wp, a, 1) * wv[1] + wp_at(Wp, a, 2) * wv[2] + wp_at(Wp, a, 3) * wv[3];
            double bn[6] = {0, 0, 0, 0, 0, 0}, sterm = 0.0;                  // strong direction: -cs (B'n) n'(A dt - w)
            if (factored) {
                const double* sn = g.pp + GH_PP * (long)N + GH_SN * (long)i;
                const double nn[4] = {sn[0], sn[1], sn[2], sn[3]};
                const double cs = sn[4];
#pragma unroll
                for (int k = 0; k < 6; ++k) bn[k] = B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3];
                sterm = cs * (nn[0] * wv[0] + nn[1] * wv[1] + nn[2] * wv[2] + nn[3] * wv[3]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double v = -(B[0][k] * r[0] + B[1][k] * r[1] + B[2][k] * r[2] + B[3][k] * r[3]) - bn[k] * sterm;
                g.pp[GH_PP * i + k] = v;
                obj += v * v;
                const double d = o[k] - x.v[k] - v;
                diff += d * d;
            }
        }
        obj = block_sum_w<WV>(obj, red);
        diff = block_sum_w<WV>(diff, red);
        if (it == 1) phase_stamp(sdbg, 47);
        double ndt2 = 0.0;
        for (int k = 0; k < u; ++k) ndt2 += g.dt[k] * g.dt[k];               // same order on every thread
        if (sqrt(ndt2) < GH_TOL && sqrt(diff) < GH_TOL) break;               // :71-73 (dy is empty)
        if (obj > objFunc) break;                                            // :75-76, factor = 1
        objFunc = obj;                                                       // :78
        for (int i = tid; i < N; i += THREADS) {                       // xi = x + v; ti = ti + dt   (:80)
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) g.xi[GH_XI * i + k] = x.v[k] + g.pp[GH_PP * i + k];
        }
        if (tid < u) g.p[tid] += g.dt[tid];
        __syncthreads();
    }
    __syncthreads();
    return (it > GH_IT_MAX) ? GH_IT_MAX : it;                                // :82
}

// Wavefronts per workgroup (round 4): TWO for the models whose KKT system is solved by pivoted elimination on one wavefront (Ressl, Nordberg) --
// four workgroups per CU at 256 registers per thread instead of two (Ressl) or three with 200 registers spilled (Nordberg): what overlaps the
// wave-serial steps is the number of workgroups per CU (pi_wg_kernel.h).  Four for FaugPapa's generic kernel (the fall-back of gh_fp_kernel.h).
template <class Model> struct gh_wg_waves { static constexpr int value = Model::REDUNDANT_CONSTRAINTS ? 4 : 2; };
template <class Model> struct gh_wg_per_cu { static constexpr int value = Model::REDUNDANT_CONSTRAINTS ? Model::WG_PER_CU : 4; };
template <class Model>
__global__ void __launch_bounds__(gh_wg_waves<Model>::value * WAVE, gh_wg_per_cu<Model>::value * gh_wg_waves<Model>::value / 4) k_gh_block(const GhWgArgs a) {   // (second argument: wavefronts per SIMD)
    constexpr int WV = gh_wg_waves<Model>::value;
    static_assert(!Model::REDUNDANT_CONSTRAINTS || (Model::IDENTITY_D && 2 * 27 * Model::U + 298 >= (Model::U + Model::C) * (Model::U + Model::C + 2)),
                  "pseudo-inverse workspace must fit D | H | Y");
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    double* ghbase = smem + base;
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        // block-uniform: too few points / unresolved; FLAG_ONLY_RETRY: the triplets a specialised block kernel handed over (gh_fp_kernel.h)
        if ((a.flags & FLAG_ONLY_RETRY) ? (a.status[b] != ST_RETRY) : (a.status[b] != ST_OK)) continue;
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        double* red;
        GhWork g = gh_wg_carve(ghbase, w, Model::U, Model::C, a.spill ? 0 : N, Model::REDUNDANT_CONSTRAINTS, &red);
        if (a.spill) {
            double* slice = a.spill + blockIdx.x * a.spill_stride;
            // FLAG_XI_IN_LDS: xi (read by every pass) behind the fixed part in LDS, W+ (10 N, the bigger half) in the slice -- when that still fits
            // the workgroups per CU the launcher wants (Ressl at N = 200: 38.5 KB, four per CU)
            g.xi = (a.flags & FLAG_XI_IN_LDS) ? ghbase + gh_wg_lds_doubles(Model::U, Model::C, 0, Model::REDUNDANT_CONSTRAINTS) : slice;
            g.pp = slice + GH_XI * (long)N;
        }
        const int own = pick_serial_wave_w<WV>(red);
        const double* r = a.rec + b * GH_REC_DOUBLES;
        if (tid < 27) w->t[tid] = r[tid];
        if (tid < 18) w->pa[tid] = r[27 + tid];
        if (tid < 6) w->epi[tid] = r[45 + tid];
        if (tid < 9) w->nrm[tid] = r[51 + tid];
        __syncthreads();
        Model model;
        if constexpr (gh_has_preinit<Model>::value) model.pre = a.init_rec ? a.init_rec + b * Model::PRE_DOUBLES : nullptr;
        double* sdbg = (a.dbg && wave == own) ? a.dbg + b * DBG_STRIDE : nullptr;
        phase_stamp(sdbg, 36);
        if (wave == own) {                                                   // initial parameters; cameras P1, P2, P3 of the linear solution
            model.init(w, g);
            if (lane == 0) model.share(red + 10);
        }
        __syncthreads();
        if (wave != own) model.adopt(red + 10);
        phase_stamp(sdbg, 37);
        gh_block_reproject<WV>(w, pts, N, g.xi);
        __syncthreads();
        phase_stamp(sdbg, 38);
        int gst = ST_OK;
        const int iters = gauss_helmert_block<Model, WV>(w, g, red, model, own, pts, N, &gst, (a.flags & FLAG_GH_EXACT) != 0,
                                                       a.dbg ? a.dbg + b * DBG_STRIDE : nullptr);
        phase_stamp(sdbg, 39);
        if (wave == own) {
            model.eval(g);                                                   // T from p_opt   (:87-94)
            if (lane < 27) a.topt[b * 27 + lane] = g.Tc[lane];
            if (lane == 0) {
                if (a.iter) a.iter[b] = iters;
                int s = gst;
                if (gh_model_bad(model)) s = ST_RANK;
                if (s != ST_OK) a.status[b] = -s;                            // negative: reported after k_gh_finish has produced the outputs
                else if (a.flags & FLAG_ONLY_RETRY) a.status[b] = ST_OK;
            }
        }
    }
}

// transform_TFT + R_t_from_TFT + Reconst for the optimised tensor   (ResslTFTPoseEstimation.m:96-103)
__global__ void __launch_bounds__(64, 2) k_gh_finish(const GhWgArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    const int lane = lane_id();
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        wave_sync();
        const int s0 = a.status[b];
        if (s0 > 0) {                                                        // ST_TOO_FEW (or an unresolved retry): no outputs
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
            continue;
        }
        if (lane < 27) { w->calm[lane] = a.calm[b * a.calm_stride + lane]; w->t[lane] = a.topt[b * 27 + lane]; }
        if (lane < 9) w->nrm[lane] = a.rec[b * GH_REC_DOUBLES + 51 + lane];
        wave_sync();
        transform_tft_inverse(w->t, w->T1, w->Lp, [w](int v) { return normal_matrix(w->nrm, v); });
        // fast tiers first (certified votes, inverse-iteration null vectors and DLT points: what k_linear_tft_pose<false> runs); the exact
        // tiers redo the triplet only when one of them could not finish or certify its part (wave-uniform)
        bool fine = true;
        int status = rt_from_tft_wave<false>(w, pts, N, nullptr, &fine);
        if (!fine) status = rt_from_tft_wave<true>(w, pts, N, nullptr);
        if (s0 < 0) status = -s0;
        double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
        const bool bad = !(fabs(chk) <= 1.79e308);
        if (wave_any(bad)) {                                                 // non-finite outputs: status 2 (unless the iteration reported first), ALL outputs NaN
            if (status == ST_OK) status = ST_NONFINITE;
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        } else {
            write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
            if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
            if (a.reconst) final_reconst(w, pts, N, a.reconst + b * 3 * (long)N);
        }
        if (lane == 0) a.status[b] = status;
    }
}

}  // namespace tff
