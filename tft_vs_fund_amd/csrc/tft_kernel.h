// LinearTFTPoseEstimation as one fused kernel: one wavefront per triplet.
//
//   Normalize2Ddata x3  ->  linearTFT (moment-form Gram matrix, smallest
//   eigenvector, epipoles, constrained re-solve in the 15-dim range of E)
//   ->  transform_TFT  ->  R_t_from_TFT (calibrate, epipoles, E21/E31,
//   recover_R_t with the cheirality vote, t3 scale)  ->  optional Reconst.
//
// Reference: TFT_methods/LinearTFTPoseEstimation.m:44-62, linearTFT.m:33-91,
// transform_TFT.m:42-49, R_t_from_TFT.m:40-106, triangulation3D.m:51-63.
//
// The 4N x 27 design matrix A of linearTFT.m:36-62 is never formed.  Its rows
// are Kronecker products h1 (x) c3 (x) c2 with h1 = (x1,y1,1),
// c2 in {(1,0,-x2),(0,1,-y2)}, c3 likewise, so
//     G = A'A = sum_n (h1 h1') (x) (C3'C3) (x) (C2'C2)
// has only 6 x 4 x 4 = 96 distinct entries: products of the monomials
//   p1 = {x1^2, x1 y1, x1, y1^2, y1, 1},  q3 = {1, x3, y3, x3^2+y3^2},  q2 likewise.
// With M_h the 4x4 moment block of p1-monomial h, every quadratic form of G is a
// 4x4 bilinear form:  (u (x) a3 (x) a2)' G (v (x) b3 (x) b2)
//     = sum_{i,i'} u_i v_i'  c3' M_{h(i,i')} c2,
//   c2 = (a2_0 b2_0 + a2_1 b2_1, -(a2_0 b2_2 + a2_2 b2_0), -(a2_1 b2_2 + a2_2 b2_1), a2_2 b2_2), c3 likewise.
// The wave accumulates the 96 sums (one correspondence per lane, halving
// butterfly across lanes); lane r then builds row r of G in registers and the
// wave takes the eigenvector of the smallest eigenvalue (== V(:,end) of
// svd(A), :64-67) with wave_min_eigvec_reg<27>.  G itself never exists in memory.
//
// Constrained re-solve (:82-91): range(E) = { T : Q2' T_i Q3 has a zero lower
// right 2x2 block }, with Q2 = [e21 | complement], Q3 = [e31 | complement]
// orthonormal.  An orthonormal basis Up of range(E) is therefore 15 columns of
// I3 (x) Q3 (x) Q2; (A Up)'(A Up) = Up' G Up is a 15x15 matrix whose entries are
// again 4x4 bilinear forms of the moments -- no second pass over the data.
// rank(E) = 15 always (e21, e31 are unit vectors).
//
// Two instantiations:
//   k_linear_tft_pose<false>  the fast tiers only: Gram matrix + Cholesky inverse iteration, certified sign-only votes, DLT
//       points by inverse iteration.  Whatever a fast tier cannot finish or certify (iteration cap, a Gram matrix whose
//       eigenvector would lose more than ~1e-9 to the squared conditioning, an uncertified cheirality sign) marks the
//       triplet ST_RETRY;
//   k_linear_tft_pose<true>   the exact kernel: streaming Householder QR of the explicit 4N x 27 system (wave_qr.h), the
//       4N x 15 re-solve from R * Up, one-sided Jacobi fall-backs everywhere -- the accuracy of the reference's svd() calls
//       whatever the gaps.  Runs over the ST_RETRY triplets, for whole batches of minimal samples (N < EXACT_BELOW_N), and
//       for everything with TFF_OPT_SOLVER = 1.
#pragma once
#include "pose_common.h"

namespace tff {

struct LinearTftArgs {
    const double* corresp;   // B x (6 x N), column-major per triplet
    const double* calm;      // B x 27 (9x3 column-major) or 27 shared
    long calm_stride;        // 27 or 0
    long B;
    int N;
    int flags;
    double* Rt2;             // B x 12 (3x4 column-major)
    double* Rt3;             // B x 12
    double* T;               // B x 27
    double* reconst;         // B x 3N or null
    int* iter;               // B or null
    int* status;             // B (never null inside the library: the context supplies one)
    double* dbg;             // B x DBG_STRIDE or null
    const int* sample_idx;   // null, or B x N int32 indices into ONE shared scene at `corresp` (config 4: minimal samples)
    double* init_p;          // null, or B x 27: initial parameters of the Pi methods (debug / building-block output)
    double* init_x;          // with init_p: B x 6N initial observation estimates
    double* spill;           // null, or global workspace for the per-correspondence state of the iterative methods when it does not
    long spill_stride;       //   fit the 160 KB of LDS (large N): gridDim.x blocks of spill_stride doubles
    int sample_ns;           // with sample_idx: number of correspondences in the shared scene (indices outside [0, sample_ns) -> ST_TOO_FEW)
    const double* pre;       // null, or B x PRE_DOUBLES: moment sums and normalisations from k_tft_moments (tft_moments_kernel.h; the row kernels' <true> variants)
    // The exact kernels as the fix-up of a row kernel: the triplets to redo arrive as a compact list instead of a scan of the status array --
    // retry_list[0 .. *retry_count).  The ROW kernels append to it themselves (one atomic per flagged triplet, rows_publish_status) and zero
    // *retry_zero, the counter of the context's NEXT call (two counters alternate: the one in use was zeroed during the previous call).
    int* retry_list;
    int* retry_count;
    int* retry_zero;
};

// Inverse-iteration cap before a triplet is handed to the Jacobi fix-up pass: 300 iterations (~0.13 M
// instructions) are still ~10x cheaper than the LDS Jacobi, and cover spectral-gap ratios up to ~0.9
// (minimal 7-point samples, samples with outliers); typical well-posed triplets use 4.
constexpr int EIG_MAXIT = 300;
// entries of the retry list a row kernel hands to the exact kernel: index | hints << 28 (bit 0: the 27-column inverse iteration hit its cap,
// bit 1: the 15-column one did); batches of 2^28 triplets and more go without the list (capi.hip)
constexpr int RETRY_HINT_SHIFT = 28;
constexpr int RETRY_INDEX_MASK = (1 << RETRY_HINT_SHIFT) - 1;

// c-vector of a pair of 3-vectors (see header): bilinear weights of the q-monomials
__device__ __forceinline__ void cvec(const double* a, const double* b, double (&c)[4]) {
    c[0] = a[0] * b[0] + a[1] * b[1];
    c[1] = -(a[0] * b[2] + a[2] * b[0]);
    c[2] = -(a[1] * b[2] + a[2] * b[1]);
    c[3] = a[2] * b[2];
}
__device__ __forceinline__ int hht_index(int i, int ip) {          // index into p1 of entry (i,i') of h1 h1'
    const int a = (i < ip) ? i : ip, b = (i < ip) ? ip : i;
    return (a == 0) ? b : ((a == 1) ? 2 + b : 5);
}
__device__ __forceinline__ double bilinear44(const double* M, const double (&c3)[4], const double (&c2)[4]) {
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        acc += c3[p] * (M[4 * p] * c2[0] + M[4 * p + 1] * c2[1] + M[4 * p + 2] * c2[2] + M[4 * p + 3] * c2[3]);
    return acc;
}

// 96 moment sums -> w->mom.  Three passes of 32 accumulators per lane.
__device__ inline void accumulate_moments(PoseLds* w, const double* pts, int N) {
    const int lane = lane_id();
    const double s1 = w->nrm[0], ox1 = w->nrm[1], oy1 = w->nrm[2];
    const double s2 = w->nrm[3], ox2 = w->nrm[4], oy2 = w->nrm[5];
    const double s3 = w->nrm[6], ox3 = w->nrm[7], oy3 = w->nrm[8];
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        double acc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = 0.0;
        for (int i = lane; i < N; i += WAVE) {
            const Pt6 p = load_pt(pts, i);
            // new_points = N_matrix(1:2,:) * [points; 1]   (Normalize2Ddata.m:39)
            const double x1 = s1 * p.v[0] + ox1, y1 = s1 * p.v[1] + oy1;
            const double x2 = s2 * p.v[2] + ox2, y2 = s2 * p.v[3] + oy2;
            const double x3 = s3 * p.v[4] + ox3, y3 = s3 * p.v[5] + oy3;
            const double q2[4] = {1.0, x2, y2, x2 * x2 + y2 * y2};
            const double q3[4] = {1.0, x3, y3, x3 * x3 + y3 * y3};
            const double pa = (pass == 0) ? x1 * x1 : ((pass == 1) ? x1 : y1);
            const double pb = (pass == 0) ? x1 * y1 : ((pass == 1) ? y1 * y1 : 1.0);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double wv = q3[b] * q2[c];
                    acc[4 * b + c] += pa * wv;
                    acc[16 + 4 * b + c] += pb * wv;
                }
        }
        const double tot = wave_reduce_scatter<32>(acc);
        if ((lane & 1) == 0) w->mom[32 * pass + reduce32_index(lane)] = tot;
    }
    wave_sync();
}

// sign and q-monomial index of entry (j,j') of C'C,  C = [1 0 -x; 0 1 -y]:
// C'C = [1 0 -x; 0 1 -y; -x -y x^2+y^2]  <->  q-indices [0 . 1; . 0 2; 1 2 3]
__device__ __forceinline__ void ctc_entry(int j, int jp, double& sign, int& idx) {
    const int a = (j < jp) ? j : jp, b = (j < jp) ? jp : j;
    if (b < 2) { sign = (a == b) ? 1.0 : 0.0; idx = 0; }
    else if (a < 2) { sign = -1.0; idx = 1 + a; }
    else { sign = 1.0; idx = 3; }
}
// Row r = (j,k,i) of G: entry (j',k',i') = s2(j,j') s3(k,k') mom[h(i,i')][i3(k,k')][i2(j,j')].
__device__ __forceinline__ void gram_row27(const double* mom, int r, double (&g)[27], double& diag) {
    const int i = r / 9, k = (r % 9) / 3, j = r % 3;
    double s2[3], s3[3];
    int i2[3], i3[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { ctc_entry(j, q, s2[q], i2[q]); ctc_entry(k, q, s3[q], i3[q]); }
#pragma unroll
    for (int ip = 0; ip < 3; ++ip) {
        const double* M = mom + 16 * hht_index(i, ip);
#pragma unroll
        for (int kp = 0; kp < 3; ++kp)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) g[jp + 3 * kp + 9 * ip] = s2[jp] * s3[kp] * M[4 * i3[kp] + i2[jp]];
    }
    diag = 0.0;
#pragma unroll
    for (int c = 0; c < 27; ++c) diag = (c == r) ? g[c] : diag;
}

// orthonormal frame [e | q | q'] of a unit vector e, row-major Q[3*r + c]
__device__ __forceinline__ void frame_of(const double* e, double* Q) {
    const double a0 = fabs(e[0]), a1 = fabs(e[1]), a2 = fabs(e[2]);
    double ax[3] = {0.0, 0.0, 0.0};
    if (a0 <= a1 && a0 <= a2) ax[0] = 1.0; else if (a1 <= a2) ax[1] = 1.0; else ax[2] = 1.0;
    double q1[3], q2[3];
    cross3(e, ax, q1);
    const double n1 = rsqrt(q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2]);
    q1[0] *= n1; q1[1] *= n1; q1[2] *= n1;
    cross3(e, q1, q2);
    const double n2 = rsqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { Q[3 * r] = e[r]; Q[3 * r + 1] = q1[r]; Q[3 * r + 2] = q2[r] * n2; }
}

// column a = (i, m) of Up: the 3-vectors (Q2 column jj, Q3 column kk), (jj,kk) = m<3 ? (0,m) : (m-2,0)
__device__ __forceinline__ void up_factors(const double* Q, int m, double (&a2)[3], double (&a3)[3]) {
    const int jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { a2[q] = Q[3 * q + jj]; a3[q] = Q[9 + 3 * q + kk]; }
}

// Row e (0..3) of correspondence (x1,y1,x2,y2,x3,y3) in the 4N x 27 system of linearTFT.m:51-62 is
//   A(e, j + 3k + 9i) = h1[i] c3[k] c2[j],  h1 = (x1,y1,1),  c2 = (1,0,-x2) | (0,1,-y2),  c3 = (1,0,-x3) | (0,1,-y3).
// R of the QR factorisation of the 4N x 27 system (normalised correspondences), left in Rl (27 x 27 LDS, row-major).
// Column layout (wave_qr_cols_append): lane c < 27 owns column c = j + 3k + 9i and builds its entry of the four rows of each
// correspondence from the formula above; seven correspondences (28 rows) per chunk -- a minimal sample is one chunk.
__device__ inline void tft_system_qr(const double* pts, const int N, const double* nrm, double* Rl) {
    const int lane = lane_id();
    for (int e = lane; e < 27 * 27; e += WAVE) Rl[e] = 0.0;
    wave_sync();
    const int col = (lane < 27) ? lane : 0, ci = col / 9, ck = (col % 9) / 3, cj = col % 3;
#pragma unroll 1
    for (int base = 0; base < N; base += 7) {
        double a[28];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int i = base + q;                                          // wave-uniform: every lane reads the same point
            const Pt6 p = premap(load_pt(pts, (i < N) ? i : 0), nrm);
            const double h = (i < N && lane < 27) ? ((ci == 0) ? p.v[0] : (ci == 1) ? p.v[1] : 1.0) : 0.0;
            const double c2x = (cj == 0) ? 1.0 : (cj == 1) ? 0.0 : -p.v[2], c2y = (cj == 0) ? 0.0 : (cj == 1) ? 1.0 : -p.v[3];
            const double c3x = (ck == 0) ? 1.0 : (ck == 1) ? 0.0 : -p.v[4], c3y = (ck == 0) ? 0.0 : (ck == 1) ? 1.0 : -p.v[5];
            const double hx = h * c3x, hy = h * c3y;
            a[4 * q + 0] = hx * c2x; a[4 * q + 1] = hx * c2y; a[4 * q + 2] = hy * c2x; a[4 * q + 3] = hy * c2y;
        }
        wave_qr_cols_append<27, 28>(a, Rl);
    }
}

// linearTFT.m:64-91: everything of linearTFT after the data pass, for the lane group G (the whole wavefront, or one half of it
// working on its own triplet).  EXACT = false works from the 96 moment sums in w->mom; EXACT = true from the explicit system.
// Leaves the constrained tensor in w->t, the epipoles in w->epi and (if want_P) linearTFT's `a`
// (P2 = [reshape(a(1:9),3,3) e21], P3 = [reshape(a(10:18),3,3) e31]) in w->pa.
// Returns false when a fast tier could not finish (EXACT = false only): eigen-solve not converged or at risk, null vector capped.
// |G| / (lambda_(n-1) - lambda_n) beyond which the Gram eigenvector is not trusted: 1e7 (error ~ 1e-16 x this), the default limit of
// wave_min_eigvec_reg's gram_risk flag
template <bool JAC, int G>
__device__ inline bool linear_tft_middle(PoseLds* w, JacobiLds* jw, const double* pts, int N, bool want_P, double* dbg, const int hint = 0) {
    static_assert(!JAC || G == 64, "the exact solver works on whole wavefronts");
    using Grp = Group<G>;
    const int lane = Grp::lane();
    const int wl = Grp::index() * G;                                         // first lane of this group (stamp writer)
    int it1 = 0, it2 = 0;
    bool ok = true;
    {                                                                        // :64-67
        double x;
        if (JAC) {
            tft_system_qr(pts, N, w->nrm, jw->A);                            // R of the 4N x 27 system; survives in jw->A
            phase_stamp(dbg, 3, wl);
            double g[27];
            wave_qr_rows_from_lds<27>(jw->A, g);
            // (hint bit 0: the row kernel that handed this triplet on has run THIS inverse iteration into its cap already -- tft_rows_exact_kernel.h:
            // same matrix, same rate -- so the one-sided Jacobi takes over after a single step instead of 300)
            x = wave_qr_min_rsv<27>(g, jw->A, w->Lp, w->Lp, (hint & 1) ? 0 : EIG_MAXIT, &it1);
            if (it1 >= 1000) {                                               // the fall-back rotated R away: factor again (rare)
                if (lane < 27) w->t[lane] = x;
                wave_sync();
                tft_system_qr(pts, N, w->nrm, jw->A);
                x = (lane < 27) ? w->t[lane] : 0.0;
            }
            it1 += 10000;
        } else {
            double r2, risk;
            if constexpr (G == 64) {                                         // rows p and 16 + p of G on every position p of a row of 16 lanes (row_eig.h)
                const int p = opaque_lane_int(lane & 15);                    // (the rows' index tables are rebuilt per triplet, not kept across the loop)
                const bool hi = p < 11;
                double ga[27], gb[27], g0[16], da, db;
                gram_row27(w->mom, p, ga, da);
                gram_row27(w->mom, hi ? 16 + p : 0, gb, db);
#pragma unroll
                for (int c = 0; c < 16; ++c) g0[c] = ga[c];
#pragma unroll
                for (int c = 0; c < 27; ++c) gb[c] = hi ? gb[c] : 0.0;
                phase_stamp(dbg, 3, wl);
                x = row_min_eigvec<27>(g0, gb, da, hi ? db : 0.0, w->Lp, EIG_MAXIT, &it1, &r2, false, 0.0, 0.0, &risk);
            } else {
                double g[27], diag;
                gram_row27(w->mom, (lane < 27) ? lane : 0, g, diag);
                phase_stamp(dbg, 3, wl);
                x = wave_min_eigvec_reg<27, G>(g, diag, w->Lp, EIG_MAXIT, &it1, &r2, false, 0.0, &risk);
            }
            ok = ok && eig_converged(r2) && risk == 0.0;
        }
        if (lane < 27) w->t[lane] = x;
        wave_sync();
    }
    if (dbg && lane < 27) dbg[lane] = w->t[lane];
    phase_stamp(dbg, 4, wl);
    ok = epipoles_from_tensor<G, JAC>(w->t, w->nullv, w->epi, false) && ok;  // :71-79
    phase_stamp(dbg, 5, wl);
    if (dbg && lane < 6) dbg[27 + lane] = w->epi[lane];
    if (lane == 0) frame_of(w->epi, w->Q);                                   // Q2 from e21
    if (lane == 1) frame_of(w->epi + 3, w->Q + 9);                           // Q3 from e31
    wave_sync();
    if (JAC) {                                                               // :84 from R: svd(A Up) == svd(R Up), A = Q R
        double* Bm = w->Lp;                                                  // R Up, 27 x 15: row r built by lane r ...
        if (lane < 27) {
            double g[27];                                                    // row `lane` of R (zeros below the diagonal)
#pragma unroll
            for (int c = 0; c < 27; ++c) g[c] = jw->A[lane * 27 + c];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const int jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc += g[j + 3 * k + 9 * i] * w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk];
                    Bm[lane * 15 + 5 * i + m] = acc;
                }
        }
        wave_sync();
        double a[27];                                                        // ... column c read back by lane c < 15
#pragma unroll
        for (int r = 0; r < 27; ++r) a[r] = (lane < 15) ? Bm[r * 15 + lane] : 0.0;
        wave_sync();
        double* R2 = jw->A;                                                  // 15 x 15; R itself is no longer needed
        for (int e = lane; e < 225; e += WAVE) R2[e] = 0.0;
        wave_sync();
        wave_qr_cols_append<15, 27>(a, R2);
        phase_stamp(dbg, 6, wl);
        double h[15];
        wave_qr_rows_from_lds<15>(R2, h);
        const double x = wave_qr_min_rsv<15>(h, jw->A, jw->A + 256, w->Lp, (hint & 2) ? 0 : EIG_MAXIT, &it2);
        it2 += 10000;
        if (lane < 15) w->tp[lane] = x;
        wave_sync();
    } else {
    // Gp = Up' G Up (15x15), lower triangle, packed into Lp; entry (a,b), a = 5 i + m
    double* Gp = w->Lp;
    for (int e = lane; e < 120; e += G) {
        int a = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (tri_index(a + 1, 0) <= e) ++a;
        while (tri_index(a, 0) > e) --a;
        const int b = e - tri_index(a, 0);
        double a2[3], a3[3], b2[3], b3[3], c2[4], c3[4];
        up_factors(w->Q, a % 5, a2, a3);
        up_factors(w->Q, b % 5, b2, b3);
        cvec(a2, b2, c2);
        cvec(a3, b3, c3);
        Gp[e] = bilinear44(w->mom + 16 * hht_index(a / 5, b / 5), c3, c2);
    }
    wave_sync();
    phase_stamp(dbg, 6, wl);
    {                                                                        // :84
        double g[15], diag = 0.0, x;
        const int rr = (G == 64) ? (lane & 15) : lane;                       // G == 64: every row of 16 lanes holds the matrix (row_eig.h)
        const bool have = rr < 15;
        const int r = have ? rr : 0;
#pragma unroll
        for (int c = 0; c < 15; ++c) { g[c] = (c <= r && have) ? Gp[tri_index(r, c)] : 0.0; diag = (c == r) ? g[c] : diag; }
        wave_sync();
        // start from the unconstrained solution projected onto range(E): tp0 = Up' t (9 non-zeros per basis vector).  The
        // constrained tensor differs from it at noise level, which saves one of the four inverse iterations.
        double tp0 = 0.0;
        if (have) {
            const int i = r / 5, m = r % 5, jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) tp0 += w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk] * w->t[j + 3 * k + 9 * i];
        }
        double r2, risk;
        if constexpr (G == 64) {
            double none[1] = {0.0};
            x = row_min_eigvec<15>(g, none, diag, 0.0, w->Lp, EIG_MAXIT, &it2, &r2, true, tp0, 0.0, &risk);
        } else {
            x = wave_min_eigvec_reg<15, G>(g, diag, w->Lp, EIG_MAXIT, &it2, &r2, true, tp0, &risk);
        }
        ok = ok && eig_converged(r2) && risk == 0.0;
        if (lane < 15) w->tp[lane] = x;
        wave_sync();
    }
    }
    phase_stamp(dbg, 7, wl);
    if (lane < 27) {                                                         // t = Up * tp   (:85)
        const int i = lane / 9, k = (lane % 9) / 3, j = lane % 3;
        double acc = 0.0;
        for (int m = 0; m < 5; ++m) {
            const int jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
            acc += w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk] * w->tp[5 * i + m];
        }
        w->t[lane] = acc;
    }
    wave_sync();
    if (dbg && lane < 27) dbg[33 + lane] = w->t[lane];
    if (dbg && lane == 0) { dbg[69] = (double)it1; dbg[70] = (double)it2; }
    if (want_P && lane < 3) {
        // a = pinv(E) t (:86): T_i = a_i e31' - e21 b_i'; particular solution with b_i.e31 = 0, then
        // the minimum-norm gauge  a_i += c e21, b_i += c e31,  c = -(a_i.e21 + b_i.e31)/(|e21|^2+|e31|^2)
        const int i = lane;
        const double* e21 = w->epi; const double* e31 = w->epi + 3;
        double ai[3], bi[3];
        for (int j = 0; j < 3; ++j) ai[j] = w->t[j + 9 * i] * e31[0] + w->t[j + 3 + 9 * i] * e31[1] + w->t[j + 6 + 9 * i] * e31[2];
        const double ae = ai[0] * e21[0] + ai[1] * e21[1] + ai[2] * e21[2];
        for (int k = 0; k < 3; ++k) {
            const double tte = w->t[3 * k + 9 * i] * e21[0] + w->t[1 + 3 * k + 9 * i] * e21[1] + w->t[2 + 3 * k + 9 * i] * e21[2];
            bi[k] = e31[k] * ae - tte;
        }
        const double be = bi[0] * e31[0] + bi[1] * e31[1] + bi[2] * e31[2];
        const double n21 = e21[0] * e21[0] + e21[1] * e21[1] + e21[2] * e21[2];
        const double n31 = e31[0] * e31[0] + e31[1] * e31[1] + e31[2] * e31[2];
        const double c = -(ae + be) / (n21 + n31);
        for (int j = 0; j < 3; ++j) { w->pa[3 * i + j] = ai[j] + c * e21[j]; w->pa[9 + 3 * i + j] = bi[j] + c * e31[j]; }
    }
    wave_sync();
    return !wave_any(!ok);                                                   // wave-uniform
}

// linearTFT.m:33-91 on the normalised correspondences (data pass + the rest), whole wavefront.
template <bool JAC>
__device__ __forceinline__ bool linear_tft_wave(PoseLds* w, JacobiLds* jw, const double* pts, int N, bool want_P, double* dbg, const int hint = 0) {
    if (!JAC) accumulate_moments(w, pts, N);
    phase_stamp(dbg, 2);
    return linear_tft_middle<JAC, 64>(w, jw, pts, N, want_P, dbg, hint);
}

// R_t_from_TFT.m:44-58 and the decomposition of E21, E31 (:85-88): the lane-sparse part of
// R_t_from_TFT on the de-normalised tensor w->T1, up to the candidate cameras.
template <int G, bool EXACT = true>
__device__ inline bool rt_prepare(PoseLds* w, double* dbg) {
    const int lane = Group<G>::lane();
    transform_tft_inverse<G>(w->T1, w->T2, w->Lp, [w](int v) { return load_K(w->calm, v); });   // :44
    const bool ok = epipoles_from_tensor<G, EXACT>(w->T2, w->nullv, w->epi, true);   // :47-55
    double* Ein = w->Minv;                                                   // 18 doubles of scratch
    if (lane < 2) {
        const double* e21 = w->epi; const double* e31 = w->epi + 3;
        Mat3 M;                                                              // [T1*e T2*e T3*e]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    // lane 0: (T_i e31)(r) = sum_k T(r,k,i) e31(k); lane 1: (T_i' e21)(r) = sum_j T(j,r,i) e21(j)
                    acc += (lane == 0) ? w->T2[r + 3 * q + 9 * i] * e31[q] : w->T2[q + 3 * r + 9 * i] * e21[q];
                }
                M.m[r][i] = acc;
            }
        const double* e = (lane == 0) ? e21 : e31;
        const double sg = (lane == 0) ? 1.0 : -1.0;                          // E31 = -crossM(epi31)*[...]  (:58)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Ein[9 * lane + 0 + c] = sg * (-e[2] * M.m[1][c] + e[1] * M.m[2][c]);
            Ein[9 * lane + 3 + c] = sg * (e[2] * M.m[0][c] - e[0] * M.m[2][c]);
            Ein[9 * lane + 6 + c] = sg * (-e[1] * M.m[0][c] + e[0] * M.m[1][c]);
        }
    }
    wave_sync();
    phase_stamp(dbg, 9, Group<G>::index() * G);
    recover_prepare<G>(w, Ein);                                              // svd(E), candidate poses and cameras
    return ok;
}

// R_t_from_TFT.m:40-76 on the de-normalised tensor w->T1 (whole wavefront).
// *ok (EXACT = false): cleared when a fast tier could not finish or certify its part.
template <bool EXACT = true>
__device__ inline int rt_from_tft_wave(PoseLds* w, const double* pts, int N, double* dbg, bool* ok = nullptr) {
    bool fine = wave_any(!rt_prepare<64, EXACT>(w, dbg)) == false;
    const int st = recover_vote<EXACT>(w, pts, N, dbg, &fine);               // :61,:64
    phase_stamp(dbg, 11);
    fine = scale_t3<EXACT>(w, pts, N, dbg) && fine;                          // :68-74
    phase_stamp(dbg, 12);
    if (ok && !fine) *ok = false;
    return st;
}

template <bool JAC>
__global__ void __launch_bounds__(64, 2) k_linear_tft_pose(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLds* jw = JAC ? reinterpret_cast<JacobiLds*>(smem + base) : nullptr;
    double* lds_pts = smem + base + (JAC ? ((JACOBI_LDS_DOUBLES + 1) & ~1) : 0);
    const int lane = lane_id();
    const long nwork = (a.retry_list && (a.flags & FLAG_ONLY_RETRY)) ? (long)*a.retry_count : a.B;
    for (long wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        // a list entry: triplet index in bits 0 .. 27, what the row kernel knows about its failure in bits 28 .. 29 (RETRY_HINT_*)
        const int entry = (a.retry_list && (a.flags & FLAG_ONLY_RETRY)) ? a.retry_list[wi] : 0;
        const long b = (a.retry_list && (a.flags & FLAG_ONLY_RETRY)) ? (long)(entry & RETRY_INDEX_MASK) : wi;
        const int hint = JAC ? (int)((unsigned)entry >> RETRY_HINT_SHIFT) : 0;
        // (opaque: the loop makes one trip per workgroup; what the optimiser derives from N and the flags ahead of it -- N * 6, N < 7,
        // flag tests as scalar masks, ... -- would be computed in the pre-header and spilled across the whole body, see wave.h::lane_id)
        const int N = opaque_int(a.N), flags = opaque_int(a.flags);
        if ((flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;        // wave-uniform
        double* dbg = a.dbg ? a.dbg + b * DBG_STRIDE : nullptr;
        const double* src = a.corresp + b * 6 * (long)N;
        const double* pts = src;
        wave_sync();
        bool bad_index = false;
        if (a.sample_idx) {
            bad_index = gather_points(a.corresp, a.sample_idx + b * (long)N, lds_pts, N, a.sample_ns);
            pts = lds_pts;
        } else if (flags & FLAG_STAGE_LDS) {
            stage_points(src, lds_pts, N);
            pts = lds_pts;
        }
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        phase_stamp(dbg, 0);
        int status = ST_OK;
        if (N < 7 || bad_index) {                                            // experiments.m:99 (or a sample index outside the scene)
            status = ST_TOO_FEW;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        } else {
            normalise3(pts, N, w->nrm);                                      // LinearTFTPoseEstimation.m:45-47
            if (dbg && lane < 9) dbg[71 + lane] = w->nrm[lane];
            phase_stamp(dbg, 1);
            bool ok = linear_tft_wave<JAC>(w, jw, pts, N, false, dbg, hint);  // :50
            phase_stamp(dbg, 8);
            if (ok) {
                transform_tft_inverse(w->t, w->T1, w->Lp, [w](int v) { return normal_matrix(w->nrm, v); });   // :53
                status = rt_from_tft_wave<JAC>(w, pts, N, dbg, &ok);         // :56
            }
            if (ok && a.reconst) ok = final_reconst<JAC>(w, pts, N, a.reconst + b * 3 * (long)N);   // :59-60
            if (!ok) {
                status = ST_RETRY;                                           // redone by k_linear_tft_pose<true>
            } else {
                write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
                if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
                phase_stamp(dbg, 13);
                // non-finite outputs -> status 2
                double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
                const bool bad = !(fabs(chk) <= 1.79e308);
                if (wave_any(bad)) { if (status == ST_OK) status = ST_NONFINITE; wave_nan_outputs(a.Rt2, a.Rt3, a.T, a.reconst, b, N); }
            }
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = 0;                                       // :62
            a.status[b] = status;
        }
    }
}

}  // namespace tff
