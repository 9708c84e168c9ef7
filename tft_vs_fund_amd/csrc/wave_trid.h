// x = pinv(S) b for a symmetric, possibly indefinite and rank-deficient n x n matrix (n <= 32) under MATLAB's truncation
//     x = sum_{|lambda_k| > tol} y_k (y_k' b) / lambda_k                         (Gauss_Helmert.m:67, `pinv(M + 1e-12 I) * b`)
// by ONE wavefront and WITHOUT forming the eigenvector matrix -- the solver of the Gauss-Helmert models whose constraints are
// redundant (FaugPapaTFTPoseEstimation.m:114-150: twelve constraints on a variety of codimension nine, so the KKT matrix is
// singular by construction and `pinv` really truncates).
//
// wave_eigh_ql (wave_eig.h) spends ~60 k instructions of one wavefront on the 39 x 39 matrix, two thirds of them accumulating
// reflections and ~600 QL rotations into the eigenvectors.  Here:
//   1. Householder reduction to tridiagonal form, in place in LDS; the right-hand side rides along as column n of the augmented
//      array (its lane sees v = 0, so the rank-two update degenerates to b -= beta v v'b); the reflectors stay in the annihilated
//      rows for the back-transformation.  Nothing is accumulated.
//   2. Eigenvalues by counting: a Sturm count of T - sigma I costs one n-step recurrence PER LANE, so 64 shifts spaced geometrically
//      over (tol, |T|] (and over the negative side when eigenvalues lie under -tol) bracket every kept eigenvalue in one pass; a few
//      bisection rounds (one eigenvalue per lane, all lanes in lockstep) isolate them.
//   3. One kept eigenpair per lane by Rayleigh-quotient iteration with the twisted factorisation N_r D N_r' = T - sigma I
//      (Parlett / Dhillon): forward pivots Dp -- their signs are the inertia, a free Sturm count that keeps shrinking the bracket --
//      backward pivots Dm, gamma_k = Dp_k + Dm_k - (d_k - sigma), twist index r = argmin |gamma|, z_r = 1,
//      z_i = -(e_i / Dp_i) z_(i+1) below r, z_(i+1) = -(e_i / Dm_(i+1)) z_i above: (T - sigma) z = gamma_r e_r, so the Rayleigh
//      correction is gamma_r / |z|^2 and the residual |gamma_r| / |z|.  A correction that leaves the bracket becomes a bisection
//      step, so a lane can only converge to ITS eigenvalue.  Cubic convergence: 4 .. 6 iterations from a 1 % bracket.
//   4. x^ = sum_k coef_k z_k through a transposing pass over LDS, x = H_0 ... H_(n-3) x^.
// ~12 k instructions for n = 31.  The eigenvectors of a pair of kept eigenvalues closer than ~1e-4 |T| lose orthogonality in
// proportion (error ~1e-16 |T| / gap, as in any inverse-iteration scheme without re-orthogonalisation); the Gauss-Helmert systems
// this serves have their kept eigenvalues spread over three decades (tools/proto_trid_pinv.py is the numpy twin of this file, step
// for step, and the tests run both).
#pragma once
#include "wave.h"

namespace tff {

constexpr int TRID_MAX = 32;                                   // largest matrix: one kept eigenvalue per lane of a half-wavefront
constexpr int TRID_LD = 33;                                    // odd leading dimension of the per-lane arrays: conflict-free by lane AND by row
constexpr int TRID_WORK_DOUBLES = 2 * TRID_MAX * TRID_LD;      // z | Dm  (the shift / count tables of step 2 overlay Dm)
constexpr int TRID_SMALL_DOUBLES = 3 * TRID_MAX;               // d | e | e^2

// reciprocal to ~1e-16: v_rcp_f64 (~1e-7) + two Newton steps (five instructions against ~12 for an IEEE division)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = fast_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double trid_guard(double q) { return (fabs(q) < 1e-300) ? -1e-300 : q; }

// number of eigenvalues of the tridiagonal matrix (d, e^2 in LDS) below sig: negative pivots of the LDL' factorisation of T - sig I
__device__ __forceinline__ int trid_count(const lds_ptr dS, const lds_ptr e2S, const int n, const double sig) {
    double q = dS[0] - sig;
    int cnt = (q < 0.0) ? 1 : 0;
#pragma unroll 4
    for (int j = 1; j < n; ++j) {
        q = trid_guard(q);
        q = (dS[j] - sig) - e2S[j - 1] * rcp_nr(q);
        cnt += (q < 0.0) ? 1 : 0;
    }
    return cnt;
}

// Step 1.  A: n x n symmetric (full storage, leading dimension lda >= n + 1) with the right-hand side in column n; on return the
// tridiagonal entries are in dS / eS / e2S, the transformed right-hand side in column n, reflector k in A[k][k+1 .. n) with
// beta_k = 2 / v'v in A[k][k] (0: no reflection).  scr: 2 n doubles.
__device__ __forceinline__ void wave_tridiag_inplace(lds_ptr A, const int lda, const int n, lds_ptr dS, lds_ptr eS, lds_ptr e2S, lds_ptr scr) {
    const int lane = lane_id();
    const int rl = (lane <= n) ? lane : 0;
    double dreg = 0.0, ereg = 0.0;                                           // lane k: T[k][k], T[k][k+1]
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        const bool act = lane > k && lane < n;
        const double x = act ? A[k * lda + lane] : 0.0;                      // row k right of the diagonal (= column k below it)
        if (lane == k) dreg = A[k * lda + k];
        const double x1 = wave_bcast(x, k + 1);
        const double tail = wave_sum((lane > k + 1) ? x * x : 0.0);
        if (wave_uniform_i(tail == 0.0)) {                                   // already tridiagonal in this column
            if (lane == k) { ereg = x1; A[k * lda + k] = 0.0; }
            wave_sync();
            continue;
        }
        const double sigma = tail + x1 * x1;
        const double nrm = sqrt(sigma);
        const double alpha = (x1 > 0.0) ? -nrm : nrm;
        const double v = (lane == k + 1) ? x - alpha : x;                    // Householder vector (0 on lanes <= k and >= n)
        const double beta = 1.0 / (sigma + fabs(x1) * nrm);                  // 2 / v'v
        if (lane == k) { ereg = alpha; A[k * lda + k] = beta; }
        if (act) A[k * lda + lane] = v;                                      // kept for the back-transformation
        if (lane < n) scr[lane] = v;
        wave_sync();
        double p = 0.0;                                                      // lane n: v'b (the right-hand side column)
#pragma unroll 4
        for (int c = k + 1; c < n; ++c) p += A[c * lda + rl] * scr[c];
        p = (lane > k && lane <= n) ? p * beta : 0.0;
        const double K = 0.5 * beta * wave_sum(p * v);
        const double q = p - K * v;
        if (lane < n) scr[n + lane] = q;
        wave_sync();
        if (lane > k && lane <= n) {
#pragma unroll 4
            for (int c = k + 1; c < n; ++c) A[c * lda + lane] -= scr[c] * q + scr[n + c] * v;
        }
        wave_sync();
    }
    if (n >= 2 && lane == n - 2) { dreg = A[(n - 2) * lda + n - 2]; ereg = A[(n - 2) * lda + n - 1]; }
    if (lane == n - 1) { dreg = A[(n - 1) * lda + n - 1]; ereg = 0.0; }
    if (lane < n) { dS[lane] = dreg; eS[lane] = ereg; e2S[lane] = ereg * ereg; }
    wave_sync();
}

// Steps 2 - 4 for the tridiagonal matrix (dS, eS, e2S; n <= TRID_MAX) and the right-hand side bh (stride ldb): x^ -> lane j (< n).
// work: TRID_WORK_DOUBLES.  *kept_out: eigenvalues kept; *fail: 1 when more than TRID_MAX eigenvalues are kept (cannot happen for
// n <= TRID_MAX) or an eigenpair did not converge.
__device__ __forceinline__ double wave_trid_pinv(const lds_ptr dS, const lds_ptr eS, const lds_ptr e2S, const int n, const lds_ptr bh, const int ldb,
                                                 const double tol, lds_ptr work, int* kept_out, int* fail, double* dbg = nullptr) {
    const int lane = lane_id();
    lds_ptr Z = work;                                                        // Z[j * TRID_LD + lane]: pivots Dp, then the eigenvector
    lds_ptr Dm = work + TRID_MAX * TRID_LD;
    lds_ptr shS = Dm;                                                        // step 2 only: 64 shifts | 64 counts
    lds_ptr cntS = Dm + 64;
    // Gershgorin bounds
    double gl, gu, nrmT;
    {
        const int j = (lane < n) ? lane : 0;
        const double rad = fabs(eS[j]) + ((j > 0) ? fabs(eS[j - 1]) : 0.0);
        const double lo_j = (lane < n) ? dS[j] - rad : 1e300, hi_j = (lane < n) ? dS[j] + rad : -1e300;
        gl = -wave_max(-lo_j);
        gu = wave_max(hi_j);
        nrmT = (fabs(gl) > fabs(gu)) ? fabs(gl) : fabs(gu);
    }
    // kept eigenvalues: lambda < -tol (c_neg of them) and lambda > tol (n - c_pos)
    const double tol_up = __longlong_as_double(__double_as_longlong(tol) + 1);   // nextafter(tol, +inf), tol > 0
    const int cq = trid_count(dS, e2S, n, (lane & 1) ? tol_up : -tol);
    const int c_neg = wave_bcast_i(cq, 0), c_pos = wave_bcast_i(cq, 1);
    const int m_neg = c_neg, m_pos = n - c_pos, kept = m_neg + m_pos;
    *kept_out = kept;
    *fail = 0;
    if (kept == 0 || !(nrmT > tol)) { *kept_out = 0; return 0.0; }           // wave-uniform
    // ---- step 2: one Sturm count per lane on geometrically spaced shifts ----
    int Lp = (m_pos == 0) ? 0 : ((m_neg == 0) ? 64 : (64 * m_pos + kept / 2) / kept);
    if (m_pos > 0 && Lp < 2) Lp = 2;
    if (m_neg > 0 && Lp > 62) Lp = 62;
    const int Ln = 64 - Lp;
    {
        const bool pos = lane < Lp;
        const int t = pos ? lane : lane - Lp, L = pos ? Lp : Ln;
        const double top = (pos ? gu : -gl) * (1.0 + 1e-12) + 1e-300;
        const double lg = log2(((top > tol) ? top : tol * 2.0) / tol) / (double)((L > 1) ? L - 1 : 1);
        const double mag = (t == L - 1) ? ((top > tol) ? top : tol * 2.0) : tol * exp2(lg * (double)t);
        const double sh = pos ? mag : -mag;
        shS[lane] = sh;
        cntS[lane] = (double)trid_count(dS, e2S, n, sh);
    }
    wave_sync();
    const bool own = lane < kept;                                            // lanes 0 .. m_neg-1: negative side (ascending), then the positive side
    const bool neg_side = lane < m_neg;
    const int idx = neg_side ? lane : c_pos + (lane - m_neg);                // number of the lane's eigenvalue (ascending, 0-based)
    double lo = 0.0, hi = 0.0;
    int clo = 0, chi = 0;
    {
        int j = -1;
        if (neg_side) {                                                      // first shift -mag_j (descending) with count <= idx
#pragma unroll 1
            for (int t = 0; t < Ln; ++t) { const int c = (int)cntS[Lp + t]; if (j < 0 && c <= idx) j = t; }
            j = (j < 0) ? Ln - 1 : j;
            lo = shS[Lp + j]; clo = (int)cntS[Lp + j];
            hi = (j > 0) ? shS[Lp + j - 1] : -tol; chi = (j > 0) ? (int)cntS[Lp + j - 1] : c_neg;
        } else {                                                             // first shift +mag_j (ascending) with count > idx
#pragma unroll 1
            for (int t = 0; t < Lp; ++t) { const int c = (int)cntS[t]; if (j < 0 && c > idx) j = t; }
            j = (j < 0) ? ((Lp > 0) ? Lp - 1 : 0) : j;
            hi = shS[j]; chi = (int)cntS[j];
            lo = (j > 0) ? shS[j - 1] : tol; clo = (j > 0) ? (int)cntS[j - 1] : c_pos;
        }
    }
    wave_sync();                                                             // the tables (in Dm) are dead from here on
#pragma unroll 1
    for (int round = 0; round < 60; ++round) {                               // bisection until every bracket holds exactly one eigenvalue
        const bool iso = !own || (chi - clo) == 1;
        if (round >= 3 && !wave_any(!iso)) break;
        const double mid = 0.5 * (lo + hi);
        const int c = trid_count(dS, e2S, n, mid);
        const bool right = c <= idx;                                         // eigenvalue idx is >= mid
        lo = right ? mid : lo; clo = right ? c : clo;
        hi = right ? hi : mid; chi = right ? chi : c;
    }
    phase_stamp(dbg, 29);
    // ---- step 3: Rayleigh-quotient iteration with the twisted factorisation, one eigenpair per lane ----
    double sig = 0.5 * (lo + hi), lam = sig, nz2 = 1.0;
    bool done = !own;
    const int zl = (lane < TRID_MAX) ? lane : 0;
    int it = 0;
#pragma unroll 1
    for (it = 0; it < 48; ++it) {
        if (!wave_any(!done)) break;
        if (!done) {
            // forward pivots and the inertia
            double dp = dS[0] - sig;
            int negc = (dp < 0.0) ? 1 : 0;
#pragma unroll 2
            for (int i = 0; i + 1 < n; ++i) {
                dp = trid_guard(dp);
                Z[i * TRID_LD + zl] = dp;
                dp = (dS[i + 1] - sig) - e2S[i] * rcp_nr(dp);
                negc += (dp < 0.0) ? 1 : 0;
            }
            Z[(n - 1) * TRID_LD + zl] = dp;
            // backward pivots, gamma, twist index
            double dm = dS[n - 1] - sig;
            double gbest = dp;                                               // gamma_(n-1) = Dp_(n-1) + Dm_(n-1) - (d - sig) = Dp_(n-1)
            int r = n - 1;
#pragma unroll 2
            for (int i = n - 2; i >= 0; --i) {
                dm = trid_guard(dm);
                Dm[(i + 1) * TRID_LD + zl] = dm;
                const double di = dS[i] - sig;
                dm = di - e2S[i] * rcp_nr(dm);
                const double g = Z[i * TRID_LD + zl] + dm - di;
                if (fabs(g) < fabs(gbest)) { gbest = g; r = i; }
            }
            // eigenvector: z_r = 1, downwards with the forward pivots, upwards with the backward ones
            double zi = 1.0, s2 = 1.0;
#pragma unroll 2
            for (int i = n - 2; i >= 0; --i) {
                if (i < r) {
                    zi = -(eS[i] * rcp_nr(Z[i * TRID_LD + zl])) * zi;
                    Z[i * TRID_LD + zl] = zi;
                    s2 += zi * zi;
                }
            }
            Z[r * TRID_LD + zl] = 1.0;
            zi = 1.0;
#pragma unroll 2
            for (int i = 0; i + 1 < n; ++i) {
                if (i >= r) {
                    zi = -(eS[i] * rcp_nr(Dm[(i + 1) * TRID_LD + zl])) * zi;
                    Z[(i + 1) * TRID_LD + zl] = zi;
                    s2 += zi * zi;
                }
            }
            const double corr = gbest / s2;
            const double resid = fabs(gbest) * rsqrt(s2);
            const bool right = negc <= idx;                                  // eigenvalue idx is >= sig
            lo = right ? sig : lo;
            hi = right ? hi : sig;
            const bool conv = (resid <= 8e-16 * nrmT) || (fabs(corr) <= 4e-16 * fabs(sig)) || !(s2 == s2);
            if (conv) {
                lam = sig + corr; nz2 = s2; done = true;
            } else {
                const double cand = sig + corr;
                sig = (cand >= lo && cand <= hi) ? cand : 0.5 * (lo + hi);
            }
        }
    }
    if (wave_any(!done)) *fail = 1;
    wave_sync();
    phase_stamp(dbg, 30);
    if (dbg && lane_id() == 0) dbg[79] = (double)it;
    // ---- step 4: coefficients, x^ ----
    if (own) {
        double dot = 0.0;
#pragma unroll 4
        for (int j = 0; j < n; ++j) dot += Z[j * TRID_LD + zl] * bh[j * ldb];
        const double coef = dot / (lam * nz2);
#pragma unroll 4
        for (int j = 0; j < n; ++j) Z[j * TRID_LD + zl] *= coef;
    }
    wave_sync();
    double x = 0.0;
    if (lane < n) {
#pragma unroll 4
        for (int l = 0; l < kept; ++l) x += Z[lane * TRID_LD + l];
    }
    wave_sync();
    return x;
}

// Steps 1 - 4: x = pinv(S) b under the tolerance tol, S and b as for wave_tridiag_inplace (DESTROYED); sol[0 .. n) <- x.
// small: TRID_SMALL_DOUBLES, work: TRID_WORK_DOUBLES (its first 2 n doubles double as the reduction's scratch).
__device__ __forceinline__ void wave_pinv_solve_trid(double* A_, const int lda, const int n, const double tol, double* sol, double* small_, double* work_,
                                                     int* kept_out, int* fail, double* dbg = nullptr) {
    const lds_ptr A = to_lds(A_), small = to_lds(small_), work = to_lds(work_);
    const int lane = lane_id();
    lds_ptr dS = small, eS = small + TRID_MAX, e2S = small + 2 * TRID_MAX;
    wave_tridiag_inplace(A, lda, n, dS, eS, e2S, work);
    phase_stamp(dbg, 28);
    double x = wave_trid_pinv(dS, eS, e2S, n, A + n, lda, tol, work, kept_out, fail, dbg);
    phase_stamp(dbg, 31);
    // x = H_0 ... H_(n-3) x^
#pragma unroll 1
    for (int k = n - 3; k >= 0; --k) {
        const double beta = A[k * lda + k];
        const double v = (lane > k && lane < n) ? A[k * lda + lane] : 0.0;
        const double s = wave_sum(v * x);
        x -= beta * s * v;
    }
    if (lane < n) sol[lane] = x;
    wave_sync();
}

}  // namespace tff
