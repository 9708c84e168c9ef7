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
constexpr int TRID_WORK_DOUBLES = (2 * TRID_MAX + 1) * TRID_LD + 1;   // z | Dm or the trailing minors q_1 .. q_32  (the shift / count tables of step 2 overlay them)
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


// Step 1 again, the version in use.  The loops above wait out one LDS round trip per element: every read-modify-write of A is followed
// by loads of scr[] that the compiler cannot move ahead of the store (216 k cycles for n = 31, a quarter of a Gauss-Helmert iteration
// of the FaugPapa kernel).  Here (i) the Householder vector v and q = beta (A v - K v) reach the other lanes through v_readlane
// (wave-uniform source lane in a scalar register), not through LDS; (ii) the lane's column entries are moved in bursts of 16: one
// burst of loads, arithmetic in registers, one burst of stores -- one exposed latency per 16 elements, 32 transient registers.
// A register-resident matrix (64 registers for the whole reduction) was faster still in isolation but made the kernels that inline
// this spill around every reduction of theirs.
// On return as wave_tridiag_inplace: reflector k in A[k][k+1 .. n), beta_k in A[k][k], transformed right-hand side in column n;
// lane k holds T[k][k] / T[k][k+1] in dreg / ereg.
__device__ __forceinline__ void wave_tridiag_burst(lds_ptr A, const int lda, const int n, double& dreg, double& ereg) {
    const int lane = lane_id();
    const int rl = (lane <= n) ? lane : 0;
    dreg = 0.0; ereg = 0.0;
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        const bool act = lane > k && lane < n;
        const double x = act ? A[k * lda + lane] : 0.0;                      // row k right of the diagonal (= column k below it)
        if (lane == k) dreg = A[k * lda + k];
        const double x1 = wave_bcast(x, k + 1);
        const double tail = wave_sum((lane > k + 1) ? x * x : 0.0);
        if (wave_uniform_i(tail == 0.0)) {                                   // already tridiagonal in this column
            if (lane == k) { ereg = x1; A[k * lda + k] = 0.0; }
            continue;
        }
        const double sigma = tail + x1 * x1;
        const double nrm = sqrt(sigma);
        const double alpha = (x1 > 0.0) ? -nrm : nrm;
        const double v = (lane == k + 1) ? x - alpha : x;                    // Householder vector (0 on lanes <= k and >= n)
        const double beta = 1.0 / (sigma + fabs(x1) * nrm);                  // 2 / v'v
        if (lane == k) { ereg = alpha; A[k * lda + k] = beta; }
        if (act) A[k * lda + lane] = v;                                      // kept for the back-transformation
        double p0 = 0.0, p1 = 0.0;                                           // p = A v (lane n: v'b), two chains
#pragma unroll 1
        for (int base = k + 1; base < n; base += 16) {
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int c = (base + j < n) ? base + j : n - 1; a[j] = A[c * lda + rl]; }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double vc = wave_bcast(v, base + j);                   // 0 beyond n (v vanishes on lanes >= n; base + j <= 62)
                if (j & 1) p1 += a[j] * vc; else p0 += a[j] * vc;
            }
        }
        const double p = (lane > k && lane <= n) ? (p0 + p1) * beta : 0.0;
        const double K = 0.5 * beta * wave_sum(p * v);
        const double q = p - K * v;
        const bool upd = lane > k && lane <= n;                              // (the broadcasts stay outside the divergent part: every lane takes part)
#pragma unroll 1
        for (int base = k + 1; base < n; base += 16) {
            double a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int c = (base + j < n) ? base + j : n - 1; a[j] = A[c * lda + rl]; }
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] -= wave_bcast(v, base + j) * q + wave_bcast(q, base + j) * v;
#pragma unroll
            for (int j = 0; j < 16; ++j) if (upd && base + j < n) A[(base + j) * lda + lane] = a[j];
        }
        wave_sync();
    }
    if (n >= 2 && lane == n - 2) { dreg = A[(n - 2) * lda + n - 2]; ereg = A[(n - 2) * lda + n - 1]; }
    if (lane == n - 1) { dreg = A[(n - 1) * lda + n - 1]; ereg = 0.0; }
    wave_sync();
}

// Step 1 in REGISTERS (the version in use where the caller has the registers: 64 for the matrix).  Lane r <= n owns column r of the
// augmented array, a[c] = A[c][r] (lane n: the right-hand side), zero beyond n.  The Householder vector reaches the other lanes through
// v_readlane, the inner loops are pure VALU (fully unrolled, compile-time register indices: ~6 k instructions of straight-line code).
// On return lane k holds T[k][k] / T[k][k+1] in dreg / ereg, lane n the transformed right-hand side in a[0 .. n), and reflector k stays
// in a[k] of the lanes > k with beta_k in lane k's a[k] (wave_tridiag_back_reg).
template <int NMAX>
__device__ __forceinline__ void wave_tridiag_reg(double (&a)[NMAX], const int n, double& dreg, double& ereg) {
    const int lane = lane_id();
    dreg = 0.0; ereg = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < NMAX; ++k) {
        const bool act = lane > k && lane < n;
        const double x = act ? a[k] : 0.0;                                   // row k right of the diagonal
        if (lane == k) dreg = a[k];
        const double x1 = wave_bcast(x, k + 1);
        const double tail = wave_sum((lane > k + 1) ? x * x : 0.0);
        if (wave_uniform_i(tail == 0.0)) {                                   // nothing to annihilate (also every step k >= n - 2)
            if (lane == k) { ereg = x1; a[k] = 0.0; }
        } else {
            const double sigma = tail + x1 * x1;
            const double nrm = sqrt(sigma);
            const double alpha = (x1 > 0.0) ? -nrm : nrm;
            const double v = (lane == k + 1) ? x - alpha : x;                // Householder vector (0 on lanes <= k and >= n)
            const double beta = 1.0 / (sigma + fabs(x1) * nrm);              // 2 / v'v
            if (lane == k) { ereg = alpha; a[k] = beta; }
            double p0 = 0.0, p1 = 0.0;                                       // p = A v (lane n: v'b), two chains
#pragma unroll
            for (int c = k + 1; c < NMAX; ++c) {
                const double vc = wave_bcast(v, c);
                if ((c - k) & 1) p0 += a[c] * vc; else p1 += a[c] * vc;
            }
            const double p = (lane > k && lane <= n) ? (p0 + p1) * beta : 0.0;
            const double K = 0.5 * beta * wave_sum(p * v);
            const double q = p - K * v;
#pragma unroll
            for (int c = k + 1; c < NMAX; ++c) a[c] -= wave_bcast(v, c) * q + wave_bcast(q, c) * v;
            if (act) a[k] = v;                                               // kept for the back-transformation
        }
    }
    if (lane == NMAX - 1) dreg = a[NMAX - 1];
}
// x (component `lane`) <- H_0 H_1 ... x with the reflectors wave_tridiag_reg left in a[]
template <int NMAX>
__device__ __forceinline__ double wave_tridiag_back_reg(const double (&a)[NMAX], const int n, double x) {
    const int lane = lane_id();
#pragma unroll
    for (int k = NMAX - 3; k >= 0; --k) {
        const double beta = wave_bcast(a[k], k);
        const double v = (lane > k && lane < n) ? a[k] : 0.0;
        const double s = wave_sum(v * x);
        x -= beta * s * v;
    }
    return x;
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
    // Guard for clustered kept eigenvalues: the eigenvectors come from INDEPENDENT iterations (no re-orthogonalisation), so two of them that
    // belong to nearly coincident eigenvalues may both be good Ritz vectors of the cluster and yet not orthogonal -- the sum over the cluster
    // would then be wrong by their overlap (eps |T| / gap).  Neighbouring eigenvalues are the closest ones: the cosine between the vectors of
    // lanes l and l + 1 is measured, and anything above 1e-9 (the parity gate of the callers) reports failure -- the caller redoes the system
    // with an eigen-decomposition that orthogonalises (wave_eigh_ql) or hands the triplet on.
    {
        const int lane_ = lane_id();
        const bool pair = own && lane_ + 1 < kept;
        double dotn = 0.0;
        const int zn = pair ? zl + 1 : zl;
#pragma unroll 4
        for (int j = 0; j < n; ++j) dotn += Z[j * TRID_LD + zl] * Z[j * TRID_LD + zn];
        const double nz2n = wave_shfl(nz2, (lane_ + 1) & 63);
        if (wave_any(pair && !(dotn * dotn <= 1e-18 * nz2 * nz2n))) *fail = 1;
    }
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
// ---- steps 2 - 4 again, division-free and in registers (the version in use; wave_trid_pinv above is its fall-back) -------------------
// The pivots of the LDL' / UDU' factorisations are ratios of consecutive leading / trailing principal minors of T - sigma I,
//     p_(k+1) = (d_k - sigma) p_k - e_(k-1)^2 p_(k-1)  (p_0 = 1),     q_k = (d_k - sigma) q_(k+1) - e_k^2 q_(k+2)  (q_n = 1),
// and everything the iteration needs can be read off the minors themselves: the number of eigenvalues below sigma is the number of sign
// changes of p_0 .. p_n; det = gamma_k p_k q_(k+1) for EVERY k, so the twist index argmin |gamma_k| is argmax |p_k q_(k+1)|;
// z_i = (-1)^(r-i) e_i ... e_(r-1) p_i / p_r below the twist, (-1)^(i-r) e_r ... e_(i-1) q_(i+1) / q_(r+1) above.  One division per lane
// and round instead of ~4 n reciprocals with their Newton steps, one fma on the dependent chain of each recurrence step instead of
// ~five instructions, and with T scaled to |T| <= 1 (by a power of two) the minors of a matrix of order <= 32 stay within 3^32 -- their
// magnitudes are watched all the same, and a lane whose minors leave [1e-250, 1e250] sends the whole solve to the fall-back.  d and e
// live one per lane and travel through v_readlane; the minors of a lane's own eigenvalue stay in its registers: no LDS in the loops.
// dsr / esr: lane j holds T[j][j] / T[j][j+1] (esr = 0 on lanes >= n - 1).  Other arguments and results as wave_trid_pinv.
__device__ __forceinline__ double trid_pow2_floor(double x) {              // 2^floor(log2 x), x > 0 normal
    return __longlong_as_double(__double_as_longlong(x) & 0x7ff0000000000000LL);
}
__device__ __forceinline__ int trid_count_fast(const double dsr, const double e2r, const int n, const double sig) {
    double po = 1.0, pc = wave_bcast(dsr, 0) - sig;
    bool sc_ = (pc < 0.0) || (pc == 0.0);                                    // a zero minor takes the sign opposite to its predecessor (p_0 = 1 > 0)
    int cnt = sc_ ? 1 : 0;
#pragma unroll
    for (int j = 1; j < TRID_MAX; ++j) {                                     // (no guard on n: the padding rows d = 4, e = 0 add eigenvalues at 4, above every shift)
        const double pn = (wave_bcast(dsr, j) - sig) * pc - wave_bcast(e2r, j - 1) * po;
        const bool sn = (pn < 0.0) || (pn == 0.0 && !sc_);
        cnt += (sn != sc_) ? 1 : 0;
        po = pc; pc = pn; sc_ = sn;
        const bool tiny = fabs(pc) < 1e-150;                                 // (counts only: a common rescaling changes no sign)
        pc = tiny ? pc * 0x1p+600 : pc;
        po = tiny ? po * 0x1p+600 : po;
    }
    (void)n;
    return cnt;
}

__device__ __forceinline__ double wave_trid_pinv_fast(const double dreg, const double ereg, const int n, const lds_ptr bh, const int ldb,
                                                      const double tol, lds_ptr work, int* kept_out, int* fail, double* dbg = nullptr) {
    const int lane = lane_id();
    lds_ptr shS = work;                                                      // step 2: 64 shifts | 64 counts; step 4: the transposing pass
    lds_ptr cntS = work + 64;
    lds_ptr Qs = work + TRID_MAX * TRID_LD;                                  // step 3: trailing minors q_k of the lane's shift, Qs[k * TRID_LD + lane]
    *fail = 0;
    // Gershgorin bounds, scaling by a power of two
    double gl, gu, nrmT;
    {
        const double e15 = wave_bcast(ereg, 15);
        const double eshr = dpp_mov<0x111>(ereg);                            // row_shr:1 = lane j - 1 (0 at the start of a row of 16; n <= 32)
        const double eprev = (lane == 16) ? e15 : eshr;
        const double rad = fabs(ereg) + fabs(eprev);
        const double lo_j = (lane < n) ? dreg - rad : 1e300, hi_j = (lane < n) ? dreg + rad : -1e300;
        gl = -wave_max(-lo_j);
        gu = wave_max(hi_j);
        nrmT = (fabs(gl) > fabs(gu)) ? fabs(gl) : fabs(gu);
    }
    *kept_out = 0;
    if (!(nrmT > tol) || !(nrmT < 1e300)) return 0.0;                        // wave-uniform (nothing above the tolerance, or non-finite data)
    const double scl = trid_pow2_floor(nrmT);                                // exact scaling: |T| / scl in [1, 2)
    const double isc = 1.0 / scl;
    // padding to TRID_MAX rows: d = 4, e = 0 (scaled units, where |T| < 2): decoupled eigenvalues at 4, above every shift, so counts, brackets and
    // twists are those of T and the unrolled recurrences need no guard on n
    const double dsr = (lane < n) ? dreg * isc : 4.0, esr = (lane + 1 < n) ? ereg * isc : 0.0, e2r = esr * esr;
    const double tols = tol * isc, gls = gl * isc, gus = gu * isc;
    // kept eigenvalues: lambda < -tol (c_neg of them) and lambda > tol (n - c_pos)
    const double tol_up = __longlong_as_double(__double_as_longlong(tols) + 1);
    const int cq = trid_count_fast(dsr, e2r, n, (lane & 1) ? tol_up : -tols);
    const int c_neg = wave_bcast_i(cq, 0), c_pos = wave_bcast_i(cq, 1);
    const int m_neg = c_neg, m_pos = n - c_pos, kept = m_neg + m_pos;
    *kept_out = kept;
    if (kept == 0) return 0.0;
    // ---- step 2: one Sturm count per lane on geometrically spaced shifts ----
    int Lp = (m_pos == 0) ? 0 : ((m_neg == 0) ? 64 : (64 * m_pos + kept / 2) / kept);
    if (m_pos > 0 && Lp < 2) Lp = 2;
    if (m_neg > 0 && Lp > 62) Lp = 62;
    const int Ln = 64 - Lp;
    {
        const bool pos = lane < Lp;
        const int t = pos ? lane : lane - Lp, L = pos ? Lp : Ln;
        const double top0 = (pos ? gus : -gls) * (1.0 + 1e-12) + 1e-300;
        const double top = (top0 > tols) ? top0 : tols * 2.0;
        const double lg = log2(top / tols) / (double)((L > 1) ? L - 1 : 1);
        const double mag = (t == L - 1) ? top : tols * exp2(lg * (double)t);
        const double sh = pos ? mag : -mag;
        const int c = trid_count_fast(dsr, e2r, n, sh);
        shS[lane] = sh;
        cntS[lane] = (double)c;
    }
    wave_sync();
    const bool own = lane < kept;                                            // lanes 0 .. m_neg-1: negative side (ascending), then the positive side
    const bool neg_side = lane < m_neg;
    const int idx = neg_side ? lane : c_pos + (lane - m_neg);                // number of the lane's eigenvalue (ascending, 0-based)
    double lo = 0.0, hi = 0.0;
    int clo = 0, chi = 0;
    {
        int jn = -1, jp = -1;
#pragma unroll 1
        for (int t = 0; t < 64; ++t) {                                       // (one loop for both sides: no divergent control flow around it)
            const int c = (int)cntS[t];
            if (t >= Lp) { if (jn < 0 && c <= idx) jn = t - Lp; }
            else if (jp < 0 && c > idx) jp = t;
        }
        const int jj = neg_side ? ((jn < 0) ? Ln - 1 : jn) : ((jp < 0) ? ((Lp > 0) ? Lp - 1 : 0) : jp);
        const int at = neg_side ? Lp + jj : jj;                              // table entry of the bracket's far end
        const double far_v = shS[at & 63]; const int far_c = (int)cntS[at & 63];
        const double near_v = (jj > 0) ? shS[(at - 1) & 63] : (neg_side ? -tols : tols);
        const int near_c = (jj > 0) ? (int)cntS[(at - 1) & 63] : (neg_side ? c_neg : c_pos);
        if (neg_side) { lo = far_v; clo = far_c; hi = near_v; chi = near_c; }   // first shift -mag_j (descending) with count <= idx
        else { hi = far_v; chi = far_c; lo = near_v; clo = near_c; }            // first shift +mag_j (ascending) with count > idx
    }
    wave_sync();
#pragma unroll 1
    for (int round = 0; round < 60; ++round) {                               // bisection until every bracket holds exactly one eigenvalue
        const bool iso = !own || (chi - clo) == 1;
        if (round >= 3 && !wave_any(!iso)) break;
        const double mid = 0.5 * (lo + hi);
        const int c = trid_count_fast(dsr, e2r, n, mid);
        const bool right = c <= idx;                                         // eigenvalue idx is >= mid
        lo = right ? mid : lo; clo = right ? c : clo;
        hi = right ? hi : mid; chi = right ? chi : c;
    }
    phase_stamp(dbg, 29);
    // ---- step 3: Rayleigh-quotient iteration on the minors, one eigenpair per lane ----
    double sig = 0.5 * (lo + hi), lam = sig, nz2 = 1.0;
    bool done = !own, risky = false;
    lds_ptr Zt = work;                                                       // Zt[j * TRID_LD + lane]: the lane's eigenvector once converged (the tables are dead)
    const int zl = (lane < TRID_MAX) ? lane : 0;
    int it = 0;
#pragma unroll 1
    for (it = 0; it < 48; ++it) {
        if (!wave_any(!done)) break;
        // (every lane computes; `done` lanes keep their results through the selects at the end: the readlanes need all lanes)
        double amin = 1e300, amax = 0.0;
        double Pn[TRID_MAX + 1];                                             // leading minors p_0 .. p_n, then the eigenvector in place
        Pn[0] = 1.0;
        Pn[1] = wave_bcast(dsr, 0) - sig;
        bool sc_ = (Pn[1] < 0.0) || (Pn[1] == 0.0);
        int negc = sc_ ? 1 : 0;
#pragma unroll
        for (int j = 1; j < TRID_MAX; ++j) {
            const double pn = (wave_bcast(dsr, j) - sig) * Pn[j] - wave_bcast(e2r, j - 1) * Pn[j - 1];
            const bool sn = (pn < 0.0) || (pn == 0.0 && !sc_);
            negc += (sn != sc_) ? 1 : 0;
            sc_ = sn;
            Pn[j + 1] = pn;
            const double ap = fabs(pn);
            amin = (ap < amin && ap > 0.0) ? ap : amin;
            amax = (ap > amax) ? ap : amax;
        }
        const double detv = Pn[TRID_MAX];                                    // determinant of the padded matrix (the padding's factor cancels below)
        // trailing minors, twist index r = argmax |p_k q_(k+1)|
        double qa = 1.0, qb = 0.0;                                           // q_(k+1), q_(k+2)
        double best = -1.0, pr = 1.0, qr1 = 1.0;
        int r = 0;
#pragma unroll
        for (int k = TRID_MAX - 1; k >= 0; --k) {
            if (lane < TRID_MAX) Qs[(k + 1) * TRID_LD + zl] = qa;            // q_(k+1)
            const double m = (k < n) ? fabs(Pn[k] * qa) : -1.0;              // (the twist stays inside T)
            const bool better = m > best;
            best = better ? m : best; r = better ? k : r; pr = better ? Pn[k] : pr; qr1 = better ? qa : qr1;
            const double qk = (wave_bcast(dsr, k) - sig) * qa - wave_bcast(e2r, k) * qb;
            qb = qa; qa = qk;
            const double aq = fabs(qk);
            amin = (aq < amin && aq > 0.0) ? aq : amin;
            amax = (aq > amax) ? aq : amax;
        }
        const double gam = detv / (pr * qr1);                                // gamma_r = det / (p_r q_(r+1))
        // eigenvector, in place of the minors
        double zc = 1.0 / pr, s2 = 1.0;
#pragma unroll
        for (int i = TRID_MAX - 1; i >= 0; --i) {                            // below the twist: z_i = (-1)^(r-i) e_i .. e_(r-1) p_i / p_r
            const double ei = wave_bcast(esr, i);
            const bool below = i < r;
            zc = below ? -zc * ei : zc;
            const double zi = below ? zc * Pn[i] : ((i == r) ? 1.0 : 0.0);
            s2 += below ? zi * zi : 0.0;
            Pn[i] = zi;
        }
        zc = 1.0 / qr1;
#pragma unroll
        for (int i = 1; i < TRID_MAX; ++i) {                                 // above: z_i = (-1)^(i-r) e_r .. e_(i-1) q_(i+1) / q_(r+1)  (0 from row n on: e_(n-1) = 0)
            const double ei = wave_bcast(esr, i - 1);
            const bool above = i > r;
            const double q1 = Qs[(i + 1) * TRID_LD + zl];
            zc = above ? -zc * ei : zc;
            const double zi = above ? zc * q1 : Pn[i];
            s2 += above ? zi * zi : 0.0;
            Pn[i] = zi;
        }
        const double corr = gam / s2;
        const double resid = fabs(gam) * rsqrt(s2);
        const bool right = negc <= idx;                                      // eigenvalue idx is >= sig
        const bool bad_range = !(amin > 1e-250) || !(amax < 1e250) || !(best > 0.0) || !(s2 == s2);
        const bool conv = (resid <= 1.6e-15) || (fabs(corr) <= 4e-16 * fabs(sig));   // (scaled units: |T| < 2)
        const bool upd = !done;
        lo = (upd && right) ? sig : lo;
        hi = (upd && !right) ? sig : hi;
        risky = risky || (upd && bad_range);
        const bool fin = upd && (conv || bad_range);
        lam = fin ? sig + corr : lam;
        nz2 = fin ? s2 : nz2;
        if (fin) {                                                           // (own lanes only: lane < kept <= TRID_MAX)
#pragma unroll
            for (int i = 0; i < TRID_MAX; ++i) Zt[i * TRID_LD + zl] = Pn[i];
        }
        const double cand = sig + corr;
        sig = (upd && !fin) ? ((cand >= lo && cand <= hi) ? cand : 0.5 * (lo + hi)) : sig;
        done = done || fin;
        wave_sync();                                                         // (Qs is rewritten by the next round)
    }
    if (wave_any(!done) || wave_any(risky && own)) *fail = 1;
    phase_stamp(dbg, 30);
    if (dbg && lane == 0) dbg[79] = (double)it;
    wave_sync();
    {                                                                        // guard for clustered kept eigenvalues: see wave_trid_pinv
        const bool pair = own && lane + 1 < kept;
        double dotn = 0.0;
        const int zn = pair ? zl + 1 : zl;
#pragma unroll 4
        for (int j = 0; j < n; ++j) dotn += Zt[j * TRID_LD + zl] * Zt[j * TRID_LD + zn];
        const double nz2n = wave_shfl(nz2, (lane + 1) & 63);
        if (wave_any(pair && !(dotn * dotn <= 1e-18 * nz2 * nz2n))) *fail = 1;
    }
    // ---- step 4: coefficients, x^ ----
    if (own) {
        double dot = 0.0;
#pragma unroll 4
        for (int j = 0; j < n; ++j) dot += Zt[j * TRID_LD + zl] * bh[j * ldb];
        const double coef = dot / (lam * scl * nz2);                         // lambda in the units of T
#pragma unroll 4
        for (int j = 0; j < n; ++j) Zt[j * TRID_LD + zl] *= coef;
    }
    wave_sync();
    double x = 0.0;
    if (lane < n) {
#pragma unroll 4
        for (int l = 0; l < kept; ++l) x += Zt[lane * TRID_LD + l];
    }
    wave_sync();
    return x;
}

// MINOR_FORM: steps 2 - 4 in the division-free minor form (wave_trid_pinv_fast) instead of the pivot form (wave_trid_pinv).
template <bool IN_REGISTERS = true, bool MINOR_FORM = false>
__device__ __forceinline__ void wave_pinv_solve_trid(double* A_, const int lda, const int n, const double tol, double* sol, double* small_, double* work_,
                                                     int* kept_out, int* fail, double* dbg = nullptr) {
    const lds_ptr A = to_lds(A_), small = to_lds(small_), work = to_lds(work_);
    const int lane = lane_id();
    lds_ptr dS = small, eS = small + TRID_MAX, e2S = small + 2 * TRID_MAX;
    double dreg, ereg, x;
    if constexpr (IN_REGISTERS) {
        double a[TRID_MAX];
        {
            const bool mine = lane <= n;
            const int rl = mine ? lane : 0;
#pragma unroll
            for (int c = 0; c < TRID_MAX; ++c) a[c] = (mine && c < n) ? A[c * lda + rl] : 0.0;
        }
        wave_tridiag_reg<TRID_MAX>(a, n, dreg, ereg);
        if (lane < n) { dS[lane] = dreg; eS[lane] = ereg; e2S[lane] = ereg * ereg; }
        if (lane <= n) {                                                     // reflector k -> A[k][k+1 ..) with beta_k in A[k][k], transformed right-hand side -> column n:
#pragma unroll                                                               // the layout of the LDS version; the 64 registers are free for the eigen-solve
            for (int c = 0; c < TRID_MAX; ++c) if (c < n && (lane >= c || lane == n)) A[c * lda + lane] = a[c];
        }
        wave_sync();
        phase_stamp(dbg, 28);
        if constexpr (MINOR_FORM) {
            x = wave_trid_pinv_fast(dreg, ereg, n, A + n, lda, tol, work, kept_out, fail, dbg);
            if (wave_uniform_i(*fail)) x = wave_trid_pinv(dS, eS, e2S, n, A + n, lda, tol, work, kept_out, fail, nullptr);   // minors out of range: the pivot form
        } else {
            x = wave_trid_pinv(dS, eS, e2S, n, A + n, lda, tol, work, kept_out, fail, dbg);
        }
        phase_stamp(dbg, 31);
#pragma unroll 2
        for (int k = n - 3; k >= 0; --k) {                                   // x = H_0 ... H_(n-3) x^
            const double beta = A[k * lda + k];
            const double v = (lane > k && lane < n) ? A[k * lda + lane] : 0.0;
            const double s = wave_sum(v * x);
            x -= beta * s * v;
        }
    } else {
        wave_tridiag_burst(A, lda, n, dreg, ereg);
        if (lane < n) { dS[lane] = dreg; eS[lane] = ereg; e2S[lane] = ereg * ereg; }
        wave_sync();
        phase_stamp(dbg, 28);
        x = wave_trid_pinv(dS, eS, e2S, n, A + n, lda, tol, work, kept_out, fail, dbg);
        phase_stamp(dbg, 31);
#pragma unroll 1
        for (int k = n - 3; k >= 0; --k) {                                   // x = H_0 ... H_(n-3) x^
            const double beta = A[k * lda + k];
            const double v = (lane > k && lane < n) ? A[k * lda + lane] : 0.0;
            const double s = wave_sum(v * x);
            x -= beta * s * v;
        }
    }
    if (lane < n) sol[lane] = x;
    wave_sync();
}

}  // namespace tff
