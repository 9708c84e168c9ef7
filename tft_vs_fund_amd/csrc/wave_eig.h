// Wavefront-cooperative symmetric eigen-solvers (n <= 32: one lane per row).
// Used for the 27x27 and 15x15 Gram matrices of linearTFT (linearTFT.m:64-67
// and :84, where the reference runs a full svd(A) of the 4N x 27 / 4N x 15
// design matrix and keeps V(:,end)) and the 9x9 Gram matrix of linearF
// (linearF.m:54-55).
//
//   wave_min_eigvec_reg<n> : lane r owns row r of the matrix IN REGISTERS.
//       Right-looking Cholesky of G + delta*I with v_readlane broadcasts of the
//       pivot column: no LDS traffic and no memory latency in the dependency
//       chain.  Fully unrolled (compile-time register indices): the kernels are
//       VALU-issue-bound (a lone wave retires ~1 fp64 instruction per 8 cycles),
//       so executed instruction count is what matters -- a rolled variant with a
//       sliding register window was 6 KB instead of 25 KB of code but executed
//       twice the instructions and ran 1.5x slower.  The factor is stored once
//       to LDS in packed form; inverse iteration then reads the lane's own row
//       (forward solve) and own column (backward solve) back into registers.
//       Typical 4..6 iterations (sigma_27/sigma_26 ~ 0.02).
//   wave_jacobi_min_eigvec : cyclic two-sided Jacobi on a full n x n matrix and
//       its eigenvector matrix in LDS -- the gap-independent solver.  Lives in
//       the LDS-heavy kernel variant only (fix-up pass for triplets whose
//       inverse iteration did not converge, and TFF_OPT_SOLVER = 1).
#pragma once
#include "wave.h"

namespace tff {

// (Round 5, measured and dropped: leaving an inverse iteration early when its observed rate says the cap will be hit anyway -- config 4's seven-point
// samples, where ~0.3 % of the rows run into the cap and hold their wavefront for 300 iterations.  Same-box A/B, tools/ab_libs_config4.py: 26.94 ->
// 30.41 ms per million hypotheses, and hypotheses that used to converge in-row after 100 - 250 iterations changed hands: the rate is not monotone
// on these clustered spectra, and the one-triplet exact kernel that takes the hand-over costs more than the iterations saved.)


__device__ __forceinline__ int tri_index(int r, int c) { return (r * (r + 1)) / 2 + c; }   // packed lower, c <= r

// Inverse iteration with a triangular factor held in LDS in ROW-SCALED form: Lp (n x n, row-major, ld = n) holds
// L' = D^-1 L (unit diagonal, D = diag(L)) with zeros on and above the diagonal, myinv = 1 / L[lane][lane]; the iterated
// matrix is L L' (= G + delta I after a Cholesky factorisation, = A'A with L = R' after a QR factorisation of A).
// The substitutions then need no division, no masked load and no per-step select --
//   L y = x   <=>  L' y = D^-1 x,          L^T z = y  <=>  L'^T (D z) = y,
// one multiplication by 1 / L_jj before the forward and one after the backward sweep instead of one per step, and a dependent
// chain of readlane + fma per step.
// On return lane r (< n) holds component r of the unit eigenvector of the smallest eigenvalue; *iters = iterations used,
// *resid2 = 0 when the iteration converged, else the last squared step.
// has_start / start: optional initial guess (component `lane` on lane `lane`), default the uniform vector.
// gap_risk (optional, two doubles): numerator and denominator of an upper estimate of 1 / (lambda_(n-1) - lambda_n)^2 from the
// observed convergence rate and the final Rayleigh quotient -- sqrt(num / den) eps |G| bounds the rounding error of an
// eigenvector taken from a FORMED Gram matrix (the squared conditioning the reference's svd(A) does not have); callers that factor
// G route a triplet to the QR-based exact path when it is large.  num = 0 when the rate could not be observed (converged in one step).
template <int n, int G = 64>
__device__ inline double wave_invit_unit(const double* Lp, const double myinv, const int maxit, int* iters, double* resid2,
                                         const bool has_start = false, const double start = 0.0, double* gap_risk = nullptr) {
    using Grp = Group<G>;
    const int lane = Grp::lane();
    const int rl = (lane < n) ? lane : 0;
    double x = (lane < n) ? rsqrt((double)n) : 0.0;
    if (has_start) {                                        // caller's guess (lane r: component r); a zero / non-finite guess falls back
        const double s0 = (lane < n) ? start : 0.0;
        const double nn0 = Grp::sum(s0 * s0);
        if (nn0 > 1e-300 && nn0 < 1e300) x = s0 * rsqrt(nn0);
    }
    double rprev2 = 1.0, res = 1.0, rk_r2 = 0.0, rk_rp = 1.0, rk_nn = 0.0;
    int it = 0;
    bool done = false;                                      // per group; the loop itself is wave-uniform
    // the lane's own row and column of L' stay in registers for all iterations (2 n doubles; the factor is read from LDS once)
    double row[n], col[n];                                  // L'[lane][j] (0 for j >= lane);  L'[j][lane] (0 for j <= lane)
#pragma unroll
    for (int j = 0; j < n; ++j) { row[j] = Lp[rl * n + j]; col[j] = Lp[j * n + rl]; }
#pragma unroll 1
    while (true) {
        double y = x * myinv;
#pragma unroll
        for (int j = 0; j < n; ++j) {                       // forward  L' y = D^-1 x
            const double yj = Grp::bcast(y, j);
            y -= row[j] * yj;
        }
#pragma unroll
        for (int j = n - 1; j >= 0; --j) {                  // backward L'^T u = y,  u = D z
            const double uj = Grp::bcast(y, j);
            y -= col[j] * uj;
        }
        y *= myinv;
        if (lane >= n) y = 0.0;
        const double nn = Grp::sum(y * y);
        const double dot = Grp::sum(y * x);
        const double rn = rsqrt(nn);
        const double yn = y * ((dot < 0.0) ? -rn : rn);
        const double dd = yn - x;
        const double r2 = Grp::sum(dd * dd);
        if (!done) {
            x = yn;
            ++it;
            // |step|^2 = r2; the error of the new iterate is ~ rho |step| / (1 - rho) with rho ~ |step| / |previous step|
            if (r2 <= 1e-26) { res = 0.0; done = true; }                                     // converged: stopped moving
            else if (it >= 2 && r2 < 0.25 * rprev2 && r2 * r2 < 1e-26 * rprev2) { res = 0.0; done = true; }   // predicted error < 1e-13
            else if (!(r2 == r2) || it >= maxit) { res = (r2 == r2) ? r2 : 1.0; done = true; }             // NaN / iteration cap: not converged
            if (it >= 2 && r2 > 1e-30) { rk_r2 = r2; rk_rp = rprev2; rk_nn = nn; }            // last observable pair of steps
            rprev2 = r2;
        }
        if (!wave_any(!done)) break;
    }
    *iters = it;
    *resid2 = res;                                                              // 0 when converged, last |step|^2 otherwise
    if (gap_risk) {
        // rate rho = (lambda_n + delta) / (lambda_(n-1) + delta) ~ sqrt(r2 / rprev2), lambda_n + delta ~ 1 / |y|:
        // 1 / gap ~ |y| rho / (1 - rho).  Returned SQUARED and without sqrt / division on the hot path: with q = rho^2,
        // 1 - rho >= (1 - q) / 2, so  (1 / gap)^2 <= 4 |y|^2 q / (1 - q)^2 =: risk2_num / risk2_den (an upper bound: flags slightly early).
        gap_risk[0] = 4.0 * rk_nn * rk_r2 * rk_rp;                          // numerator   x rk_rp^2
        const double d = rk_rp - rk_r2;                                      // (1 - q) rk_rp
        gap_risk[1] = (rk_r2 < rk_rp) ? d * d : 0.0;                         // denominator x rk_rp^2 (0: no gap observed)
    }
    return x;
}

// g[c] = G[lane][c] for c <= lane (entries c > lane are ignored), diag = G[lane][lane].
// Lp: n*n doubles of LDS.  Cholesky of G + delta I in registers, then wave_invit_unit (see there for the outputs).
// *gram_risk (optional): 1.0 when the eps-free error amplification |G| / (lambda_(n-1) - lambda_n) of the eigenvector of the formed
// Gram matrix may exceed sqrt(gram_risk_limit2), else 0.0.
template <int n, int G = 64>
__device__ inline double wave_min_eigvec_reg(double (&g)[n], const double diag, double* Lp, const int maxit,
                                             int* iters, double* resid2, const bool has_start = false, const double start = 0.0,
                                             double* gram_risk = nullptr, const double gram_risk_limit2 = 1e14) {
    using Grp = Group<G>;                                   // one lane group per matrix (the whole wave, or one half of it)
    const int lane = Grp::lane();
    const double tr = Grp::sum(lane < n ? diag : 0.0);
    const double delta = 1e-14 * tr;
    const double pfloor = 1e-3 * delta + 1e-300;
#pragma unroll
    for (int c = 0; c < n; ++c) g[c] += (c == lane) ? delta : 0.0;
    double myinv = 0.0;                                     // 1 / L[lane][lane]
#pragma unroll
    for (int k = 0; k < n; ++k) {
        double d = Grp::bcast(g[k], k);                     // pivot (lane k's diagonal, fully updated)
        d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
        const double rs = rsqrt_pos(d);
        g[k] = (lane == k) ? d * rs : g[k] * rs;            // column k of L (rows >= k meaningful)
        myinv = (lane == k) ? rs : myinv;
#pragma unroll
        for (int c = k + 1; c < n; ++c) {
            const double lck = Grp::bcast(g[k], c);         // L[c][k]
            g[c] -= g[k] * lck;                             // L[r][c] -= L[r][k] L[c][k]   (meaningful for r >= c)
        }
    }
    // store the row-scaled factor (see wave_invit_unit) once; the iteration reads the lane's own row / column back into registers
    wave_sync();
    if (lane < n) {
#pragma unroll
        for (int c = 0; c < n; ++c) Lp[lane * n + c] = (c < lane) ? g[c] * myinv : 0.0;
    }
    wave_sync();
    double risk[2] = {0.0, 1.0};
    const double x = wave_invit_unit<n, G>(Lp, myinv, maxit, iters, resid2, has_start, start, gram_risk ? risk : nullptr);
    // |G| / gap < limit  <=>  tr^2 num < limit^2 den
    if (gram_risk) *gram_risk = (tr * tr * risk[0] < gram_risk_limit2 * risk[1]) ? 0.0 : 1.0;
    return x;
}
__device__ __forceinline__ bool eig_converged(double resid2) { return resid2 == 0.0; }

// Cyclic Jacobi on the symmetric n x n matrix A (leading dimension lda, full storage, DESTROYED:
// its diagonal ends up holding the eigenvalues) with the eigenvectors accumulated in the columns of
// V (leading dimension ldv).  n <= 64 (one lane per row).  Rotation (p,q): the lanes update the
// column pair of A and V, then the row pair of A.  Works for indefinite matrices (KKT systems).
__device__ inline int wave_jacobi_sym(double* A, const int lda, double* V, const int ldv, const int n, const bool kkt = false) {
    const int lane = lane_id();
    for (int e = lane; e < n * n; e += WAVE) V[(e / n) * ldv + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    const double absfloor = 1e-22 * wave_sum(lane < n ? fabs(A[lane * lda + lane]) : 0.0) + 1e-300;
    wave_sync();
    int sweep = 0;
#pragma unroll 1
    for (; sweep < 40; ++sweep) {
        int rotations = 0;
#pragma unroll 1
        for (int p = 0; p < n - 1; ++p) {
#pragma unroll 1
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * lda + q], app = A[p * lda + p], aqq = A[q * lda + q];
                // relative threshold (de Rijk): keeps small eigenvalues accurate,
                // plus an absolute floor so rounding noise under a zero eigenvalue is not chased
                // kkt: singular indefinite matrices (pseudo-inverse of a KKT system): threshold relative to the larger diagonal entry,
                // otherwise couplings between a large eigenvalue and the null space are rotated for ever at rounding level
                const double ref = kkt ? ((fabs(app) > fabs(aqq)) ? fabs(app) : fabs(aqq)) : sqrt(fabs(app * aqq));
                if (!(fabs(apq) > 1.1e-16 * ref && fabs(apq) > absfloor)) continue;   // wave-uniform
                ++rotations;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = rsqrt(1.0 + t * t), s = t * c;
                wave_sync();
                if (lane < n) {                               // columns p,q of A and V
                    const double arp = A[lane * lda + p], arq = A[lane * lda + q];
                    A[lane * lda + p] = c * arp - s * arq;
                    A[lane * lda + q] = s * arp + c * arq;
                    const double vrp = V[lane * ldv + p], vrq = V[lane * ldv + q];
                    V[lane * ldv + p] = c * vrp - s * vrq;
                    V[lane * ldv + q] = s * vrp + c * vrq;
                }
                wave_sync();
                if (lane < n) {                               // rows p,q of A
                    const double apr = A[p * lda + lane], aqr = A[q * lda + lane];
                    A[p * lda + lane] = c * apr - s * aqr;
                    A[q * lda + lane] = s * apr + c * aqr;
                }
                wave_sync();
            }
        }
        if (rotations == 0) break;
    }
    return sweep;
}

// Eigen-decomposition of the symmetric n x n matrix A (LDS, full storage, leading dimension lda, DESTROYED; columns >= n of the
// array are not touched) by ONE wavefront, n <= 64: Householder reduction to tridiagonal form followed by the implicit-shift QL
// iteration with the rotations accumulated into the eigenvectors -- ~10x fewer operations than cyclic Jacobi on the 39 x 39 / 38 x 38
// KKT matrices whose pseudo-inverse the Gauss-Helmert models with redundant constraints need, and no barriers.
//   * reduction: step k annihilates column k below the sub-diagonal; lane r owns COLUMN r of the (symmetric) trailing block, so
//     every LDS access is A[c * lda + lane] (consecutive lanes, consecutive addresses) with the Householder vector read as a
//     broadcast; the reflections are accumulated into Z on the fly (lane r owns row r of Z, stored transposed);
//   * the reduction starts at the top-left corner and the QL sweeps run on the index-reversed tridiagonal matrix: the KKT matrices
//     are graded from ~1e13 (normal equations, top-left) down to ~1 (constraints), the ordering for which this pair is accurate;
//   * QL: the tridiagonal entries live one per lane in registers (v_readlane for the wave-uniform recurrences), each rotation costs
//     one LDS read and one write per lane (the column carried to the next rotation stays in a register).
// On return lane j (< n) holds eigenvalue j and ROW eig_row(n, j) of ZT (leading dimension ldz) its unit eigenvector.
// scr: 2 n doubles of LDS.  *fail = 1 if an eigenvalue needed more than 60 sweeps (never observed).
__device__ __forceinline__ int eig_row(int n, int j) { return n - 1 - j; }
__device__ __forceinline__ double wave_eigh_ql(double* A_, const int lda, double* ZT_, const int ldz, const int n, double* scr_, int* fail) {
    const lds_ptr A = to_lds(A_), ZT = to_lds(ZT_), scr = to_lds(scr_);
    const int lane = lane_id();
    const int rl = (lane < n) ? lane : 0;
    for (int e = lane; e < n * n; e += WAVE) ZT[(e / n) * ldz + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    double dreg = 0.0, ereg = 0.0;                                           // lane k: T[k][k], T[k][k+1]
    wave_sync();
#pragma unroll 1
    for (int k = 0; k + 2 < n; ++k) {
        const bool act = lane > k && lane < n;
        const double x = act ? A[k * lda + lane] : 0.0;                      // column k below the diagonal (= row k by symmetry)
        if (lane == k) dreg = A[k * lda + k];
        const double x1 = wave_bcast(x, k + 1);
        const double tail = wave_sum((lane > k + 1) ? x * x : 0.0);
        if (wave_uniform_i(tail == 0.0)) {                                   // already tridiagonal in this column
            if (lane == k) ereg = x1;
            continue;
        }
        const double sigma = tail + x1 * x1;
        const double nrm = sqrt(sigma);
        const double alpha = (x1 > 0.0) ? -nrm : nrm;
        const double v = (lane == k + 1) ? x - alpha : x;                    // Householder vector (0 on lanes <= k)
        const double beta = 1.0 / (sigma + fabs(x1) * nrm);                  // 2 / v'v
        if (lane == k) ereg = alpha;
        if (lane < n) scr[lane] = v;
        wave_sync();
        double p = 0.0;
#pragma unroll 4
        for (int c = k + 1; c < n; ++c) p += A[c * lda + rl] * scr[c];
        p = act ? p * beta : 0.0;
        const double K = 0.5 * beta * wave_sum(p * v);
        const double q = p - K * v;
        if (lane < n) scr[n + lane] = q;
        wave_sync();
        if (act) {
#pragma unroll 4
            for (int c = k + 1; c < n; ++c) A[c * lda + lane] -= scr[c] * q + scr[n + c] * v;
        }
        double t = 0.0;                                                      // Z <- Z H_k, row `lane` of Z
#pragma unroll 4
        for (int c = k + 1; c < n; ++c) t += ZT[c * ldz + rl] * scr[c];
        t *= beta;
        if (lane < n) {
#pragma unroll 4
            for (int c = k + 1; c < n; ++c) ZT[c * ldz + lane] -= t * scr[c];
        }
        wave_sync();
    }
    if (n >= 2) {
        if (lane == n - 2) { dreg = A[(n - 2) * lda + n - 2]; ereg = A[(n - 2) * lda + n - 1]; }
    }
    if (lane == n - 1) dreg = A[(n - 1) * lda + n - 1];
    // index reversal: logical j = n - 1 - (physical index)
    if (lane < n) { scr[lane] = dreg; scr[n + lane] = ereg; }
    wave_sync();
    double dq = (lane < n) ? scr[n - 1 - lane] : 0.0;
    double eq = (lane + 1 < n) ? scr[n + n - 2 - lane] : 0.0;               // couples logical lane, lane + 1
    wave_sync();
    int failed = 0;
#pragma unroll 1
    for (int l = 0; l < n; ++l) {
        int iter = 0;
#pragma unroll 1
        while (true) {
            if (lane < n) scr[lane] = fabs(dq);
            wave_sync();
            const double dd = fabs(dq) + ((lane + 1 < n) ? scr[lane + 1] : 0.0);
            wave_sync();
            int m = wave_first_lane(lane >= l && lane + 1 < n && fabs(eq) <= 2.220446049250313e-16 * dd);
            m = wave_uniform_i((m > n - 1) ? n - 1 : m);
            if (m == l) break;
            if (++iter > 60) { failed = 1; break; }
            const double dl = wave_bcast(dq, l), el = wave_bcast(eq, l);
            double g = (wave_bcast(dq, l + 1) - dl) / (2.0 * el);
            const double rr = sqrt(g * g + 1.0);
            g = wave_bcast(dq, m) - dl + el / (g + ((g >= 0.0) ? rr : -rr));
            double s = 1.0, c = 1.0, p = 0.0;
            double dnext = wave_bcast(dq, m);                                // d[i + 1] (not yet rewritten by this sweep)
            double carry = ZT[eig_row(n, m) * ldz + rl];                     // logical column i + 1 of Z, row `lane`
#pragma unroll 2
            for (int i = m - 1; i >= l; --i) {
                const double ei = wave_bcast(eq, i), di = wave_bcast(dq, i);
                const double zi = ZT[eig_row(n, i) * ldz + rl];
                const double f = s * ei, b = c * ei;
                const double h = f * f + g * g;
                // h == 0 (underflow): the bulge is gone; the identity rotation (s = 0, c = 1, r = 0) decouples the block here and the
                // rest of the sweep degenerates to sign flips
                const bool zero = h == 0.0;
                const double rinv = rsqrt(zero ? 1.0 : h);
                eq = (lane == i + 1) ? h * rinv : eq;                        // r
                s = f * rinv;
                c = zero ? 1.0 : g * rinv;
                g = dnext - p;
                const double r2 = (di - g) * s + 2.0 * c * b;
                p = s * r2;
                dq = (lane == i + 1) ? g + p : dq;
                g = c * r2 - b;
                if (lane < n) ZT[eig_row(n, i + 1) * ldz + lane] = s * zi + c * carry;
                carry = c * zi - s * carry;
                dnext = di;
            }
            if (lane < n) ZT[eig_row(n, l) * ldz + lane] = carry;
            dq = (lane == l) ? dnext - p : dq;
            eq = (lane == l) ? g : eq;
            eq = (lane == m) ? 0.0 : eq;
        }
    }
    wave_sync();
    *fail = failed;
    return dq;
}

// Out-of-line copy for the single-wavefront kernels (rare fall-back there; inlined it would cost them a wave of occupancy).
__device__ __attribute__((noinline)) double wave_eigh_ql_call(double* A, const int lda, double* ZT, const int ldz, const int n, double* scr, int* fail) {
    return wave_eigh_ql(A, lda, ZT, ldz, n, scr, fail);
}

// eigenvector of the smallest eigenvalue (see wave_jacobi_sym); per lane r < n component r
__device__ inline double wave_jacobi_min_eigvec(double* A, double* V, const int n, const int ld, int* sweeps_out) {
    const int lane = lane_id();
    *sweeps_out = wave_jacobi_sym(A, ld, V, ld, n);
    int best = 0;                                             // smallest diagonal entry (wave-uniform scan)
    double bv = A[0];
    for (int k = 1; k < n; ++k) { const double d = A[k * ld + k]; if (d < bv) { bv = d; best = k; } }
    double x = (lane < n) ? V[lane * ld + best] : 0.0;
    const double nn = wave_sum(x * x);
    return x * rsqrt(nn);
}

}  // namespace tff
