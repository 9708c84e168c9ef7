// Per-lane dense linear algebra on 3x3 / 4x4 problems, everything in
// registers with compile-time indices (fully unrolled): the building blocks
// of epipole extraction, essential-matrix decomposition and DLT triangulation
// (one lane per problem).
#pragma once
#include "wave.h"

namespace tff {

struct Mat3 { double m[3][3]; };

__device__ __forceinline__ double sgn(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }   // MATLAB sign()

__device__ __forceinline__ Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return c;
}
__device__ __forceinline__ Mat3 mat3_T(const Mat3& a) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[j][i];
    return c;
}
__device__ __forceinline__ double mat3_det(const Mat3& a) {
    return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1])
         - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0])
         + a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
// inverse by adjugate (the reference's inv() of 3x3 calibration / normalisation matrices)
__device__ __forceinline__ Mat3 mat3_inv(const Mat3& a) {
    Mat3 c;
    c.m[0][0] = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    c.m[0][1] = a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2];
    c.m[0][2] = a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1];
    c.m[1][0] = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    c.m[1][1] = a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0];
    c.m[1][2] = a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2];
    c.m[2][0] = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    c.m[2][1] = a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1];
    c.m[2][2] = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
    const double id = 1.0 / (a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[1][0] + a.m[0][2] * c.m[2][0]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.m[i][j] *= id;
    return c;
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// 4 x 4 determinant by 2 x 2 minors (TFT_from_P.m:25-33 takes 27 of them)
__device__ __forceinline__ double det4(const double (&m)[4][4]) {
    const double s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1], s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    const double s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3], s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    const double s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3], s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    const double c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const double c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const double c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
}

// Inverse iteration x <- (L L')^-1 x, normalised, from the unit start x, until the iterate stops moving (the loop of spd_min_eigvec, for
// callers that hold the Cholesky factor already: pose_common.h::vote_one's).  inv[i] = 1 / L[i][i].  Returns the iterations used.
template <int n>
__device__ __forceinline__ int chol_invit(const double (&L)[n][n], const double (&inv)[n], double (&x)[n], const int maxit, bool* converged) {
    bool conv = false;
    double rprev2 = 1.0;
    int it = 0;
#pragma clang loop unroll(disable)
    for (; it < maxit;) {
        double y[n];
        // forward L y = x
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = x[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
            y[i] = s * inv[i];
        }
        // backward L' z = y (in place)
#pragma unroll
        for (int i = n - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < n; ++k) s -= L[k][i] * y[k];
            y[i] = s * inv[i];
        }
        double nn = 0.0, dot = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { nn += y[i] * y[i]; dot += y[i] * x[i]; }
        const double rn = rsqrt(nn);
        const double sg = (dot < 0.0) ? -rn : rn;
        double r2 = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { const double yi = y[i] * sg; const double d = yi - x[i]; r2 += d * d; x[i] = yi; }
        ++it;
        // r2 = |step|^2.  Error of the new iterate ~ rho*|step|/(1-rho), rho ~ |step|/|previous step|.
        if (!(r2 > 1e-28)) { conv = (r2 == r2); break; }                 // also leaves on NaN
        if (it >= 2 && r2 < 0.25 * rprev2 && r2 * r2 < 1e-26 * rprev2) { conv = true; break; }
        rprev2 = r2;
    }
    if (converged) *converged = conv;
    return it;
}

// --------------------------------------------------------------------------
// Eigenvector of the smallest eigenvalue of a symmetric positive
// semi-definite n x n matrix S (n = 3, 4): Cholesky of S + delta*I followed by
// inverse iteration.  The shift only conditions the factorisation (it does not
// change eigenvectors); the loop runs until the iterate stops moving, so the
// result is the converged eigenvector whatever the spectral gap.
// Replaces the reference's [~,~,V]=svd(M); V(:,end) with S = M'M
// (triangulation3D.m:61-62, linearTFT.m:71-79, R_t_from_TFT.m:47-55).
// Returns the number of iterations used; x has unit norm, sign free.
// --------------------------------------------------------------------------
// *converged (optional): false when the iteration cap was reached before the iterate stopped moving (nearly coincident
// smallest eigenvalues) -- the callers with an exact fall-back (hestenes_min_rsv on the matrix itself) test it.
template <int n>
__device__ __forceinline__ int spd_min_eigvec(const double (&S)[n][n], double (&x)[n], int maxit = 40, bool* converged = nullptr) {
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) tr += S[i][i];
    const double delta = 1e-14 * tr;
    const double pfloor = 1e-3 * delta + 1e-300;
    double L[n][n];
    double inv[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double d = S[j][j] + delta;
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
        const double rs = rsqrt_pos(d);
        L[j][j] = d * rs;
        inv[j] = rs;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double s = S[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * rs;
        }
    }
    // Start vector: the inhomogeneous least-squares solution with the last coordinate fixed to 1,
    //   x = (-S11^-1 s, 1) = (-L11^-T L(n,1:n-1)', 1);
    // the last row of the factor already is the forward solve L11^-1 s, so this costs one (n-1) x (n-1)
    // back substitution.  For a well-posed DLT system it is within ~lambda_n/lambda_(n-1) of the answer,
    // which saves an iteration; if the last coordinate of the true vector is ~0 it is just another start.
    {
        double z[n];
#pragma unroll
        for (int i = n - 2; i >= 0; --i) {
            double sum = -L[n - 1][i];
#pragma unroll
            for (int k = i + 1; k < n - 1; ++k) sum -= L[k][i] * z[k];
            z[i] = sum * inv[i];
        }
        double nn0 = 1.0;
#pragma unroll
        for (int i = 0; i < n - 1; ++i) nn0 += z[i] * z[i];
        const double r0 = rsqrt(nn0);
        const bool fin = nn0 <= 1e300;                                         // false for inf / NaN
        const double xu = (n == 4) ? 0.5 : 0.57735026918962576;
#pragma unroll
        for (int i = 0; i < n - 1; ++i) x[i] = fin ? z[i] * r0 : xu;
        x[n - 1] = fin ? r0 : xu;
    }
    return chol_invit<n>(L, inv, x, maxit, converged);
}

// --------------------------------------------------------------------------
// The same eigenvector for the exact tiers, at SVD accuracy and with an iteration count that does not depend on the gap between
// the two smallest eigenvalues (inconsistent DLT systems -- outliers, wrong pose candidates, minimal samples -- have
// lambda_n / lambda_(n-1) anywhere in (0, 1) and plain inverse iteration then needs up to hundreds of steps):
//   A. spd_min_eigvec with a cap of three iterations: well-posed systems finish here, exactly as in the fast tier;
//   B. shifted inverse iteration, the shift sigma = rho - |S x - rho x| re-chosen every round (rho the Rayleigh quotient): for
//      an iterate a v_n + b v_(n-1), a > b, sigma lies below lambda_n and the round contracts by ~b, i.e. quadratically; every
//      factorisation is a CHOLESKY of S - sigma I, so a shift that overshoots lambda_n is detected (non-positive pivot) and
//      retried four residuals lower -- no indefinite factorisation anywhere;
//   C. certificate: S - (rho + g) I + tr(S) x x' positive definite  =>  (Courant-Fischer on the complement of x) every other
//      eigenvalue exceeds rho + g, so x belongs to the smallest one and sin(angle) <= |S x - rho x| / g <= 1e-3;
//   D. two refinement steps x -= (S - rho I + tr x x')^-1 (M'(M x) - rho x) with the residual evaluated from the matrix M
//      itself (mtm): the fixed point no longer depends on the rounding of S = M'M (eps tr / gap), only the contraction does,
//      which the certified gap g >= 1e-11 tr bounds by 1e-5.  Result: eps sigma_1 / (sigma_(n-1) - sigma_n), what svd(M) gives.
// Returns false (x unspecified) when B does not converge in 12 rounds or C fails -- nearly coincident smallest singular
// values, 7e-4 of the inconsistent systems of a RANSAC scene with the cap at 8 -- for the caller's one-sided Jacobi on M.
// mtm(x, out, rho): out = M'(M x), rho = |M x|^2.
// --------------------------------------------------------------------------
template <int n>
__device__ __forceinline__ bool chol_shifted(const double (&S)[n][n], const double sigma, const double kappa, const double (&x)[n],
                                             const double pfloor, double (&L)[n][n], double (&inv)[n]) {
    bool pd = true;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double d = S[j][j] - sigma + kappa * x[j] * x[j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        pd = pd && ((j == n - 1) ? (d > 0.0) : (d > pfloor));
        d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
        const double rs = rsqrt_pos(d);
        L[j][j] = d * rs;
        inv[j] = rs;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double s = S[i][j] + kappa * x[i] * x[j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * rs;
        }
    }
    return pd;
}
template <int n>
__device__ __forceinline__ void chol_solve(const double (&L)[n][n], const double (&inv)[n], const double (&b)[n], double (&y)[n]) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < n; ++k) s -= L[k][i] * y[k];
        y[i] = s * inv[i];
    }
}
// rho = x'Sx and |S x - rho x| for a unit x (lower triangle of S read)
template <int n>
__device__ __forceinline__ void rayleigh(const double (&S)[n][n], const double (&x)[n], double& rho, double& rn) {
    double Sx[n];
    rho = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) s += ((j <= i) ? S[i][j] : S[j][i]) * x[j];
        Sx[i] = s;
        rho += s * x[i];
    }
    double r2 = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) { const double d = Sx[i] - rho * x[i]; r2 += d * d; }
    rn = sqrt(r2);
}
// steps B - D, from the iterate x that step A left
template <int n, class MtM>
__device__ __forceinline__ bool spd_min_eigvec_cert_tail(const double (&S)[n][n], double (&x)[n], MtM&& mtm) {
    bool conv = false;
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) tr += S[i][i];
    const double pfloor = 1e-30 * tr + 1e-300;
    double L[n][n], inv[n], y[n];
    double mult = 1.0;
#pragma unroll 1
    for (int rnd = 0; rnd < 12; ++rnd) {                                    // B
        double rho, rn;
        rayleigh<n>(S, x, rho, rn);
        if (!(rn > 4e-14 * tr)) { conv = (rn == rn); break; }               // an eigenvector of S to rounding (also leaves on NaN)
        if (!chol_shifted<n>(S, rho - mult * rn - 1e-15 * tr, 0.0, x, pfloor, L, inv)) { mult *= 4.0; continue; }
        mult = 1.0;
        chol_solve<n>(L, inv, x, y);
        double nn = 0.0, dot = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { nn += y[i] * y[i]; dot += y[i] * x[i]; }
        const double rs = rsqrt(nn), sg = (dot < 0.0) ? -rs : rs;
        double r2 = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { const double yi = y[i] * sg; const double d = yi - x[i]; r2 += d * d; x[i] = yi; }
        if (r2 < 1e-12) { conv = true; break; }
    }
    if (!conv) return false;
    {                                                                       // C
        double rho, rn;
        rayleigh<n>(S, x, rho, rn);
        const double g = fmax(1e3 * rn, 1e-11 * tr);
        if (!chol_shifted<n>(S, rho + g, tr, x, pfloor, L, inv)) return false;
    }
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {                                           // D
        double out[n], rho;
        mtm(x, out, rho);
#pragma unroll
        for (int i = 0; i < n; ++i) out[i] -= rho * x[i];
        if (!chol_shifted<n>(S, rho, tr, x, pfloor, L, inv)) return false;
        chol_solve<n>(L, inv, out, y);
        double nn = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) { x[i] -= y[i]; nn += x[i] * x[i]; }
        const double rs = rsqrt(nn);
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] *= rs;
    }
    return true;
}
template <int n, class MtM>
__device__ __forceinline__ bool spd_min_eigvec_cert(const double (&S)[n][n], double (&x)[n], MtM&& mtm) {
    bool conv;
    spd_min_eigvec<n>(S, x, opaque_int(3), &conv);                          // A
    if (conv) return true;
    return spd_min_eigvec_cert_tail<n>(S, x, mtm);
}

// --------------------------------------------------------------------------
// Right singular vector of the smallest singular value of an R_ x C_ matrix M (per lane, registers) by one-sided
// Jacobi (Hestenes) rotations of its columns: the reference's [~,~,V] = svd(M); V(:,end) at SVD accuracy
// (eps sigma_1 / (sigma_(C-1) - sigma_C), no squared conditioning and no dependence of the run time on the gap).
// The gap-independent fall-back of the Cholesky / inverse-iteration path above for nearly coincident smallest
// singular values (inconsistent DLT systems of minimal samples: triangulation3D.m:61-62, linearTFT.m:71-79).
// M is destroyed (its columns end up orthogonal, their norms are the singular values).  x: unit norm, sign free.
// --------------------------------------------------------------------------
template <int R_, int C_>
__device__ inline void hestenes_min_rsv(double (&M)[R_][C_], double (&x)[C_]) {
    double V[C_][C_];
#pragma unroll
    for (int i = 0; i < C_; ++i)
#pragma unroll
        for (int j = 0; j < C_; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < C_ - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < C_; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int r = 0; r < R_; ++r) { al += M[r][p] * M[r][p]; be += M[r][q] * M[r][q]; ga += M[r][p] * M[r][q]; }
                if (fabs(ga) > 1e-15 * sqrt(al * be)) {                   // false for NaN and for a zero column
                    rotated = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int r = 0; r < R_; ++r) {
                        const double mp = M[r][p], mq = M[r][q];
                        M[r][p] = c * mp - s * mq;
                        M[r][q] = s * mp + c * mq;
                    }
#pragma unroll
                    for (int r = 0; r < C_; ++r) {
                        const double vp = V[r][p], vq = V[r][q];
                        V[r][p] = c * vp - s * vq;
                        V[r][q] = s * vp + c * vq;
                    }
                }
            }
        if (!rotated) break;
    }
    double best = 0.0;
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < R_; ++r) nn += M[r][c] * M[r][c];
        const bool take = (c == 0) || nn < best;
        best = take ? nn : best;
#pragma unroll
        for (int r = 0; r < C_; ++r) x[r] = take ? V[r][c] : x[r];
    }
}

// Lower bound on the largest eigenvalue of a symmetric positive semi-definite E x E matrix: Rayleigh quotients of W e_k and
// W^2 e_k (k: the largest diagonal entry), two matrix-vector products.  Within a few per cent of lambda_max when one eigenvalue
// dominates, which is what decides the binade of pinv's tolerance without an eigen-decomposition (gh_wg_kernel.h).
template <int E>
__device__ __forceinline__ double psd_lambda_max_lower(const double (&W)[E][E]) {
    int k = 0;
    double d = W[0][0];
#pragma unroll
    for (int a = 1; a < E; ++a) if (W[a][a] > d) { d = W[a][a]; k = a; }
    double v[E], w1[E], w2[E];
#pragma unroll
    for (int a = 0; a < E; ++a) {
        v[a] = W[a][0];
#pragma unroll
        for (int c = 1; c < E; ++c) v[a] = (k == c) ? W[a][c] : v[a];
    }
    double vv = 0.0, vw = 0.0, ww = 0.0, wz = 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < E; ++c) acc += W[a][c] * v[c];
        w1[a] = acc;
        vv += v[a] * v[a]; vw += v[a] * acc; ww += acc * acc;
    }
#pragma unroll
    for (int a = 0; a < E; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < E; ++c) acc += W[a][c] * w1[c];
        w2[a] = acc;
        wz += w1[a] * acc;
    }
    const double r1 = (vv > 0.0) ? vw / vv : 0.0, r2 = (ww > 0.0) ? wz / ww : 0.0;
    const double r = (r1 > r2) ? r1 : r2;
    return (r > d) ? r : d;
}

// Upper and lower bound on the largest eigenvalue of a symmetric positive semi-definite E x E matrix without an eigen-decomposition:
// trace / Frobenius norm (Wolkowicz-Styan), |W|_F, |W|_F^2 / trace and the power-step Rayleigh quotients above.  MATLAB's
// pinv / rank tolerance max(size) * eps(norm) depends only on the BINADE of the norm, which these bounds usually settle.
template <int E>
__device__ __forceinline__ void psd_lambda_max_bounds(const double (&W)[E][E], double& up, double& lo) {
    double fro2 = 0.0, tr = 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a) {
        tr += W[a][a];
#pragma unroll
        for (int b = 0; b < E; ++b) fro2 += W[a][b] * W[a][b];
    }
    constexpr double n = (double)E;
    const double m = tr / n, s2 = (fro2 - tr * m > 0.0) ? fro2 - tr * m : 0.0;
    up = m + sqrt((n - 1.0) / n * s2);
    lo = m + sqrt(s2 / (n * (n - 1.0)));
    const double fr = sqrt(fro2), ray = (tr > 0.0) ? fro2 / tr : 0.0;
    up = (fr < up) ? fr : up;
    lo = (ray > lo) ? ray : lo;
    const double pw = psd_lambda_max_lower(W);
    lo = (pw > lo) ? pw : lo;
}

// --------------------------------------------------------------------------
// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix (per lane).
// A is overwritten by its diagonal form, V (columns) are the eigenvectors.
// --------------------------------------------------------------------------
__device__ __forceinline__ void jacobi3_rotate(double (&A)[3][3], double (&V)[3][3], const int p, const int q, const int r) {
    const double apq = A[p][q];
    if (apq == 0.0) return;
    const double app = A[p][p], aqq = A[q][q];
    const double tau = (aqq - app) / (2.0 * apq);
    const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double c = rsqrt(1.0 + t * t), s = t * c;
    A[p][p] = app - t * apq;
    A[q][q] = aqq + t * apq;
    A[p][q] = A[q][p] = 0.0;
    const double arp = A[r][p], arq = A[r][q];
    A[r][p] = A[p][r] = c * arp - s * arq;
    A[r][q] = A[q][r] = s * arp + c * arq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}
__device__ __forceinline__ void jacobi3(double (&A)[3][3], double (&V)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 24; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (!(off > 1e-34 * dg)) break;
        jacobi3_rotate(A, V, 0, 1, 2);
        jacobi3_rotate(A, V, 0, 2, 1);
        jacobi3_rotate(A, V, 1, 2, 0);
    }
}

// Full SVD of a 3x3 matrix through the eigen-decomposition of E'E:
// V from Jacobi (columns sorted by descending singular value), u1,u2 = E v / |E v|
// (so that a rotation of (v1,v2) inside a nearly degenerate singular plane is
// mirrored in (u1,u2) -- the product U*W*V' of recover_R_t is then stable),
// u3 = u1 x u2.  Follows [U,~,V]=svd(E21) (R_t_from_TFT.m:85); signs free.
__device__ __forceinline__ void svd3(const Mat3& E, Mat3& U, Mat3& V, double (&sv)[3]) {
    double A[3][3], W[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = E.m[0][i] * E.m[0][j] + E.m[1][i] * E.m[1][j] + E.m[2][i] * E.m[2][j];
    jacobi3(A, W);
    double ev[3] = {A[0][0], A[1][1], A[2][2]};
    // sort indices descending (3 elements)
    int i0 = 0, i1 = 1, i2 = 2;
    if (ev[i0] < ev[i1]) { int t = i0; i0 = i1; i1 = t; }
    if (ev[i1] < ev[i2]) { int t = i1; i1 = i2; i2 = t; }
    if (ev[i0] < ev[i1]) { int t = i0; i0 = i1; i1 = t; }
    const int ord[3] = {i0, i1, i2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // select column ord[c] without dynamic register indexing
#pragma unroll
        for (int k = 0; k < 3; ++k) V.m[k][c] = (ord[c] == 0) ? W[k][0] : ((ord[c] == 1) ? W[k][1] : W[k][2]);
        const double e = (ord[c] == 0) ? ev[0] : ((ord[c] == 1) ? ev[1] : ev[2]);
        sv[c] = sqrt(e > 0.0 ? e : 0.0);
    }
    double u1[3], u2[3], u3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        u1[k] = E.m[k][0] * V.m[0][0] + E.m[k][1] * V.m[1][0] + E.m[k][2] * V.m[2][0];
        u2[k] = E.m[k][0] * V.m[0][1] + E.m[k][1] * V.m[1][1] + E.m[k][2] * V.m[2][1];
    }
    double n1 = rsqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) u1[k] *= n1;
    const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) u2[k] -= d12 * u1[k];
    double n2 = rsqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) u2[k] *= n2;
    cross3(u1, u2, u3);
#pragma unroll
    for (int k = 0; k < 3; ++k) { U.m[k][0] = u1[k]; U.m[k][1] = u2[k]; U.m[k][2] = u3[k]; }
}

// right null vector of a 3x3 matrix M: smallest eigenvector of M'M.  When its inverse iteration hits the cap (nearly
// coincident smallest singular values, rare), EXACT = true finishes with the one-sided Jacobi on M itself, EXACT = false
// only reports it (returns false) so that the caller can hand the triplet to the exact kernel.
// GAP_CHECK (fast tier only): see below -- for the callers whose matrices are well scaled (calibrated tensor slices, unit null vectors).  The
// fundamental-matrix kernels project pixel-coordinate F matrices (singular values 1, 1e-5, 0: lambda_2 ~ 1e-10 tr by construction, and a
// null vector that only enters a correction of size sigma_3) and do not ask for it.
template <bool EXACT = true, bool GAP_CHECK = false>
__device__ __forceinline__ bool null3(const Mat3& M, double (&x)[3]) {
    double S[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S[i][j] = M.m[0][i] * M.m[0][j] + M.m[1][i] * M.m[1][j] + M.m[2][i] * M.m[2][j];
    bool conv;
    if constexpr (EXACT) {
        conv = spd_min_eigvec_cert<3>(S, x, [&](const double (&v)[3], double (&out)[3], double& rho) {
            double mv[3];
            rho = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { mv[i] = M.m[i][0] * v[0] + M.m[i][1] * v[1] + M.m[i][2] * v[2]; rho += mv[i] * mv[i]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) out[j] = M.m[0][j] * mv[0] + M.m[1][j] * mv[1] + M.m[2][j] * mv[2];
        });
    } else {
        spd_min_eigvec<3>(S, x, 40, &conv);
        if constexpr (GAP_CHECK) {
        // The eigenvector of a FORMED M'M carries eps tr / (lambda_2 - lambda_3): fine for generic slices (lambda_2 ~ 1e-2 tr), seven digits
        // short for the nearly rank-one slices of collinear camera centres (R_t_3 4.7e-7 off at N = 200 before this test).  Report a gap
        // under 1e-7 tr -- the limit of the 27 x 27 solve's gram_risk flag -- as "not finished": with s = lambda_1 + lambda_2 = tr - rho and
        // p = lambda_1 lambda_2 = c2 - rho s (c2: the sum of the principal 2 x 2 minors), z = rho + 1e-7 tr lies below lambda_2 iff
        // q(z) = z^2 - s z + p > 0 and z < s / 2.  No square root, ~25 operations.
        const double tr = S[0][0] + S[1][1] + S[2][2];
        double rho = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) rho += x[i] * (S[i][0] * x[0] + S[i][1] * x[1] + S[i][2] * x[2]);
        const double c2 = (S[0][0] * S[1][1] - S[0][1] * S[0][1]) + (S[0][0] * S[2][2] - S[0][2] * S[0][2]) + (S[1][1] * S[2][2] - S[1][2] * S[1][2]);
        const double sm = tr - rho, z = rho + 1e-7 * tr;
        conv = conv && (z * z - sm * z + (c2 - rho * sm) > 0.0) && (z + z < sm);
        }
    }
    if (EXACT && !conv) {
        double A[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) A[i][j] = M.m[i][j];
        hestenes_min_rsv<3, 3>(A, x);
        conv = true;
    }
    return conv;
}

}  // namespace tff
