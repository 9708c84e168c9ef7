// The exact tier of the linear solves: the right singular vector of the smallest singular value of a tall design matrix A
// (4N x 27 and 4N x 15 in linearTFT.m:64-67,84; N x 9 in linearF.m:54) WITHOUT forming A'A.
//
// The fast tier (tft_kernel.h, f_kernel.h) takes the eigenvector of the Gram matrix; its rounding error is
// eps |A|^2 / (sigma_(n-1)^2 - sigma_n^2), the square of what the reference's svd(A) has.  Triplets for which that matters
// (minimal samples, nearly coincident smallest singular values -- detected through wave_invit_unit's gap estimate) are
// redone here:
//   1. streaming Householder QR of A, backward stable: R'R = (A + dA)'(A + dA), |dA| ~ eps |A|, any N.  Two layouts:
//      wave_qr_cols_append -- lane c < n owns COLUMN c, M new rows per chunk in its registers, R in LDS; reflector through
//      v_readlane, no cross-lane reduction (the 4N x 27 and 27 x 15 systems of linearTFT, minimal samples of linearF);
//      wave_qr_append -- lane r < n holds ROW r of R in registers, the other 64 - n lanes each take one new row of A per chunk,
//      the reflector products v'A for all columns come from one halving butterfly (N x 9 systems with many rows per chunk).
//   2. inverse iteration with L = R' (wave_invit_unit): the triangular solves perturb R componentwise, so the iterate
//      converges to the singular vector of a matrix within eps |A| of A -- error eps sigma_1 / (sigma_(n-1) - sigma_n), as svd(A).
//   3. if the iteration hits its cap (sigma_n / sigma_(n-1) > ~0.97): one-sided Jacobi (Hestenes) on R in LDS, gap-independent.
#pragma once
#include "wave.h"
#include "wave_eig.h"
#include "row_eig.h"

namespace tff {

// lane that holds value index j after wave_reduce_scatter<32> (inverse of reduce32_index on the even lanes)
__device__ __forceinline__ constexpr int reduce32_lane(int j) {
    return ((j >> 4) & 1) * 32 + ((j >> 3) & 1) * 16 + ((j >> 2) & 1) * 8 + ((j >> 1) & 1) * 4 + (j & 1) * 2;
}

// One chunk of the streaming QR.  On entry lane r < n holds row r of the current R (upper triangular: entries c < r are
// ignored and must be zero or rounding-level), lanes >= n hold new rows of A (zeros where there is none).  On return lanes < n
// hold the updated R; lanes >= n hold rounding-level leftovers.  n <= 32.
template <int n>
__device__ inline void wave_qr_append(double (&g)[n]) {
    static_assert(n <= 32, "one butterfly of 32 values per reflector");
    const int lane = lane_id();
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        double xk = g[0];                                   // g[k], k wave-uniform: select chain (no dynamic register index)
#pragma unroll
        for (int c = 1; c < n; ++c) xk = (c == k) ? g[c] : xk;
        const bool act = lane == k || lane >= n;            // rows k+1..n-1 of R have a zero in column k
        const double x = act ? xk : 0.0;
        const double sigma = wave_sum(x * x);
        const double xkk = wave_bcast(xk, k);
        if (!(sigma - xkk * xkk > 0.0)) continue;           // nothing below the diagonal (wave-uniform; also NaN)
        const double nrm = sqrt(sigma);
        const double alpha = (xkk > 0.0) ? -nrm : nrm;
        const double v = x - ((lane == k) ? alpha : 0.0);   // Householder vector, 0 on the inactive lanes
        const double beta = 1.0 / (sigma - xkk * alpha);    // 2 / v'v
        double acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = (j < n) ? v * g[(j < n) ? j : 0] : 0.0;
        const double tot = wave_reduce_scatter<32>(acc);    // lane reduce32_lane(j): v' A(:,j)
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const double wj = wave_bcast(tot, reduce32_lane(j));
            g[j] -= (beta * wj) * v;                        // columns j < k: w_j is rounding-level, a no-op
        }
    }
}

// The same factorisation with the roles of lanes and registers exchanged: lane j < n owns COLUMN j, a[0..M) are that column's
// entries in M new rows of A, and R (n x n, row-major, zeros below the diagonal) lives in LDS (Rl).  A reflector then needs no
// cross-lane reduction at all -- its vector comes from lane k by v_readlane into scalar registers, v'A(:,j) is a serial dot
// product inside lane j -- 2 M readlanes + 3 M fused multiply-adds per step against ~420 instructions of wave_qr_append
// (select chain, 32 products, 31-add butterfly, n broadcasts).  Only n of 64 lanes work, but the instruction count is what a
// wavefront pays for: 28 x 27 in 5 k instead of 11 k instructions.  On return a[] holds rounding-level leftovers.
template <int n, int M>
__device__ inline void wave_qr_cols_append(double (&a)[M], double* Rl) {
    const int lane = lane_id();
    const int col = (lane < n) ? lane : 0;
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        const double rk = (lane < n) ? Rl[k * n + col] : 0.0;             // row k of R: zero left of the diagonal
        const double rkk = wave_bcast(rk, k);
        double s[M], sigma = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) { s[i] = wave_bcast(a[i], k); sigma = fma(s[i], s[i], sigma); }
        if (!(sigma > 0.0)) continue;                                       // nothing below the diagonal (wave-uniform; also NaN)
        const double nrm = sqrt(fma(rkk, rkk, sigma));
        const double alpha = (rkk > 0.0) ? -nrm : nrm;
        const double v0 = rkk - alpha;                                      // Householder vector (v0, s)
        const double beta = 1.0 / (fma(rkk, rkk, sigma) - rkk * alpha);     // 2 / v'v
        double wj = v0 * rk;
#pragma unroll
        for (int i = 0; i < M; ++i) wj = fma(s[i], a[i], wj);
        const double bw = beta * wj;
#pragma unroll
        for (int i = 0; i < M; ++i) a[i] = fma(-bw, s[i], a[i]);
        if (lane < n && lane >= k) Rl[k * n + col] = (lane == k) ? alpha : fma(-bw, v0, rk);
    }
    wave_sync();
}

// Row r of an R kept in LDS by wave_qr_cols_append -> registers of lane r (the layout wave_qr_to_factor / wave_qr_min_rsv take)
template <int n>
__device__ inline void wave_qr_rows_from_lds(const double* Rl, double (&g)[n]) {
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < n; ++c) g[c] = (lane < n) ? Rl[lane * n + c] : 0.0;
    wave_sync();
}

// R (lane r < n: row r in g) -> the row-scaled factor of wave_invit_unit in Lp (n x n) and myinv = 1 / R[lane][lane];
// Rm (n x n LDS, row-major) receives R itself with exact zeros below the diagonal (for the Hestenes fall-back and R * Up).
// A zero pivot (rank-deficient A, e.g. noise-free data) is floored at 1e-20 |R|_F: inverse iteration then converges in one step.
template <int n>
__device__ inline double wave_qr_to_factor(const double (&g)[n], double* Rm, double* Lp) {
    const int lane = lane_id();
    double fro = 0.0, diag = 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) {
        const double v = (lane < n && c >= lane) ? g[c] : 0.0;
        fro += v * v;
        diag = (c == lane) ? v : diag;
    }
    const double floor_ = 1e-20 * sqrt(wave_sum(fro)) + 1e-300;
    if (fabs(diag) < floor_) diag = (diag < 0.0) ? -floor_ : floor_;
    wave_sync();
    if (lane < n) {
#pragma unroll
        for (int c = 0; c < n; ++c) Rm[lane * n + c] = (c > lane) ? g[c] : ((c == lane) ? diag : 0.0);
    }
    wave_sync();
    const double myinv = (lane < n) ? 1.0 / diag : 0.0;
    if (lane < n) {
#pragma unroll
        for (int c = 0; c < n; ++c) Lp[lane * n + c] = (c < lane) ? Rm[c * n + lane] * myinv : 0.0;   // L = R'
    }
    wave_sync();
    return myinv;
}

// One-sided Jacobi on the columns of the n x n matrix Rm (LDS, row-major, lane r owns row r; DESTROYED), V (n x n LDS)
// accumulates the rotations.  Returns on lane r component r of the right singular vector of the smallest singular value.
__device__ inline double wave_hestenes_min_rsv(double* Rm, double* V, const int n, int* sweeps_out) {
    const int lane = lane_id();
    for (int e = lane; e < n * n; e += WAVE) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    wave_sync();
    int sweep = 0;
#pragma unroll 1
    for (; sweep < 40; ++sweep) {
        int rotations = 0;
#pragma unroll 1
        for (int p = 0; p < n - 1; ++p) {
#pragma unroll 1
            for (int q = p + 1; q < n; ++q) {
                const double rp = (lane < n) ? Rm[lane * n + p] : 0.0, rq = (lane < n) ? Rm[lane * n + q] : 0.0;
                const double al = wave_sum(rp * rp), be = wave_sum(rq * rq), ga = wave_sum(rp * rq);
                if (!(fabs(ga) > 1e-15 * sqrt(al * be))) continue;            // wave-uniform
                ++rotations;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = rsqrt(1.0 + t * t), s = c * t;
                if (lane < n) {
                    Rm[lane * n + p] = c * rp - s * rq;
                    Rm[lane * n + q] = s * rp + c * rq;
                    const double vp = V[lane * n + p], vq = V[lane * n + q];
                    V[lane * n + p] = c * vp - s * vq;
                    V[lane * n + q] = s * vp + c * vq;
                }
            }
        }
        if (rotations == 0) break;
    }
    *sweeps_out = sweep;
    int best = 0;
    double bv = 0.0;
    for (int c = 0; c < n; ++c) {                                              // smallest column norm (wave-uniform scan)
        const double r = (lane < n) ? Rm[lane * n + c] : 0.0;
        const double nn = wave_sum(r * r);
        if (c == 0 || nn < bv) { bv = nn; best = c; }
    }
    wave_sync();
    return (lane < n) ? V[lane * n + best] : 0.0;
}

// Steps 2 and 3 for an R held in registers (lane r < n: row r).  Rm: n x n LDS, receives R; Lp: n x n LDS, the factor; Vm: n x n LDS
// for the rotations of the fall-back -- it MAY BE Lp itself (the factor is dead when the fall-back starts).  The fall-back works
// in place: *iters >= 1000 (1000 + sweeps) tells the caller that Rm no longer holds R.
template <int n>
__device__ inline double wave_qr_min_rsv(const double (&g)[n], double* Rm, double* Vm, double* Lp, const int maxit, int* iters) {
    const double myinv = wave_qr_to_factor<n>(g, Rm, Lp);
    int it = 0;
    double r2 = 0.0;
    double x = row_invit_unit<n>(Lp, myinv, maxit, &it, &r2);                 // (row_eig.h: the DPP form of wave_invit_unit)
    if (!eig_converged(r2)) {
        wave_sync();
        int sw = 0;
        x = wave_hestenes_min_rsv(Rm, Vm, n, &sw);
        it = 1000 + sw;
    }
    *iters = it;
    return x;
}

}  // namespace tff
