// Wavefront-cooperative symmetric eigen-solvers on matrices held in LDS
// (n <= 32: one lane per row).  Used for the 27x27 and 15x15 Gram matrices of
// linearTFT (linearTFT.m:64-67 and :84, where the reference runs a full
// svd(A) of the 4N x 27 / 4N x 15 design matrix and keeps V(:,end)) and the
// 9x9 Gram matrix of linearF (linearF.m:54-55).
//
//   wave_min_eigvec : Cholesky of G + delta*I, then inverse iteration until the
//                     iterate stops moving.  Cost ~ n^3/6 + iters * 2 n^2 flops
//                     per wave; typical iters = 4..6 (sigma_27/sigma_26 ~ 0.02).
//   wave_jacobi_min_eigvec : cyclic Jacobi sweeps (all rotations, rows and
//                     columns updated by the 64 lanes through LDS); the
//                     gap-independent fallback when inverse iteration has not
//                     converged, and selectable for every solve with
//                     TFF_SOLVER_JACOBI for cross-checking.
#pragma once
#include "wave.h"

namespace tff {

// G, L: n x n row-major with leading dimension ld in LDS.  G is read only.
// On return lane r (< n) holds component r of the unit eigenvector (0 on the
// other lanes); *iters gets the iteration count, *resid2 the last squared step.
__device__ inline double wave_min_eigvec(const double* G, double* L, const int n, const int ld,
                                         const int maxit, int* iters, double* resid2) {
    const int lane = lane_id();
    const double tr = wave_sum(lane < n ? G[lane * ld + lane] : 0.0);
    const double delta = 1e-14 * tr;
    const double pfloor = 1e-3 * delta + 1e-300;
    const int lr = lane & 7, lc = lane >> 3;
    // L <- lower(G) + delta I
    for (int r = lr; r < n; r += 8)
        for (int c = lc; c <= r; c += 8) L[r * ld + c] = G[r * ld + c] + ((r == c) ? delta : 0.0);
    wave_sync();
    double myinv = 0.0;   // 1/L[lane][lane]
    for (int k = 0; k < n; ++k) {
        double d = L[k * ld + k];
        d = (d > pfloor) ? d : pfloor;
        const double rs = 1.0 / sqrt(d);
        wave_sync();                                        // everyone has read the pivot before lane k overwrites it
        if (lane >= k && lane < n) {
            const double v = (lane == k) ? d * rs : L[lane * ld + k] * rs;
            L[lane * ld + k] = v;
            if (lane == k) myinv = rs;
        }
        wave_sync();
        for (int r = k + 1 + lr; r < n; r += 8) {
            const double lrk = L[r * ld + k];
            for (int c = k + 1 + lc; c <= r; c += 8) L[r * ld + c] -= lrk * L[c * ld + k];
        }
        wave_sync();
    }
    double x = (lane < n) ? 1.0 / sqrt((double)n) : 0.0;
    double rprev2 = 1.0, r2 = 1.0;
    int it = 0;
    while (it < maxit) {
        double y = x;
        for (int j = 0; j < n; ++j) {                       // forward  L y = x
            const double yj = wave_bcast(y * myinv, j);
            if (lane == j) y = yj;
            else if (lane > j && lane < n) y -= L[lane * ld + j] * yj;
        }
        for (int j = n - 1; j >= 0; --j) {                  // backward L' z = y
            const double zj = wave_bcast(y * myinv, j);
            if (lane == j) y = zj;
            else if (lane < j) y -= L[j * ld + lane] * zj;
        }
        const double nn = wave_sum(y * y);
        const double dot = wave_sum(y * x);
        const double rn = 1.0 / sqrt(nn);
        const double yn = y * ((dot < 0.0) ? -rn : rn);
        const double dd = yn - x;
        r2 = wave_sum(dd * dd);
        x = yn;
        ++it;
        if (!(r2 > 1e-26)) break;
        if (it >= 2 && r2 < 0.25 * rprev2 && r2 * r2 < 1e-26 * rprev2) break;
        rprev2 = r2;
    }
    *iters = it;
    *resid2 = r2;
    return x;
}

// Cyclic Jacobi on A (n x n, ld, symmetric, full storage, DESTROYED) with
// eigenvectors accumulated in V (n x n, ld); returns, per lane r < n,
// component r of the eigenvector of the smallest eigenvalue.  Rotation (p,q):
// the lanes update the column pair of A and V, then the row pair of A.
__device__ inline double wave_jacobi_min_eigvec(double* A, double* V, const int n, const int ld, int* sweeps_out) {
    const int lane = lane_id();
    const int lr = lane & 7, lc = lane >> 3;
    for (int r = lr; r < n; r += 8)
        for (int c = lc; c < n; c += 8) V[r * ld + c] = (r == c) ? 1.0 : 0.0;
    const double absfloor = 1e-22 * wave_sum(lane < n ? fabs(A[lane * ld + lane]) : 0.0);
    wave_sync();
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        int rotations = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * ld + q], app = A[p * ld + p], aqq = A[q * ld + q];
                // relative threshold (de Rijk): keeps small eigenvalues accurate
                // plus an absolute floor so rounding noise under a zero eigenvalue is not chased
                if (!(fabs(apq) > 1.1e-16 * sqrt(fabs(app * aqq)) && fabs(apq) > absfloor)) continue;   // wave-uniform
                ++rotations;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                wave_sync();
                if (lane < n) {                               // columns p,q of A and V
                    const double arp = A[lane * ld + p], arq = A[lane * ld + q];
                    A[lane * ld + p] = c * arp - s * arq;
                    A[lane * ld + q] = s * arp + c * arq;
                    const double vrp = V[lane * ld + p], vrq = V[lane * ld + q];
                    V[lane * ld + p] = c * vrp - s * vrq;
                    V[lane * ld + q] = s * vrp + c * vrq;
                }
                wave_sync();
                if (lane < n) {                               // rows p,q of A
                    const double apr = A[p * ld + lane], aqr = A[q * ld + lane];
                    A[p * ld + lane] = c * apr - s * aqr;
                    A[q * ld + lane] = s * apr + c * aqr;
                }
                wave_sync();
            }
        }
        if (rotations == 0) break;
    }
    *sweeps_out = sweep;
    // index of the smallest diagonal entry (wave-uniform scan, n <= 32)
    int best = 0;
    double bv = A[0];
    for (int k = 1; k < n; ++k) { const double d = A[k * ld + k]; if (d < bv) { bv = d; best = k; } }
    double x = (lane < n) ? V[lane * ld + best] : 0.0;
    const double nn = wave_sum(x * x);
    return x * (1.0 / sqrt(nn));
}

}  // namespace tff
