// The exact tier of the linear solves (wave_qr.h: streaming Householder QR of the explicit design matrix, then inverse iteration with L = R')
// in the ROW layout of the four-triplets-per-wavefront kernels: every row of 16 lanes factors its OWN system.
//
//   * position p owns COLUMNS p and 16 + p of the system (n <= 32); a chunk of M new rows lives in its registers (a0 / a1), R itself in the
//     row's LDS workspace in packed upper-triangular form (entry (r, c >= r) at r n - r (r - 1) / 2 + c - r: n (n + 1) / 2 doubles);
//   * a Householder step needs the owner's M new entries on every lane of the row: the owner publishes them through M doubles of LDS (a broadcast
//     read per entry, no VALU slot; the step index is a loop variable, so a DPP row_newbcast, whose lane is an immediate, would need the 27 steps
//     unrolled -- 50 KB of code); v'A(:,c) is then a serial dot product inside the lane that owns column c, as in wave_qr_cols_append:
//     ~200 VALU instructions per step and FOUR systems per wavefront, against 185 for one;
//   * the inverse iteration reads rows and columns of the row-scaled factor L' = D^-1 R' straight from the packed R (and 1 / diag(R) from n
//     doubles of LDS): no second copy of the factor.
// Same arithmetic per entry as wave_qr_cols_append / wave_qr_to_factor / row_invit_core; what a row cannot finish here (iteration cap: nearly
// coincident smallest singular values) is reported, and the caller hands that triplet to the one-triplet exact kernel, whose one-sided Jacobi
// on R does not depend on the gap.
// Reference: linearTFT.m:64-67,84 ([~,~,V] = svd(A); V(:,end)).
#pragma once
#include "wave.h"
#include "row_eig.h"

namespace tff {

__device__ __forceinline__ int rows_up_index(const int n, const int r, const int c) { return r * n - (r * (r - 1)) / 2 + (c - r); }   // c >= r
template <int n> constexpr int rows_up_doubles() { return n * (n + 1) / 2; }

// Rp <- 0
template <int n>
__device__ __forceinline__ void rows_qr_clear(double* Rp) {
    const int p = lane_id() & 15;
    for (int e = p; e < rows_up_doubles<n>(); e += 16) Rp[e] = 0.0;
    wave_sync();
}

// One chunk of the streaming QR: M new rows, column p in a0 and column 16 + p in a1 (zeros where the position owns no column / the chunk
// has fewer rows).  Rp: the row's packed R; xch: M doubles of the row's LDS.  On return a0 / a1 hold rounding-level leftovers.
template <int n, int M>
__device__ __forceinline__ void rows_qr_append(double (&a0)[M], double (&a1)[M], double* Rp, double* xch) {
    constexpr bool HI = n > 16;
    const int p = lane_id() & 15;
    const bool own0 = p < n, own1 = HI && 16 + p < n;
#pragma unroll 1
    for (int k = 0; k < n; ++k) {
        wave_sync();                                                         // (the previous step's readers are done with xch)
        if (k < 16) {
            if (p == k) {
#pragma unroll
                for (int i = 0; i < M; ++i) xch[i] = a0[i];
            }
        } else if (HI) {
            if (p == k - 16) {
#pragma unroll
                for (int i = 0; i < M; ++i) xch[i] = a1[i];
            }
        }
        wave_sync();
        // (the owner's entries are re-read from LDS in each of the three loops below: broadcast reads cost no VALU slot, M more live doubles would)
        const double* s = xch + opaque_int(0);
        double sigma = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) sigma = fma(s[i], s[i], sigma);
        const int kk = rows_up_index(n, k, k);
        const double rkk = Rp[kk];
        const bool in0 = own0 && p >= k, in1 = own1 && 16 + p >= k;          // the columns of row k of R this position holds
        const double rk0 = in0 ? Rp[kk + (in0 ? p - k : 0)] : 0.0;
        const double rk1 = in1 ? Rp[kk + (in1 ? 16 + p - k : 0)] : 0.0;
        const bool live = sigma > 0.0;                                       // (per row; false also for NaN: nothing below the diagonal, the step is skipped)
        const double nrm = sqrt(fma(rkk, rkk, sigma));
        const double alpha = (rkk > 0.0) ? -nrm : nrm;
        const double v0 = rkk - alpha;                                       // Householder vector (v0, s)
        const double beta = live ? 1.0 / (fma(rkk, rkk, sigma) - rkk * alpha) : 0.0;   // 2 / v'v
        double w0 = v0 * rk0, w1 = v0 * rk1;
#pragma unroll
        for (int i = 0; i < M; ++i) { w0 = fma(s[i], a0[i], w0); if (HI) w1 = fma(s[i], a1[i], w1); }
        const double bw0 = beta * w0, bw1 = beta * w1;
#pragma unroll
        for (int i = 0; i < M; ++i) { a0[i] = fma(-bw0, s[i], a0[i]); if (HI) a1[i] = fma(-bw1, s[i], a1[i]); }
        wave_sync();                                                         // (every lane has read row k of R, its diagonal included)
        if (live) {
            if (in0) Rp[kk + p - k] = (p == k) ? alpha : fma(-bw0, v0, rk0);
            if (in1) Rp[kk + 16 + p - k] = (16 + p == k) ? alpha : fma(-bw1, v0, rk1);
        }
    }
    wave_sync();
}

// Right singular vector of the smallest singular value of the system whose R sits in Rp: inverse iteration with L = R' (row_invit_core's loop
// and stopping tests).  dinv: n doubles of the row's LDS.  Position p returns components p (x0) and 16 + p (x1); *resid2 == 0 when converged.
// A zero pivot (rank-deficient system, e.g. noise-free data) is floored at 1e-20 |R|_F as in wave_qr_to_factor.
template <int n>
__device__ __forceinline__ void rows_invit_from_R(const double* Rp, double* dinv, const int maxit, int* iters, double* resid2, double& x0, double& x1) {
    constexpr int N0 = RowEigDims<n>::N0, N1 = RowEigDims<n>::N1;
    constexpr bool HI = RowEigDims<n>::HI;
    const int p = lane_id() & 15;
    const bool valid0 = p < n, valid1 = HI && 16 + p < n;
    const int q0 = valid0 ? p : 0, q1 = valid1 ? 16 + p : 0;
    double fro = 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) {                                            // columns q0 / q1 of R (rows c <= q)
        const double v0 = (valid0 && c <= q0) ? Rp[rows_up_index(n, (c <= q0) ? c : 0, q0)] : 0.0;
        const double v1 = (valid1 && c <= q1) ? Rp[rows_up_index(n, (c <= q1) ? c : 0, q1)] : 0.0;
        fro += v0 * v0 + v1 * v1;
    }
    const double floor_ = 1e-20 * sqrt(row_sum16(fro)) + 1e-300;
    double d0 = valid0 ? Rp[rows_up_index(n, q0, q0)] : 1.0, d1 = valid1 ? Rp[rows_up_index(n, q1, q1)] : 1.0;
    if (fabs(d0) < floor_) d0 = (d0 < 0.0) ? -floor_ : floor_;
    if (fabs(d1) < floor_) d1 = (d1 < 0.0) ? -floor_ : floor_;
    const double myinv0 = valid0 ? 1.0 / d0 : 0.0, myinv1 = valid1 ? 1.0 / d1 : 0.0;
    wave_sync();
    if (valid0) dinv[q0] = myinv0;
    if (valid1) dinv[q1] = myinv1;
    wave_sync();
    // rows q0 / q1 of L' = D^-1 R' (g0[c] = R[c][q0] / R[q0][q0], c < q0) and its columns q0 / q1 (c0[j] = L'[j][q0] = R[q0][j] / R[j][j], j > q0)
    double g0[N0], g1[N1], c0[n], c1[n];
#pragma unroll
    for (int c = 0; c < N0; ++c) g0[c] = (valid0 && c < q0) ? Rp[rows_up_index(n, (c < q0) ? c : 0, q0)] * myinv0 : 0.0;
    if constexpr (HI) {
#pragma unroll
        for (int c = 0; c < n; ++c) g1[c] = (valid1 && c < q1) ? Rp[rows_up_index(n, (c < q1) ? c : 0, q1)] * myinv1 : 0.0;
    } else {
        g1[0] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < n; ++j) {
        c0[j] = (valid0 && j > q0) ? Rp[rows_up_index(n, q0, (j > q0) ? j : q0)] * dinv[j] : 0.0;
        c1[j] = (valid1 && j > q1) ? Rp[rows_up_index(n, q1, (j > q1) ? j : q1)] * dinv[j] : 0.0;
    }
    x0 = valid0 ? rsqrt((double)n) : 0.0;
    x1 = valid1 ? rsqrt((double)n) : 0.0;
    double rprev2 = 1.0, res = 1.0;
    int it = 0;
    bool done = false;
    // Extrapolation of the slow mode (round 5).  Minimal samples of outlier-ridden scenes often have TWO small singular values close together
    // (sigma_n / sigma_(n-1) ~ 0.9 .. 0.99): the iteration then crawls along one direction, step_(k+1) ~ q step_k with q = (sigma_n / sigma_(n-1))^2
    // -- 100 to 250 iterations, and the three other rows of the wavefront wait.  Once consecutive steps are aligned (cos^2 >= 0.97) and shrink
    // slowly (q > 0.15; measured on config 4: q > 0.5 gives 22.8 ms per million samples, q > 0.15 21.4, q > 0.08 with cos^2 >= 0.95 21.3 and the first
    // hand-overs), the geometric tail is summed in one go: x += step q / (1 - q), q = <step_(k+1), step_k> / |step_k|^2.  The iterate is still
    // only ever ACCEPTED by the tests below (it stopped moving under the plain iteration), so what converges is the same fixed point.
    double pd0 = 0.0, pd1 = 0.0;                            // the previous step
    int since = 0;                                          // iterations since the start / the last extrapolation
#pragma unroll 1
    while (true) {
        double y0 = x0 * myinv0, y1 = x1 * myinv1;
        RowForward<n, 0>::run(y0, y1, g0, g1);
        RowBackward<n, n - 1>::run(y0, y1, c0, c1);
        y0 *= myinv0; y1 *= myinv1;
        const double nn = row_sum16(y0 * y0 + y1 * y1);
        const double dot = row_sum16(y0 * x0 + y1 * x1);
        const double rn = rsqrt(nn);
        const double sc = (dot < 0.0) ? -rn : rn;
        const double yn0 = y0 * sc, yn1 = y1 * sc;
        const double dd0 = yn0 - x0, dd1 = yn1 - x1;
        const double r2 = row_sum16(dd0 * dd0 + dd1 * dd1);
        const double cross = row_sum16(dd0 * pd0 + dd1 * pd1);
        const bool live = !done;                            // (a row that has stopped keeps its iterate and its counters)
        bool jumped = false;
        double q = 0.0;
        if (live) {                                         // the same tests as wave_invit_unit
            x0 = yn0; x1 = yn1;
            ++it; ++since;
            if (r2 <= 1e-26) { res = 0.0; done = true; }
            else if (it >= 2 && r2 < 0.25 * rprev2 && r2 * r2 < 1e-26 * rprev2) { res = 0.0; done = true; }
            else if (!(r2 == r2) || it >= maxit) { res = (r2 == r2) ? r2 : 1.0; done = true; }
            jumped = !done && since >= 2 && r2 > 1e-24 && r2 > 0.0225 * rprev2 && cross > 0.0 && cross * cross >= 0.97 * r2 * rprev2;
            q = jumped ? cross / rprev2 : 0.0;
            jumped = jumped && q > 0.15 && q < 0.9995;
        }
        if (wave_any(jumped)) {                             // (outside every per-row branch: the ballot and the row reduction are the whole wavefront's)
            const double f = jumped ? q / (1.0 - q) : 0.0;
            const double e0 = x0 + f * dd0, e1 = x1 + f * dd1;
            const double nr = rsqrt(row_sum16(e0 * e0 + e1 * e1));
            x0 = jumped ? e0 * nr : x0; x1 = jumped ? e1 * nr : x1;
        }
        if (live) {
            pd0 = jumped ? 0.0 : dd0; pd1 = jumped ? 0.0 : dd1;
            since = jumped ? 0 : since;
            rprev2 = jumped ? 0.0 : r2;                     // (0: the step after a jump says nothing about the rate -- the predictive test sits out one iteration)
        }
        if (!wave_any(!done)) break;                        // the four rows iterate on four different systems
    }
    *iters = it;
    *resid2 = res;
}

}  // namespace tff
