// Smallest eigenvector of a symmetric positive semi-definite n x n matrix (n <= 32) by Cholesky + inverse iteration, with every
// cross-lane operand taken through the DP-ALU DPP path: v_fmac_f64 ... row_newbcast:J reads lane J of the caller's ROW OF 16 as one of its
// factors, so one step of a factorisation or substitution is ONE VALU instruction where a v_readlane broadcast (2 x v_readlane_b32 +
// v_fma_f64) takes three.  The price is the layout, since the broadcast does not leave a row of 16 lanes:
//   * position p = lane & 15 owns matrix rows p ("lo", g0 / y0) and 16 + p ("hi", g1 / y1; zero rows when 16 + p >= n);
//   * all four rows of 16 lanes of the wavefront hold the SAME data and execute the same arithmetic (bit-identical replicas) -- the lanes
//     were idle anyway (27 / 15 / 9 of 64 in the broadcast form), and no lane ever needs a value from another row of 16.
// Same algorithm, same order of the floating-point operations per matrix entry and the same outputs as wave_min_eigvec_reg / wave_invit_unit
// (wave_eig.h; linearTFT.m:64-67, :84 and linearF.m:54-55 keep V(:,end) of an svd): 27 x 27: 471 + 77 per iteration instead of 1053 + 162
// cross-lane VALU instructions; 15 x 15 and 9 x 9 (one half only): a third of the broadcast form.
// Wait states: a VGPR written by a VALU instruction may be read through DPP two issue slots later at the earliest; the primitives insert
// them (wave_target.h), and the asm statements are volatile, i.e. stay in program order, which is what the WAIT = 0 uses below rely on.
#pragma once
#include "wave.h"
#include "wave_eig.h"

namespace tff {

template <int n> struct RowEigDims {
    static_assert(n >= 2 && n <= 32, "two matrix rows per position of a row of 16 lanes");
    static constexpr int N0 = (n < 16) ? n : 16;          // columns a lo row needs (c <= p <= 15)
    static constexpr int N1 = (n > 16) ? n : 1;           // columns a hi row needs (dummy when there is no hi half)
    static constexpr bool HI = n > 16;
};

// right-looking Cholesky, the updates of pivot column K: entry C of every row r >= C loses L[r][K] L[C][K]; L[C][K] sits in row C's
// (position C & 15, half C >> 4) entry K
template <int n, int K, int C>
struct RowCholUpdate {
    template <int N0, int N1>
    static __device__ __forceinline__ void run(double (&g0)[N0], double (&g1)[N1]) {
        if constexpr (C < n) {
            constexpr int Q = C & 15, W = (C == K + 1) ? 2 : 0;               // the column was scaled just before its first use
            if constexpr (C < 16) {
                if constexpr (RowEigDims<n>::HI) {
                    g1[C] = fnmac_row_bcast<Q, W>(g1[C], g0[K], g1[K]);
                    g0[C] = fnmac_row_bcast<Q, 0>(g0[C], g0[K], g0[K]);
                } else {
                    g0[C] = fnmac_row_bcast<Q, W>(g0[C], g0[K], g0[K]);
                }
            } else {
                g1[C] = fnmac_row_bcast<Q, W>(g1[C], g1[K], g1[K]);
            }
            RowCholUpdate<n, K, C + 1>::run(g0, g1);
        }
    }
};
template <int n, int K>
struct RowChol {
    template <int N0, int N1>
    static __device__ __forceinline__ void run(double (&g0)[N0], double (&g1)[N1], double& myinv0, double& myinv1, const double delta, const double pfloor, const int p) {
        if constexpr (K < n) {
            constexpr int Q = K & 15;
            double d;
            if constexpr (K < 16) d = row_bcast<Q>(g0[K]); else d = row_bcast<Q>(g1[K]);      // pivot: row K's diagonal, fully updated
            d += delta;                                                      // the shift of G + delta I enters here: the diagonal is only ever read as a pivot
            d = fmax(d, pfloor);                                               // (NaN -> pfloor, as the select did)
            const double rs = rsqrt_pos(d);
            // column K of L: rows > K are meaningful.  Row K's own entry (the diagonal of L) is never read again -- 1 / L[K][K] = rs is what the
            // substitutions use, and the factor's diagonal is masked out below -- so the whole column is scaled without a select
            if constexpr (K < 16) {
                g0[K] *= rs;
                if constexpr (RowEigDims<n>::HI) g1[K] *= rs;
                myinv0 = (p == Q) ? rs : myinv0;
            } else {
                g1[K] *= rs;
                myinv1 = (p == Q) ? rs : myinv1;
            }
            RowCholUpdate<n, K, K + 1>::run(g0, g1);
            RowChol<n, K + 1>::run(g0, g1, myinv0, myinv1, delta, pfloor, p);
        }
    }
};
// forward substitution L' y = D^-1 x with the row-scaled unit factor L' (g0 / g1: the position's own rows, zero on and above the diagonal)
template <int n, int J>
struct RowForward {
    template <int N0, int N1>
    static __device__ __forceinline__ void run(double& y0, double& y1, const double (&g0)[N0], const double (&g1)[N1]) {
        if constexpr (J < n - 1) {
            if constexpr (J < 16) {
                if constexpr (RowEigDims<n>::HI) {
                    y1 = fnmac_row_bcast<J, 2>(y1, y0, g1[J]);
                    if constexpr (J < 15) y0 = fnmac_row_bcast<J, 0>(y0, y0, g0[J]);
                } else {
                    y0 = fnmac_row_bcast<J, 2>(y0, y0, g0[J]);
                }
            } else {
                y1 = fnmac_row_bcast<J - 16, 2>(y1, y1, g1[J]);
            }
            RowForward<n, J + 1>::run(y0, y1, g0, g1);
        }
    }
};
// backward substitution L'^T u = y (c0 / c1: the position's own COLUMNS of L', c0[j] = L'[j][p], c1[j] = L'[j][16 + p]; they come from the LDS copy
// of the factor, read by the caller BEFORE the chain starts: the DPP statements are volatile and nothing moves across them)
template <int n, int J>
struct RowBackward {
    static __device__ __forceinline__ void run(double& y0, double& y1, const double (&c0)[n], const double (&c1)[n]) {
        if constexpr (J >= 1) {
            if constexpr (J >= 16) {
                y0 = fnmac_row_bcast<J - 16, 2>(y0, y1, c0[J]);
                if constexpr (J > 16) y1 = fnmac_row_bcast<J - 16, 0>(y1, y1, c1[J]);
            } else {
                y0 = fnmac_row_bcast<J, 2>(y0, y0, c0[J]);
            }
            RowBackward<n, J - 1>::run(y0, y1, c0, c1);
        }
    }
};

// Inverse iteration with the row-scaled unit factor L' (wave_invit_unit's algorithm and tests) in the row-of-16 layout.  Lp: the factor in LDS
// (n x n, row-major, zeros on and above the diagonal); g0 / g1: the position's rows of it (zero rows on positions without a matrix row),
// myinv0 / myinv1 = 1 / L[r][r] of those rows (0 without one).  x0 / x1: the start on entry, the unit eigenvector on return.
// risk[0] / risk[1] (optional): wave_invit_unit's gap_risk numerator and denominator.
template <int n>
__device__ __forceinline__ void row_invit_core(double (&g0)[RowEigDims<n>::N0], const double (&g1)[RowEigDims<n>::N1], const double* Lp,
                                               const double myinv0, const double myinv1, double& x0, double& x1, const int maxit,
                                               int* iters, double* resid2, double* risk) {
    constexpr int N0 = RowEigDims<n>::N0;
    constexpr bool HI = RowEigDims<n>::HI;
    const int p = lane_id() & 15;
    const bool valid0 = p < n, valid1 = HI && 16 + p < n;
    const int q0 = valid0 ? p : n - 1, q1 = valid1 ? 16 + p : n - 1;         // column n - 1 of a strictly lower triangular matrix: zeros
    double c0[n], c1[n];                                                     // (entries the substitution never touches are never loaded)
#pragma unroll
    for (int j = 0; j < n; ++j) { c0[j] = Lp[j * n + q0]; c1[j] = HI ? Lp[j * n + q1] : 0.0; }
    double rprev2 = 1.0, res = 1.0, rk_r2 = 0.0, rk_rp = 1.0, rk_nn = 0.0;
    int it = 0;
    bool done = false;
#pragma unroll 1
    while (true) {
        double y0 = x0 * myinv0, y1 = x1 * myinv1;
        if constexpr (HI) {                                 // the lo rows come back from the LDS copy every iteration: 2 N0 fewer registers live across the
            const double* rows = Lp + opaque_int(0);        // backward substitution, where the register demand peaks (rows + columns of both halves)
#pragma unroll
            for (int c = 0; c < N0; ++c) g0[c] = rows[q0 * n + c];         // (n > 16: every position has a lo row)
        }
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
        if (!done) {                                        // the same tests as wave_invit_unit
            x0 = yn0; x1 = yn1;
            ++it;
            if (r2 <= 1e-26) { res = 0.0; done = true; }
            else if (it >= 2 && r2 < 0.25 * rprev2 && r2 * r2 < 1e-26 * rprev2) { res = 0.0; done = true; }
            else if (!(r2 == r2) || it >= maxit) { res = (r2 == r2) ? r2 : 1.0; done = true; }
            if (it >= 2 && r2 > 1e-30) { rk_r2 = r2; rk_rp = rprev2; rk_nn = nn; }
            rprev2 = r2;
        }
        if (wave_uniform_i(done ? 1 : 0)) break;            // every lane holds the same r2 (replicated rows of 16): no vote needed
    }
    *iters = it;
    *resid2 = res;
    if (risk) {
        risk[0] = 4.0 * rk_nn * rk_r2 * rk_rp;
        const double d = rk_rp - rk_r2;
        risk[1] = (rk_r2 < rk_rp) ? d * d : 0.0;
    }
}

// g0[c] = G[p][c] (c <= p), g1[c] = G[16 + p][c] (c <= 16 + p; zeros when 16 + p >= n), d0 / d1 the rows' diagonal entries (0 for a row
// that does not exist), p = lane & 15, the same in all four rows of 16 lanes.  start0 / start1 (has_start): the initial guess, components p
// and 16 + p.  Lp: n * n doubles of LDS.  Outputs as wave_min_eigvec_reg: lane r < n returns component r of the unit eigenvector (0 on the
// other lanes); *iters, *resid2 (0: converged), *gram_risk (optional).
template <int n>
__device__ inline double row_min_eigvec(double (&g0)[RowEigDims<n>::N0], double (&g1)[RowEigDims<n>::N1], const double d0, const double d1,
                                        double* Lp, const int maxit, int* iters, double* resid2, const bool has_start = false,
                                        const double start0 = 0.0, const double start1 = 0.0, double* gram_risk = nullptr,
                                        const double gram_risk_limit2 = 1e14) {
    constexpr int N0 = RowEigDims<n>::N0;
    constexpr bool HI = RowEigDims<n>::HI;
    const int lane = lane_id(), p = lane & 15;
    const bool valid0 = p < n, valid1 = HI && 16 + p < n;
    const double tr = row_sum16((valid0 ? d0 : 0.0) + (valid1 ? d1 : 0.0));
    const double delta = 1e-14 * tr;
    const double pfloor = 1e-3 * delta + 1e-300;
    double myinv0 = 0.0, myinv1 = 0.0;                                       // 1 / L[r][r] of the position's rows
    RowChol<n, 0>::run(g0, g1, myinv0, myinv1, delta, pfloor, p);
    // row-scaled unit factor L' = D^-1 L in place (zeros on and above the diagonal); its transpose goes through LDS once
#pragma unroll
    for (int c = 0; c < N0; ++c) g0[c] = (c < p && valid0) ? g0[c] * myinv0 : 0.0;
    if constexpr (HI) {
#pragma unroll
        for (int c = 0; c < n; ++c) g1[c] = (c < 16 + p && valid1) ? g1[c] * myinv1 : 0.0;
    }
    wave_sync();
    if (lane < 16) {
        if (valid0) {
#pragma unroll
            for (int c = 0; c < n; ++c) Lp[p * n + c] = (c < N0) ? g0[c < N0 ? c : 0] : 0.0;
        }
        if constexpr (HI) {
            if (valid1) {
#pragma unroll
                for (int c = 0; c < n; ++c) Lp[(16 + p) * n + c] = g1[c];
            }
        }
    }
    wave_sync();
    double x0 = valid0 ? rsqrt((double)n) : 0.0, x1 = valid1 ? rsqrt((double)n) : 0.0;
    if (has_start) {                                        // a zero / non-finite guess falls back to the uniform vector
        const double s0 = valid0 ? start0 : 0.0, s1 = valid1 ? start1 : 0.0;
        const double nn0 = row_sum16(s0 * s0 + s1 * s1);
        if (nn0 > 1e-300 && nn0 < 1e300) { const double r0 = rsqrt(nn0); x0 = s0 * r0; x1 = s1 * r0; }
    }
    double risk[2] = {0.0, 1.0};
    row_invit_core<n>(g0, g1, Lp, myinv0, myinv1, x0, x1, maxit, iters, resid2, gram_risk ? risk : nullptr);
    if (gram_risk) *gram_risk = (tr * tr * risk[0] < gram_risk_limit2 * risk[1]) ? 0.0 : 1.0;
    return (lane < 16) ? x0 : ((lane < 32 && lane < n) ? x1 : 0.0);
}

// The same iteration for a factor that already sits in LDS in wave_invit_unit's form (wave_qr.h: L = R' of a QR factorisation): Lp (n x n,
// row-major, row-scaled, zeros on and above the diagonal), myinv = 1 / L[lane][lane] on lane `lane` < n.  Uniform start.  Lane r < n returns
// component r of the unit eigenvector.
template <int n>
__device__ inline double row_invit_unit(const double* Lp, const double myinv, const int maxit, int* iters, double* resid2) {
    constexpr int N0 = RowEigDims<n>::N0, N1 = RowEigDims<n>::N1;
    constexpr bool HI = RowEigDims<n>::HI;
    const int lane = lane_id(), p = lane & 15;
    const bool valid0 = p < n, valid1 = HI && 16 + p < n;
    const double m0 = wave_shfl(myinv, valid0 ? p : 0), m1 = wave_shfl(myinv, valid1 ? 16 + p : 0);
    const double myinv0 = valid0 ? m0 : 0.0, myinv1 = valid1 ? m1 : 0.0;
    double g0[N0], g1[N1];
#pragma unroll
    for (int c = 0; c < N0; ++c) { const double v = Lp[(valid0 ? p : 0) * n + c]; g0[c] = valid0 ? v : 0.0; }
    if constexpr (HI) {
#pragma unroll
        for (int c = 0; c < n; ++c) { const double v = Lp[(valid1 ? 16 + p : 0) * n + c]; g1[c] = valid1 ? v : 0.0; }
    } else {
        g1[0] = 0.0;
    }
    double x0 = valid0 ? rsqrt((double)n) : 0.0, x1 = valid1 ? rsqrt((double)n) : 0.0;
    row_invit_core<n>(g0, g1, Lp, myinv0, myinv1, x0, x1, maxit, iters, resid2, nullptr);
    return (lane < 16) ? x0 : ((lane < 32 && lane < n) ? x1 : 0.0);
}

}  // namespace tff
