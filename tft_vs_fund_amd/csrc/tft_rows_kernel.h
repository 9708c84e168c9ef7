// LinearTFTPoseEstimation, FOUR triplets per wavefront: every row of 16 lanes owns one triplet from the first load to the last store.
//
// k_linear_tft_pose<false> (tft_kernel.h) gives a triplet the whole wavefront.  Its per-correspondence passes use all 64 lanes, but more than
// half of its cycles go to stages that cannot: the 27 x 27 / 15 x 15 eigen-solves (row_eig.h: the DP-ALU DPP broadcast never leaves a row of
// 16 lanes, so the four rows computed four bit-identical replicas), epipoles and frames (6 / 2 lanes), the tensor transforms (27),
// the 3 x 3 SVDs and candidate cameras (2 / 4).  Here the four rows hold four DIFFERENT triplets:
//   * the lane-sparse middle costs the same instructions per wavefront as before and serves four triplets;
//   * the per-correspondence passes stride over a triplet's correspondences 16 at a time, all four rows in the same instruction stream:
//     ceil(N / 16) trips per four triplets instead of 4 ceil(N / 64) (N = 200: 13 against 16, and no 8-of-64-lanes tail trip);
//   * every reduction is a row reduction (four DPP steps), every broadcast a row_newbcast; nothing crosses a row, nothing is wave-uniform
//     per triplet any more: cameras and normalisations live in vector registers / LDS instead of scalar registers;
//   * LDS per triplet is 4.4 KB (the Cholesky factor packed, the R_t_from_TFT workspace overlaid on it), 17.6 KB per wavefront: two
//     wavefronts per SIMD as before.  The correspondences are not staged (4 x 48 N bytes would not fit): they are re-read through L2 / MALL.
// Same arithmetic per matrix entry / correspondence as the one-triplet kernel (same functions wherever a stage is per lane); sums over
// correspondences are taken in row order, so results agree to rounding, not bit for bit.
// A triplet a fast tier cannot finish or certify is marked ST_RETRY for k_linear_tft_pose<true>, as before.
//
// Reference: TFT_methods/LinearTFTPoseEstimation.m:44-62, linearTFT.m:33-91, transform_TFT.m:42-49, R_t_from_TFT.m:40-106,
// auxiliar_functions/Normalize2Ddata.m:33-39, triangulation3D.m:51-63.
#pragma once
#include "tft_kernel.h"
#include "tft_moments_kernel.h"

namespace tff {

constexpr int ROW_TRIPLETS = 4;          // triplets per wavefront
constexpr int ROWL = 16;                 // lanes per triplet

// R_t_from_TFT workspace of one triplet; lives in RowLds::ov once the eigen-solves are done with it
struct RowRt {
    double T1[28];         // tensor after de-normalisation (output T)
    double T2[28];         // calibrated tensor
    double nullv[18];
    double mats[28];       // M1, inv(M2), inv(M3) of transform_TFT
    double Ein[18];        // E21, E31 (row-major)
    double cand[2][21];    // per call: R (9, row-major), Rp (9), t (3)
    double P[4][12];       // candidate cameras K_v [R_c | t]
    double candRt[4][12];  // candidate poses
    double Rt[2][12];      // chosen poses, row-major 3x4
    double Pfin[3][12];    // final cameras
};
constexpr int ROW_OV_DOUBLES = 380;      // >= 27 * 26 / 2 + 1 (packed factor + the zero slot), >= 27 * 28 / 2 (R of the exact tier, rows_qr.h), >= sizeof(RowRt)
static_assert(sizeof(RowRt) <= ROW_OV_DOUBLES * sizeof(double), "overlay");
struct RowLds {
    double mom[96];        // moment sums (tft_kernel.h)
    double nrm[9];
    double calm[27];
    double t[27];          // linearTFT's tensor (normalised frame)
    double epi[6];
    double Q[18];
    double tp[15];
    double pa[18];         // linearTFT's a (-> P2, P3 of the constrained solution), for the iterative methods' linear stage
    double ov[ROW_OV_DOUBLES];   // overlay: packed Cholesky factor | slice null vectors | Gp | RowRt
};
constexpr int ROW_LDS_DOUBLES = (int)(sizeof(RowLds) / sizeof(double));
static_assert(ROW_LDS_DOUBLES % 2 == 0, "16-byte row stride");
inline size_t rows_lds_bytes() { return (size_t)ROW_TRIPLETS * sizeof(RowLds); }
inline unsigned rows_grid(long B) { const long g = (B + ROW_TRIPLETS - 1) / ROW_TRIPLETS; return (unsigned)(g > 0 ? g : 1); }

__device__ __forceinline__ int rows_p() { return lane_id() & 15; }

// where a row's correspondences come from: its own 6 x N block, or (config 4) indices into one shared scene
struct RowSrc { const double* pts; const int* idx; int ns; bool sampled; };
__device__ __forceinline__ Pt6 rows_load(const RowSrc& s, const int i) {
    if (s.sampled) {                                                         // wave-uniform (a kernel argument)
        int k = s.idx[i];
        k = (k >= 0 && k < s.ns) ? k : 0;                                    // (a row with such an index reports ST_TOO_FEW)
        return load_pt(s.pts, k);
    }
    return load_pt(s.pts, i);
}

__device__ __forceinline__ void rows_stamp(double* dbg, const int slot) {
    if (dbg) {
        const long long t = shader_clock();
        if (rows_p() == 0) dbg[80 + slot] = (double)t;
    }
}

// ---- the data passes ------------------------------------------------------------------------------------------------------
// The correspondences cannot be staged (4 x 48 N bytes per wavefront), so every pass over them is a pass over L2 / MALL, and with the
// lane-sparse middle four times cheaper those passes are what the kernel waits for.  There are three of them (the one-triplet kernel makes
// eight over its LDS copy): centroids | mean distances + all 96 moment sums | cheirality votes of both essential matrices, with the t3-scale
// sums of the main candidates riding along (a separate t3-scale pass only for a row whose pick is not a main candidate).

// Pass 1, Normalize2Ddata.m:33: points0 = mean(points,2) for the three views; every lane of the row ends with c[0..5].
// Nothing but loads and six additions per correspondence: four trips' loads are issued before the first is consumed.
__device__ __forceinline__ void rows_centroids(const RowSrc& s, const int N, double (&c)[6]) {
    const int p = rows_p();
    double sm[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += 4 * ROWL) {                               // wave-uniform trip count; lanes past the end add zeros
        Pt6 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + ROWL * u + p; q[u] = rows_load(s, (i < N) ? i : 0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool have = i0 + ROWL * u + p < N;
#pragma unroll
            for (int k = 0; k < 6; ++k) sm[k] += have ? q[u].v[k] : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) c[k] = row_sum16(sm[k]) / (double)N;
}

// Pass 2, Normalize2Ddata.m:35-39 + the 96 moment sums of tft_kernel.h::accumulate_moments, in ONE pass:
//   * the normalised coordinates are s (x - c) with s = sqrt(2) / mean|x - c| (N_matrix = [s 0 -s cx; 0 s -s cy; 0 0 1]), and s is not known
//     before the pass ends -- so the sums are taken over the CENTRED coordinates and every moment is multiplied by its power of the three
//     scales afterwards (a monomial of degree d1 in view 1, d2 in view 2, d3 in view 3 scales by s1^d1 s2^d2 s3^d3: no cancellation);
//   * two lanes share a correspondence: the even lane accumulates the 48 sums of the p1-monomials {x1^2, x1 y1, x1}, the odd lane those of
//     {y1^2, y1, 1} (96 accumulators per lane would not fit the register file; 48 each do, and the pair's loads coalesce into one request);
//   * the pair also splits the distance sums: even lane view 1, odd lane view 3, view 2 alternately (two correspondences per loop iteration).
// Leaves nrm[0..8] (LDS, and nr[] on every lane) and mom[0..95] (LDS).
__device__ __forceinline__ void rows_distances_moments(const RowSrc& s, const int N, const double (&c)[6], double* nrm, double (&nr)[9], double* mom) {
    const int p = rows_p();
    const int slot = p >> 1;
    const bool odd = (p & 1) != 0;
    double acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.0;
    double dA = 0.0, dB = 0.0;
    // (the loads of trips t + 1 and t + 2 are in flight while trip t is consumed: a trip issues ~80 fp64 instructions, an L2 / MALL round trip
    // lasts five times that, and the other wavefront of the SIMD is in the same pass more often than not)
    // (view 2's distance is needed once per correspondence and the pair holds two of them per loop iteration: the even lane takes the first one's
    // square root, the odd lane the second one's -- three square roots per lane and two trips instead of four)
    auto body = [&](const Pt6& q, double& r2_out) {
        const double x1 = q.v[0] - c[0], y1 = q.v[1] - c[1];
        const double x2 = q.v[2] - c[2], y2 = q.v[3] - c[3];
        const double x3 = q.v[4] - c[4], y3 = q.v[5] - c[5];
        const double r1 = x1 * x1 + y1 * y1, r2 = x2 * x2 + y2 * y2, r3 = x3 * x3 + y3 * y3;
        dA += sqrt_nonneg(odd ? r3 : r1);                                    // Normalize2Ddata.m:35
        r2_out = r2;
        const double q2[4] = {1.0, x2, y2, r2};
        const double q3[4] = {1.0, x3, y3, r3};
        const double pa = odd ? y1 * y1 : x1 * x1, pb = odd ? y1 : x1 * y1, pc = odd ? 1.0 : x1;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const double wv = q3[b] * q2[cc];
                acc[4 * b + cc] += pa * wv;
                acc[16 + 4 * b + cc] += pb * wv;
                acc[32 + 4 * b + cc] += pc * wv;
            }
    };
    constexpr int STEP = ROWL / 2;
    Pt6 pe = rows_load(s, (slot < N) ? slot : 0);
    Pt6 po = rows_load(s, (slot + STEP < N) ? slot + STEP : 0);
#pragma unroll 1
    for (int i = slot; i < N; i += 2 * STEP) {
        const Pt6 q = pe;
        if (i + 2 * STEP < N) pe = rows_load(s, i + 2 * STEP);
        double r2a, r2b = 0.0;
        body(q, r2a);
        if (i + STEP < N) {
            const Pt6 r = po;
            if (i + 3 * STEP < N) po = rows_load(s, i + 3 * STEP);
            body(r, r2b);
        }
        dB += sqrt_nonneg(odd ? r2b : r2a);
    }
    // mean distances -> scales and offsets (every lane of the row)
    const double d1 = row_sum16(odd ? 0.0 : dA), d3 = row_sum16(odd ? dA : 0.0), d2 = row_sum16(dB);
    const double r2c = sqrt(2.0);
    const double dd[3] = {d1, d2, d3};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double norm0 = dd[v] / (double)N;                              // :35
        nr[3 * v + 0] = r2c / norm0;                                         // :36
        nr[3 * v + 1] = -r2c * c[2 * v] / norm0;                             // :37
        nr[3 * v + 2] = -r2c * c[2 * v + 1] / norm0;
    }
    if (p < 9) {
        double mine = nr[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) mine = (p == k) ? nr[k] : mine;
        nrm[p] = mine;
    }
    // sums over the eight lanes of equal parity: halving butterfly (masks 8, 4, 2), 48 -> 6 values per lane
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = halve_sum<8>(acc[i], acc[i + 24]);
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = halve_sum<4>(acc[i], acc[i + 12]);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = halve_sum<2>(acc[i], acc[i + 6]);
    // the lane holds local indices base .. base + 5 of its parity's 48 sums; global moment index 48 * parity + local = 16 h + 4 i3 + i2
    const int base = 6 * ((p >> 1) & 1) + 12 * ((p >> 2) & 1) + 24 * ((p >> 3) & 1);
    const double s1 = nr[0], s2 = nr[3], s3 = nr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = 48 * (p & 1) + base + i;
        const int h = e >> 4, i3 = (e >> 2) & 3, i2 = e & 3;
        // degrees: p1 = {x^2, xy, x, y^2, y, 1} -> {2,2,1,2,1,0};  q = {1, x, y, x^2+y^2} -> {0,1,1,2}
        const double f1 = (h == 5) ? 1.0 : ((h == 2 || h == 4) ? s1 : s1 * s1);
        const double f3 = (i3 == 0) ? 1.0 : ((i3 == 3) ? s3 * s3 : s3);
        const double f2 = (i2 == 0) ? 1.0 : ((i2 == 3) ? s2 * s2 : s2);
        mom[e] = acc[i] * (f1 * f3 * f2);
    }
}

// Smallest eigenvector of the row's own n x n matrix: row_min_eigvec's algorithm (row_eig.h: same Cholesky, substitutions, stopping tests and
// gap estimate), four independent problems per wavefront.  What differs is the bookkeeping around the DPP chains:
//   * the factor goes through the ROW's LDS workspace in packed form -- strictly lower triangle, entry (r, c < r) at r (r - 1) / 2 + c, one
//     slot that holds 0.0 for everything on and above the diagonal: n (n - 1) / 2 + 1 doubles instead of n * n;
//   * the loop runs until every row of the wavefront has stopped (a row that is done keeps its iterate);
//   * position p returns components p (x0) and 16 + p (x1) of the unit eigenvector.
// g0 / g1 / d0 / d1, start0 / start1, *iters, *resid2, *gram_risk: as row_min_eigvec.
template <int n>
__device__ __forceinline__ void rows_min_eigvec(double (&g0)[RowEigDims<n>::N0], double (&g1)[RowEigDims<n>::N1], const double d0, const double d1,
                                                double* Lp, const int maxit, int* iters, double* resid2, const bool has_start,
                                                const double start0, const double start1, double* gram_risk, double& x0, double& x1,
                                                const double gram_risk_limit2 = 1e14) {
    constexpr int N0 = RowEigDims<n>::N0;
    constexpr bool HI = RowEigDims<n>::HI;
    constexpr int Z = n * (n - 1) / 2;                                       // the zero slot
    const int p = rows_p();
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
    {
        const int t0 = (p * (p - 1)) / 2, t1 = ((16 + p) * (15 + p)) / 2;
#pragma unroll
        for (int c = 0; c < N0; ++c) Lp[(c < p && valid0) ? t0 + c : Z] = g0[c];     // (everything else is 0.0 and lands on the zero slot)
        if constexpr (HI) {
#pragma unroll
            for (int c = 0; c < n - 1; ++c) Lp[(c < 16 + p && valid1) ? t1 + c : Z] = g1[c];
        }
    }
    wave_sync();
    x0 = valid0 ? rsqrt((double)n) : 0.0;
    x1 = valid1 ? rsqrt((double)n) : 0.0;
    if (has_start) {                                        // a zero / non-finite guess falls back to the uniform vector
        const double s0 = valid0 ? start0 : 0.0, s1 = valid1 ? start1 : 0.0;
        const double nn0 = row_sum16(s0 * s0 + s1 * s1);
        if (nn0 > 1e-300 && nn0 < 1e300) { const double r0 = rsqrt(nn0); x0 = s0 * r0; x1 = s1 * r0; }
    }
    // the position's own COLUMNS of L' (c0[j] = L'[j][p], c1[j] = L'[j][16 + p]) stay in registers for all iterations
    double c0[n], c1[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        c0[j] = Lp[(valid0 && p < j) ? (j * (j - 1)) / 2 + p : Z];
        c1[j] = (HI && j > 16) ? Lp[(valid1 && 16 + p < j) ? (j * (j - 1)) / 2 + 16 + p : Z] : 0.0;
    }
    double rprev2 = 1.0, res = 1.0, rk_r2 = 0.0, rk_rp = 1.0, rk_nn = 0.0;
    int it = 0;
    bool done = false;
#pragma unroll 1
    while (true) {
        double y0 = x0 * myinv0, y1 = x1 * myinv1;
        if constexpr (HI) {                                 // the lo rows come back from LDS every iteration (row_eig.h: register demand peaks below)
            const double* rows = Lp + opaque_int(0);
            const int t0 = (p * (p - 1)) / 2;
#pragma unroll
            for (int c = 0; c < N0; ++c) g0[c] = rows[(c < p) ? t0 + c : Z];
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
        if (!wave_any(!done)) break;                        // the four rows iterate on four different matrices
    }
    *iters = it;
    *resid2 = res;
    if (gram_risk) {
        const double num = 4.0 * rk_nn * rk_r2 * rk_rp;
        const double d = rk_rp - rk_r2;
        const double den = (rk_r2 < rk_rp) ? d * d : 0.0;
        *gram_risk = (tr * tr * num < gram_risk_limit2 * den) ? 0.0 : 1.0;
    }
}

// transform_TFT.m:42-49 with inverse = 1 (pose_common.h::transform_tft_inverse), 27 entries on the 16 lanes of a row: positions p and 16 + p
template <class MatFn>
__device__ __forceinline__ void rows_transform_tft_inverse(const double* to, double* tn, double* mats, MatFn matrix_of) {
    const int p = rows_p();
    if (p < 3) {
        Mat3 M = matrix_of(p);
        if (p > 0) M = mat3_inv(M);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) mats[9 * p + 3 * r + c] = M.m[r][c];
    }
    wave_sync();
    double val[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = 16 * h + p;
        const bool have = e < 27;
        const int ee = have ? e : 0;
        const int i = ee / 9, k = (ee % 9) / 3, j = ee % 3;                  // entry T(j,k,i)
        const double m0 = mats[i], m1 = mats[3 + i], m2 = mats[6 + i];       // M1(:,i)
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double mix = m0 * to[c + 3 * d] + m1 * to[c + 3 * d + 9] + m2 * to[c + 3 * d + 18];
                v += mats[9 + 3 * j + c] * mix * mats[18 + 3 * k + d];
            }
        val[h] = have ? v : 0.0;
    }
    const double nn = row_sum16(val[0] * val[0] + val[1] * val[1]);
    const double rs = rsqrt(nn);
    wave_sync();
    tn[p] = val[0] * rs;
    if (p < 11) tn[16 + p] = val[1] * rs;
    wave_sync();
}

// linearTFT.m:64-91 from the moment sums of the row's triplet (tft_kernel.h::linear_tft_middle, fast tier): w->t = the constrained tensor.
// Returns (per lane, the same on every lane of a row) false when a fast tier could not finish.
__device__ __forceinline__ bool rows_linear_tft_middle(RowLds* w, double* dbg, const bool want_P = false) {
    const int p = opaque_lane_int(rows_p());
    bool ok = true;
    int it1 = 0, it2 = 0;
    {                                                                        // :64-67
        const bool hi = p < 11;
        double ga[27], gb[27], g0[16], da, db, r2, risk, x0, x1;
        gram_row27(w->mom, p, ga, da);
        gram_row27(w->mom, hi ? 16 + p : 0, gb, db);
#pragma unroll
        for (int c = 0; c < 16; ++c) g0[c] = ga[c];
#pragma unroll
        for (int c = 0; c < 27; ++c) gb[c] = hi ? gb[c] : 0.0;
        rows_stamp(dbg, 3);
        rows_min_eigvec<27>(g0, gb, da, hi ? db : 0.0, w->ov, EIG_MAXIT, &it1, &r2, false, 0.0, 0.0, &risk, x0, x1);
        ok = ok && eig_converged(r2) && risk == 0.0;
        wave_sync();
        w->t[p] = x0;
        if (hi) w->t[16 + p] = x1;
        wave_sync();
    }
    if (dbg) { dbg[p] = w->t[p]; if (p < 11) dbg[16 + p] = w->t[16 + p]; }
    rows_stamp(dbg, 4);
    const bool eok = epipoles_from_tensor<16, false>(w->t, w->ov, w->epi, false);   // :71-79 (slice null vectors in the overlay)
    ok = !row_any(!eok) && ok;
    rows_stamp(dbg, 5);
    if (dbg && p < 6) dbg[27 + p] = w->epi[p];
    if (p == 0) frame_of(w->epi, w->Q);                                      // Q2 from e21
    if (p == 1) frame_of(w->epi + 3, w->Q + 9);                              // Q3 from e31
    wave_sync();
    // Gp = Up' G Up (15x15), lower triangle, packed into the overlay; entry (a,b), a = 5 i + m
    double* Gp = w->ov;
#pragma unroll 1
    for (int e = p; e < 120; e += ROWL) {
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
    rows_stamp(dbg, 6);
    {                                                                        // :84
        double g[15], none[1] = {0.0}, diag = 0.0, x0, x1, r2, risk;
        const bool have = p < 15;
        const int r = have ? p : 0;
#pragma unroll
        for (int c = 0; c < 15; ++c) { g[c] = (c <= r && have) ? Gp[tri_index(r, c)] : 0.0; diag = (c == r) ? g[c] : diag; }
        // start from the unconstrained solution projected onto range(E): tp0 = Up' t (tft_kernel.h)
        double tp0 = 0.0;
        if (have) {
            const int i = r / 5, m = r % 5, jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) tp0 += w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk] * w->t[j + 3 * k + 9 * i];
        }
        wave_sync();                                                         // Gp is read; the factor may overwrite it
        rows_min_eigvec<15>(g, none, diag, 0.0, w->ov, EIG_MAXIT, &it2, &r2, true, tp0, 0.0, &risk, x0, x1);
        ok = ok && eig_converged(r2) && risk == 0.0;
        if (have) w->tp[p] = x0;
        wave_sync();
    }
    rows_stamp(dbg, 7);
    {                                                                        // t = Up * tp   (:85)
        double tv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = (16 * h + p < 27) ? 16 * h + p : 0;
            const int i = e / 9, k = (e % 9) / 3, j = e % 3;
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int jj = (m < 3) ? 0 : m - 2, kk = (m < 3) ? m : 0;
                acc += w->Q[3 * j + jj] * w->Q[9 + 3 * k + kk] * w->tp[5 * i + m];
            }
            tv[h] = acc;
        }
        wave_sync();
        w->t[p] = tv[0];
        if (p < 11) w->t[16 + p] = tv[1];
        wave_sync();
    }
    if (dbg) {
        dbg[33 + p] = w->t[p];
        if (p < 11) dbg[33 + 16 + p] = w->t[16 + p];
        if (p == 0) { dbg[69] = (double)it1; dbg[70] = (double)it2; }
    }
    if (want_P) {                                                            // a = pinv(E) t (:86), see tft_kernel.h::linear_tft_middle
        if (p < 3) {
            const int i = p;
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
    }
    return ok;
}

__device__ __forceinline__ void rows_recover_prepare(RowLds* w, RowRt* rt);
// (EXACT: the certified / one-sided-Jacobi null vectors of small_la.h instead of the fast tier that only reports)
// R_t_from_TFT.m:44-58 and svd(E), candidate poses and cameras (:85-88) for the row's triplet (tft_kernel.h::rt_prepare + recover_prepare)
template <bool EXACT = false>
__device__ __forceinline__ bool rows_rt_prepare(RowLds* w, RowRt* rt, double* dbg) {
    const int p = rows_p();
    rows_transform_tft_inverse(rt->T1, rt->T2, rt->mats, [w](int v) { return load_K(w->calm, v); });   // :44
    const bool eok = epipoles_from_tensor<16, EXACT>(rt->T2, rt->nullv, w->epi, true);                 // :47-55
    const bool ok = !row_any(!eok);
    if (p < 2) {
        const double* e21 = w->epi; const double* e31 = w->epi + 3;
        Mat3 M;                                                              // [T1*e T2*e T3*e]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) acc += (p == 0) ? rt->T2[r + 3 * q + 9 * i] * e31[q] : rt->T2[q + 3 * r + 9 * i] * e21[q];
                M.m[r][i] = acc;
            }
        const double* e = (p == 0) ? e21 : e31;
        const double sg = (p == 0) ? 1.0 : -1.0;                             // E31 = -crossM(epi31)*[...]  (:58)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rt->Ein[9 * p + 0 + c] = sg * (-e[2] * M.m[1][c] + e[1] * M.m[2][c]);
            rt->Ein[9 * p + 3 + c] = sg * (e[2] * M.m[0][c] - e[0] * M.m[2][c]);
            rt->Ein[9 * p + 6 + c] = sg * (-e[1] * M.m[0][c] + e[0] * M.m[1][c]);
        }
    }
    wave_sync();
    rows_stamp(dbg, 9);
    rows_recover_prepare(w, rt);
    return ok;
}

// recover_R_t up to the candidate cameras (R_t_from_TFT.m:84-88 == LinearFPoseEstimation.m:86-90) from the two essential matrices in rt->Ein
__device__ __forceinline__ void rows_recover_prepare(RowLds* w, RowRt* rt) {
    const int p = rows_p();
    if (p < 2) {                                                             // svd(E), R = U W V', Rp = U W' V', t = U(:,3)
        Mat3 E, U, V;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) E.m[r][c] = rt->Ein[9 * p + 3 * r + c];
        double sv[3];
        svd3(E, U, V, sv);
        Mat3 UW, UWt, Vt = mat3_T(V);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            UW.m[r][0] = U.m[r][1];  UW.m[r][1] = -U.m[r][0]; UW.m[r][2] = U.m[r][2];
            UWt.m[r][0] = -U.m[r][1]; UWt.m[r][1] = U.m[r][0]; UWt.m[r][2] = U.m[r][2];
        }
        Mat3 R = mat3_mul(UW, Vt), Rp = mat3_mul(UWt, Vt);
        const double sR = sgn(mat3_det(R)), sRp = sgn(mat3_det(Rp));        // :87
        double* c = rt->cand[p];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { c[3 * r + cc] = R.m[r][cc] * sR; c[9 + 3 * r + cc] = Rp.m[r][cc] * sRp; }
        c[18] = U.m[0][2]; c[19] = U.m[1][2]; c[20] = U.m[2][2];            // t = U(:,3)   (:88)
    }
    wave_sync();
    if (p < 4) {                                                             // cameras of (R,t) and (Rp,t) for both calls
        const int call = p >> 1, cd = p & 1;
        const Mat3 K = load_K(w->calm, call + 1);
        compose_camera(K, rt->cand[call] + 9 * cd, rt->cand[call] + 18, rt->P[p]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rt->candRt[p][4 * r + c] = rt->cand[call][9 * cd + 3 * r + c];
            rt->candRt[p][4 * r + 3] = rt->cand[call][18 + r];
        }
    }
    if (p == 4) {                                                            // P1 = K1 [I | 0]
        const Mat3 K1 = load_K(w->calm, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) { rt->Pfin[0][4 * r] = K1.m[r][0]; rt->Pfin[0][4 * r + 1] = K1.m[r][1]; rt->Pfin[0][4 * r + 2] = K1.m[r][2]; rt->Pfin[0][4 * r + 3] = 0.0; }
    }
    wave_sync();
}

// Pass 3: cheirality votes (R_t_from_TFT.m:91-104) for both essential matrices in one pass over the row's correspondences, with the certified
// sign-only fast tier of pose_common.h (vote_one).  score(R,-t) = -score(R,t) exactly (pose_common.h), so each essential matrix has two scores
// to find: sR = score(R,t) and sRp = score(Rp,t).  The reference's selection loop needs them only as far as this:
//     |score| = 2 N  <=>  every correspondence votes +2, or every one -2;
//     if one candidate of the pair reaches |s| = 2 N and the other is KNOWN to stay below 2 N in magnitude, the loop's pick (order k = 1..4,
//     `>=`, start 0) does not depend on the other's value: it takes the sign variant of the first that scores +2 N.
// Hence: the first trip (16 correspondences) evaluates all four candidates; a candidate that shows a certified vote != +2 AND a certified vote
// != -2 there is proven |s| < 2 N.  When exactly one candidate of a pair is proven so, only its partner ("main", A) is evaluated over the
// remaining trips -- half the work of the pass on well-posed triplets, whose true rotation votes 2 N and whose twisted partner splits at once.
// If the main candidate then fails to reach 2 N (noisy or degenerate data), the other one (B) is evaluated in a second sweep: every outcome is
// decided on exact, certified scores.  all4 (debug entry points): always evaluate all four, so that the four scores can be reported.
// The trip loops are wave-uniform (lanes past the end work on correspondence 0 and are masked out): the four rows decide independently, the
// wavefront skips a candidate only when no row needs it.  Cameras are re-read from the row's LDS workspace per trip (row-uniform addresses:
// broadcast reads that cost no VALU slot).
// Returns false (per row) when a needed vote could not be certified.  sc[call][0] = sR, sc[call][1] = sRp; a score that was not needed is
// reported as 0 (it is below 2 N in magnitude and its partner is +-2 N: same pick).
// The sums of the t3 scale (R_t_from_TFT.m:68-74) ride along (RowScale, SPEC): they need the two-view point of (P1, P2) for the CHOSEN second
// camera and the chosen third camera, neither known before the votes are in -- but the main candidates are the choice on every triplet whose votes
// are unanimous, the point is the null vector of the very system vote_one has just factored (dlt_from_vote: no second build, no second
// factorisation), and a candidate's sign variant (R, -t) mirrors everything exactly: X -> diag(1,1,1,-1) X, so num -> s2 s3 num, den -> den.
// rows_pose_tail takes the speculative sums when the picks' rotations are the main candidates' and makes the separate pass otherwise.
struct RowScale { double num, den; bool conv; bool main1[2]; };
template <bool SPEC = false>
__device__ __forceinline__ bool rows_votes(const RowSrc& s, const int N, const RowRt* rt, const bool all4, int (&sc)[2][2], int* sweeps_out = nullptr,
                                           bool* cert_out = nullptr, RowScale* scale = nullptr) {
    const int p = rows_p();
    double PA[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) PA[c] = rt->Pfin[0][c];                     // PA[3] = PA[7] = PA[11] = 0
    int scA[2] = {0, 0}, scB[2] = {0, 0};
    bool certA[2] = {true, true}, certB[2] = {true, true};
    int offA[2] = {0, 24}, offB[2] = {12, 36};                              // doubles into RowRt::P / RowRt::candRt: A = (R,t), B = (Rp,t) to begin with
    bool main1[2] = {false, false};                                          // the row's main candidate is Rp
    bool evalA = true, evalB[2] = {true, true};                              // wave-uniform: what the running sweep evaluates
    bool fullB[2] = {true, true};                                            // wave-uniform: B covers every trip
    bool rowB[2] = {true, true};                                             // per ROW: this row's pick needs B's score (what a neighbour needs must not reach this row's result)
    int sweeps = 0;
    double snum = 0.0, sden = 0.0;
    bool sconv = true;
#pragma unroll 1
    for (int sweep = 0; sweep < 2; ++sweep) {
        ++sweeps;
        const int start = (sweep == 0) ? 0 : ROWL;
        Pt6 pnext = rows_load(s, (start + p < N) ? start + p : 0);
#pragma unroll 1
        for (int i0 = start; i0 < N; i0 += ROWL) {
            const Pt6 q = pnext;
            const bool have = i0 + p < N;
            pnext = rows_load(s, (i0 + ROWL + p < N) ? i0 + ROWL + p : 0);
            const double x1 = q.v[0], y1 = q.v[1];
            double a0[3], a1[3], SA[6];                                      // rows [0 -1 y; 1 0 -x] * P1 and their A'A   (triangulation3D.m:58-59)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a0[c] = y1 * PA[8 + c] - PA[4 + c]; a1[c] = PA[c] - x1 * PA[8 + c]; }
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int c = 0; c <= rr; ++c) SA[rr * (rr + 1) / 2 + c] = a0[rr] * a0[c] + a1[rr] * a1[c];
            const bool first = sweep == 0 && i0 == 0;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int call = 1 - cc;                                     // view 3 first: the scale sums of view 2's main candidate need view 3's
                const double x2 = (call == 0) ? q.v[2] : q.v[4], y2 = (call == 0) ? q.v[3] : q.v[5];
                VoteFactor fA, fB;
                if (evalA) {
                    VoteCam cam;
                    const int off = opaque_lane_int(offA[call]);
                    const double* pb = rt->P[0] + off;
                    const double* pr = rt->candRt[0] + off;
#pragma unroll
                    for (int c = 0; c < 12; ++c) cam.PB[c] = pb[c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) cam.R3[c] = pr[8 + c];
                    int term = 0;
                    bool cert = true;
                    // (view 2's main candidate past the first trip: its two signs come from the converged point of the scale sums below)
                    if (SPEC && call == 0 && !first) vote_one<true, false>(SA, cam, x2, y2, term, cert, &fA);
                    else if (SPEC && call == 0) vote_one<true>(SA, cam, x2, y2, term, cert, &fA);
                    else vote_one(SA, cam, x2, y2, term, cert);
                    scA[call] += have ? term : 0;
                    certA[call] = certA[call] && (cert || !have);
                }
                if (evalB[call]) {
                    VoteCam cam;
                    const int off = opaque_lane_int(offB[call]);
                    const double* pb = rt->P[0] + off;
                    const double* pr = rt->candRt[0] + off;
#pragma unroll
                    for (int c = 0; c < 12; ++c) cam.PB[c] = pb[c];
#pragma unroll
                    for (int c = 0; c < 4; ++c) cam.R3[c] = pr[8 + c];
                    int term = 0;
                    bool cert = true;
                    if (SPEC && call == 0 && first) vote_one<true>(SA, cam, x2, y2, term, cert, &fB);
                    else vote_one(SA, cam, x2, y2, term, cert);
                    scB[call] += have ? term : 0;
                    certB[call] = certB[call] && (cert || !have);
                }
                if (first) {                                                 // after the first trip: which candidates are already out of the race?
                    // the accumulators hold the vote of this lane's first correspondence
                    // (every ballot is taken by the whole wavefront: no short-circuit between them)
                    const bool v0 = row_any(have && certA[call] && scA[call] != 2), v1 = row_any(have && certA[call] && scA[call] != -2);
                    const bool b0 = row_any(have && certB[call] && scB[call] != 2), b1 = row_any(have && certB[call] && scB[call] != -2);
                    const bool mixA = v0 && v1, mixB = b0 && b1;
                    const bool single = !all4 && (mixA != mixB);
                    main1[call] = single && mixA;                            // (R,t) is out: (Rp,t) is the row's main candidate
                    if (main1[call]) {
                        const int ts = scA[call]; scA[call] = scB[call]; scB[call] = ts;
                        const bool tc = certA[call]; certA[call] = certB[call]; certB[call] = tc;
                        offA[call] = 12 + 24 * call; offB[call] = 24 * call;
                        if (SPEC && call == 0) fA = fB;
                    }
                    rowB[call] = !single;
                    evalB[call] = wave_any(!single);                         // some row of the wavefront needs both candidates of this pair
                    fullB[call] = evalB[call];
                }
                if (SPEC && call == 0 && evalA) {                            // scale sums with the main candidates (R_t_from_TFT.m:70-73)
                    double X[4];
                    const bool conv = dlt_from_vote(fA, X);
                    sconv = sconv && (conv || !have);
                    if (!first) {                                            // the vote's exact tier (pose_common.h::tri_vote_exact): signs of the converged point
                        const double* pr = rt->candRt[0] + opaque_lane_int(offA[0]);
                        const double s4 = sgn(X[3]);                         // X1 = X ./ X(4)
                        const double d1 = X[2] * s4, d2 = (pr[8] * X[0] + pr[9] * X[1] + pr[10] * X[2] + pr[11] * X[3]) * s4;
                        scA[0] += have ? (int)sgn(d1) + (int)sgn(d2) : 0;
                        certA[0] = certA[0] && (conv || !have);
                    }
                    const double iw = 1.0 / X[3];
                    const double X0 = X[0] * iw, X1 = X[1] * iw, X2 = X[2] * iw;
                    const double* ax = rt->P[0] + opaque_lane_int(offA[1]);  // K3 [R3 | t3] of view 3's main candidate
                    double X3[3], u3[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { X3[r] = ax[4 * r] * X0 + ax[4 * r + 1] * X1 + ax[4 * r + 2] * X2; u3[r] = ax[4 * r + 3]; }
                    const double p3[3] = {q.v[4], q.v[5], 1.0};
                    double c1[3], c2[3];
                    cross3(p3, X3, c1);
                    cross3(p3, u3, c2);
                    snum += have ? c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2] : 0.0;
                    sden += have ? c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2] : 0.0;
                }
            }
        }
        if (sweep == 1) break;
        // a main candidate that did not reach 2 N: its partner's score decides, evaluate it over the trips it skipped (rare)
        bool again = false;
#pragma unroll
        for (int call = 0; call < 2; ++call) {
            const int tA = (int)row_sum16((double)scA[call]);                // |score| <= 2 N: exact
            const bool mine = !rowB[call] && tA != 2 * N && tA != -2 * N;    // this row's main candidate fell short: its partner's score decides
            rowB[call] = rowB[call] || mine;
            const bool need = wave_any(!fullB[call] && tA != 2 * N && tA != -2 * N);
            evalB[call] = need;
            fullB[call] = fullB[call] || need;
            again = again || need;
        }
        evalA = false;
        if (!again) break;
    }
    bool ok = true;
#pragma unroll
    for (int call = 0; call < 2; ++call) {
        const int tA = (int)row_sum16((double)scA[call]);
        // B's score and certificate count for a row only when the ROW needs them -- B may have been evaluated in full because a neighbour did: a
        // triplet's result (and whether it is handed to the exact kernel) never depends on the wavefront it travels in
        const int tBsum = (int)row_sum16((double)scB[call]);
        const int tB = (fullB[call] && rowB[call]) ? tBsum : 0;
        const bool badA = row_any(!certA[call]), badB = row_any(!certB[call]);
        ok = ok && !badA && (!(fullB[call] && rowB[call]) || !badB);
        if (cert_out) {                                                      // (with all4: A = (R,t), B = (Rp,t), both evaluated over every trip)
            cert_out[2 * call] = main1[call] ? !badB : !badA;
            cert_out[2 * call + 1] = main1[call] ? !badA : !badB;
        }
        sc[call][0] = main1[call] ? tB : tA;
        sc[call][1] = main1[call] ? tA : tB;
    }
    if constexpr (SPEC) {
        scale->num = row_sum16(snum);
        scale->den = row_sum16(sden);
        scale->conv = !row_any(!sconv);
        scale->main1[0] = main1[0];
        scale->main1[1] = main1[1];
    }
    if (sweeps_out) *sweeps_out = sweeps + (fullB[0] ? 16 : 0) + (fullB[1] ? 32 : 0);   // (debug: sweeps made, which pairs were evaluated in full)
    return ok;
}

// One pass over the row's correspondences with the fast DLT tier (pose_common.h::tri_pass_fast): MODE TRI_SCALE -> num / den of
// R_t_from_TFT.m:72-73 (every lane of the row), TRI_RECONST -> dehomogenised points to out (3 x N).  Returns false (per row) when some
// correspondence's inverse iteration hit its cap.  out is per ROW: nullptr for a row that stores nothing (a tail row, a failed triplet whose
// outputs are already NaN, a row the caller only carries along) -- its lanes still make the pass, the trip count is the wavefront's.
template <int MODE, bool EXACT = false>
__device__ __forceinline__ bool rows_tri_pass(const RowSrc& s, const int N, const double* camA, const double* camB, const double* aux,
                                              double* out, double& num_out, double& den_out) {
    const int p = rows_p();
    // EXACT: the cameras are read from LDS where they are used, so that none of their 36 values stays live across the out-of-line tiers of
    // dlt_point (pose_common.h::tri_pass_impl)
    double PAr[12], PBr[12], AXr[12];
    if (!EXACT) {
#pragma unroll
        for (int c = 0; c < 12; ++c) { PAr[c] = camA[c]; PBr[c] = camB[c]; AXr[c] = aux[c]; }
    }
    typedef const double (&cam_ref)[12];
    cam_ref PA = EXACT ? *reinterpret_cast<const double(*)[12]>(camA) : PAr;
    cam_ref PB = EXACT ? *reinterpret_cast<const double(*)[12]>(camB) : PBr;
    cam_ref AX = EXACT ? *reinterpret_cast<const double(*)[12]>(aux) : AXr;
    bool all_conv = true;
    double num = 0.0, den = 0.0;
    Pt6 pnext = rows_load(s, (p < N) ? p : 0);
#pragma unroll 1
    for (int i = p; i < N; i += ROWL) {
        const Pt6 q = pnext;
        if (i + ROWL < N) pnext = rows_load(s, i + ROWL);
        double X[4];
        const bool conv = dlt_point<EXACT, EXACT>(PA, PB, AX, camA, camB, aux, MODE == TRI_RECONST, q.v[0], q.v[1], q.v[2], q.v[3], q.v[4], q.v[5], X);
        all_conv = all_conv && conv;
        const double iw = 1.0 / X[3];
        const double X0 = X[0] * iw, X1 = X[1] * iw, X2 = X[2] * iw;         // X./X(4)
        if constexpr (MODE == TRI_SCALE) {
            double X3[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) X3[r] = AX[4 * r] * X0 + AX[4 * r + 1] * X1 + AX[4 * r + 2] * X2;   // X3 = K3*R3*X  (:71)
            const double u3[3] = {AX[3], AX[7], AX[11]};
            const double p3[3] = {q.v[4], q.v[5], 1.0};
            double c1[3], c2[3];
            cross3(p3, X3, c1);
            cross3(p3, u3, c2);
            num += c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
            den += c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
        } else if (out) {                                                    // (per row: a row that must not store passes nullptr)
            out[3 * (long)i + 0] = X0;
            out[3 * (long)i + 1] = X1;
            out[3 * (long)i + 2] = X2;
        }
    }
    if constexpr (MODE == TRI_SCALE) {
        num_out = row_sum16(num);
        den_out = row_sum16(den);
    }
    return !row_any(!all_conv);
}

// ---- what the rows kernels share around their linear stage ------------------------------------------------------------------------
struct RowJob {
    long b;                // the row's triplet (a tail row repeats the last one)
    bool valid;            // ... and stores nothing
    bool bad_index;        // sampled hypotheses: an index outside the scene
    double* dbg;
    RowSrc src;
};
// the row's triplet, its correspondences and calibration (-> w->calm)
__device__ __forceinline__ RowJob rows_begin(const LinearTftArgs& a, RowLds* w, const long blk, const int N) {
    const int lane = lane_id(), p = lane & 15, row = lane >> 4;
    RowJob j;
    const long b_raw = blk * ROW_TRIPLETS + row;
    j.valid = b_raw < a.B;
    j.b = j.valid ? b_raw : a.B - 1;
    j.dbg = a.dbg ? a.dbg + j.b * DBG_STRIDE : nullptr;
    j.src.idx = a.sample_idx ? a.sample_idx + j.b * (long)N : nullptr;
    j.src.pts = a.sample_idx ? a.corresp : a.corresp + j.b * 6 * (long)N;
    j.src.ns = a.sample_ns;
    j.src.sampled = a.sample_idx != nullptr;
    wave_sync();
    j.bad_index = false;
    if (a.sample_idx) {
        bool bad = false;
        for (int i = p; i < N; i += ROWL) { const int k = j.src.idx[i]; bad = bad || !(k >= 0 && k < j.src.ns); }
        j.bad_index = row_any(bad);
    }
    w->calm[p] = a.calm[j.b * a.calm_stride + p];
    if (p < 11) w->calm[16 + p] = a.calm[j.b * a.calm_stride + 16 + p];
    return j;
}
__device__ __forceinline__ void rows_store_nan(const LinearTftArgs& a, const RowJob& j, const int N) {
    const int p = rows_p();
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (j.valid) {
        if (p < 12) { a.Rt2[j.b * 12 + p] = qnan; a.Rt3[j.b * 12 + p] = qnan; }
        a.T[j.b * 27 + p] = qnan;
        if (p < 11) a.T[j.b * 27 + 16 + p] = qnan;
        if (a.reconst) for (int i = p; i < 3 * N; i += ROWL) a.reconst[j.b * 3 * (long)N + i] = qnan;
    }
}

// Everything after the candidate cameras (rows_recover_prepare): cheirality votes and the reference's selection (R_t_from_TFT.m:91-104 ==
// LinearFPoseEstimation.m:93-107), t3 scale (:68-74 == :64-70), optional Reconst, stores.  T_FROM_CAMERAS: T = TFT_from_P(K1 [I|0], K2 R_t_2,
// K3 R_t_3) (LinearFPoseEstimation.m:78, TFT_from_P.m:25-33); else the tensor in rt->T1.  Returns the row's status.
// exact tier of one candidate's cheirality score for the row (pose_common.h::tri_vote_exact: every correspondence from its converged homogeneous
// DLT point, R_t_from_TFT.m:98-99).  Rolled, cameras from LDS: it runs for the candidates whose fast vote was not certified.
__device__ __forceinline__ int rows_vote_exact(const RowSrc& s, const int N, const RowRt* rt, const int cand_off, const int view) {
    const int p = rows_p();
    typedef const double (&cam_ref)[12];
    cam_ref PA = *reinterpret_cast<const double(*)[12]>(rt->Pfin[0]);
    const double* camB = rt->P[0] + cand_off;
    cam_ref PB = *reinterpret_cast<const double(*)[12]>(camB);
    const double* pr = rt->candRt[0] + cand_off;
    double score = 0.0;
#pragma unroll 1
    for (int i = p; i < N; i += ROWL) {
        const Pt6 q = rows_load(s, i);
        double X[4];
        dlt_point<true, true>(PA, PB, PB, rt->Pfin[0], camB, camB, false, q.v[0], q.v[1], (view == 1) ? q.v[2] : q.v[4], (view == 1) ? q.v[3] : q.v[5], 0.0, 0.0, X);
        const double s4 = sgn(X[3]);                                         // X1 = X ./ X(4)
        const double d1 = X[2] * s4, d2 = (pr[8] * X[0] + pr[9] * X[1] + pr[10] * X[2] + pr[11] * X[3]) * s4;
        score += sgn(d1) + sgn(d2);
    }
    return (int)row_sum16(score);
}

// Minimal samples (N <= 8: half of a row's sixteen positions would idle above): the two candidates of one essential matrix side by side --
// positions 0..7 score candidate k, positions 8..15 candidate k + 1, correspondence p & 7.  Per correspondence the same arithmetic as
// rows_vote_exact, and the scores are sums of +-1: bit-identical results (config 4: exact votes were 30 % / 16 % of the eight- / seven-point kernels).
__device__ __forceinline__ void rows_vote_exact_pair(const RowSrc& s, const int N, const RowRt* rt, const int k, const int view, int& exA, int& exB) {
    const int p = rows_p(), half = p >> 3, i = p & 7;
    const int cand_off = 12 * (k + half);
    typedef const double (&cam_ref)[12];
    cam_ref PA = *reinterpret_cast<const double(*)[12]>(rt->Pfin[0]);
    const double* camB = rt->P[0] + cand_off;
    cam_ref PB = *reinterpret_cast<const double(*)[12]>(camB);
    const double* pr = rt->candRt[0] + cand_off;
    double score = 0.0;
    if (i < N) {
        const Pt6 q = rows_load(s, i);
        double X[4];
        dlt_point<true, true>(PA, PB, PB, rt->Pfin[0], camB, camB, false, q.v[0], q.v[1], (view == 1) ? q.v[2] : q.v[4], (view == 1) ? q.v[3] : q.v[5], 0.0, 0.0, X);
        const double s4 = sgn(X[3]);                                         // X1 = X ./ X(4)
        const double d1 = X[2] * s4, d2 = (pr[8] * X[0] + pr[9] * X[1] + pr[10] * X[2] + pr[11] * X[3]) * s4;
        score = sgn(d1) + sgn(d2);
    }
    exA = (int)row_sum16(half ? 0.0 : score);
    exB = (int)row_sum16(half ? score : 0.0);
}

// EXACT: every score is evaluated (all four candidates), an uncertified one is recomputed by rows_vote_exact, the t3 scale and Reconst take the
// certified DLT ladder -- the row then fails only on what the caller's exact tiers reported.
template <bool T_FROM_CAMERAS, bool EXACT = false>
__device__ __forceinline__ int rows_pose_tail(const LinearTftArgs& a, RowLds* w, RowRt* rt, const RowJob& j, const int N, bool ok) {
    const int p = rows_p();
    double* dbg = j.dbg;
    const long b = j.b;
    int status = ST_OK;
    RowScale spec;
    spec.num = 0.0; spec.den = 1.0; spec.conv = true; spec.main1[0] = spec.main1[1] = false;
    bool spec_valid = !EXACT;                                                // per row: the picks' rotations are the ones the scale sums were taken with
    double spec_sign = 1.0;
    {                                                                        // recover_R_t, see pose_common.h::recover_vote
        int sc[2][2];
        int sweeps = 0;
        if constexpr (EXACT) {
            bool cert4[4];
            rows_votes(j.src, N, rt, true, sc, &sweeps, cert4);             // all four fast scores and which of them are certified
            rows_stamp(dbg, 3);
            if (N <= 8) {                                                    // (wave-uniform: N is the batch's) both candidates of an essential matrix in one pass
#pragma unroll 1
                for (int call = 0; call < 2; ++call) {
                    const bool needA = !(call ? cert4[2] : cert4[0]), needB = !(call ? cert4[3] : cert4[1]);
                    if (!wave_any(needA || needB)) continue;
                    int exA, exB;
                    rows_vote_exact_pair(j.src, N, rt, 2 * call, call + 1, exA, exB);
                    if (needA) { if (call) sc[1][0] = exA; else sc[0][0] = exA; }
                    if (needB) { if (call) sc[1][1] = exB; else sc[0][1] = exB; }
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {                                // (wave-uniform loop; a row joins when its candidate k needs the exact tier)
                    const bool need = !((k == 0) ? cert4[0] : (k == 1) ? cert4[1] : (k == 2) ? cert4[2] : cert4[3]);
                    if (!wave_any(need)) continue;
                    const int ex = rows_vote_exact(j.src, N, rt, 12 * k, (k < 2) ? 1 : 2);
                    if (need) { if (k == 0) sc[0][0] = ex; else if (k == 1) sc[0][1] = ex; else if (k == 2) sc[1][0] = ex; else sc[1][1] = ex; }
                }
            }
        } else {
            ok = rows_votes<true>(j.src, N, rt, dbg != nullptr && !(a.flags & FLAG_DBG_ADAPTIVE), sc, &sweeps, nullptr, &spec) && ok;   // an uncertified sign: the exact kernel's business
        }
        if (dbg && p == 0) dbg[94] = (double)sweeps;
#pragma unroll
        for (int call = 0; call < 2; ++call) {
            const int sR = sc[call][0], sRp = sc[call][1];
            // reference order: k=1 (R,t), k=2 (R,-t), k=3 (Rp,-t), k=4 (Rp,t)   (:92-104)
            const int score[4] = {sR, -sR, -sRp, sRp};
            int seen = 0, pick = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (score[k] >= seen) { pick = k; seen = score[k]; }
            if (pick < 0) status = ST_NO_POSE;
            if (dbg && p < 4) dbg[60 + 4 * call + p] = (double)((p == 0) ? score[0] : (p == 1) ? score[1] : (p == 2) ? score[2] : score[3]);
            if (p < 12) {
                const int r = p >> 2, c = p & 3;
                const double* R = rt->cand[call] + ((pick >= 2) ? 9 : 0);
                const double tsign = (pick == 1 || pick == 2) ? -1.0 : 1.0;
                rt->Rt[call][p] = (c < 3) ? R[3 * r + c] : tsign * rt->cand[call][18 + r];
            }
            spec_valid = spec_valid && pick >= 0 && ((pick >= 2) == spec.main1[call]);
            spec_sign = (pick == 1 || pick == 2) ? -spec_sign : spec_sign;
        }
        wave_sync();
    }
    rows_stamp(dbg, 11);
    {                                                                        // t3 scale, R_t_from_TFT.m:68-74
        if (p < 2) compose_camera_from_pose(load_K(w->calm, p + 1), rt->Rt[p], rt->Pfin[p + 1]);   // Pfin[1] = K2 [R2|t2]; Pfin[2] = [K3*R3 | K3*t3]
        wave_sync();
        // the sums taken during the votes (rows_votes) when they were taken with the chosen rotations, the separate pass otherwise
        // (wave-uniform branch; a row keeps its own speculative values, so its result does not depend on its neighbours)
        double num = spec_sign * spec.num, den = spec.den;
        bool conv = spec.conv;
        if (dbg && p == 0) dbg[95] = spec_valid ? 1.0 : 0.0;
        if (wave_any(!spec_valid)) {
            double n2, d2;
            const bool c2 = rows_tri_pass<TRI_SCALE, EXACT>(j.src, N, rt->Pfin[0], rt->Pfin[1], rt->Pfin[2], nullptr, n2, d2);
            if (!spec_valid) { num = n2; den = d2; conv = c2; }
        }
        ok = ok && conv;
        const double lam = -num / den;                                       // :72-73
        if (dbg && p == 0) dbg[68] = lam;
        if (p < 3) rt->Rt[1][4 * p + 3] *= lam;                              // :74
        wave_sync();
    }
    rows_stamp(dbg, 12);
    if (a.reconst || T_FROM_CAMERAS) {
        if (p == 0) compose_camera_from_pose(load_K(w->calm, 2), rt->Rt[1], rt->Pfin[2]);           // K3 [R3 | lam t3]
        wave_sync();
    }
    if constexpr (T_FROM_CAMERAS) {                                          // TFT_from_P.m:25-33 (f_kernel.h::tft_from_cameras): 27 determinants, positions p and 16 + p
        double val[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = (16 * h + p < 27) ? 16 * h + p : 0;
            const int i = e / 9, k = (e % 9) / 3, jj = e % 3;
            const int r0 = (i == 0) ? 1 : 0, r1 = (i == 2) ? 1 : 2;         // rows of P1 kept
            double m[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                m[0][c] = rt->Pfin[0][4 * r0 + c];
                m[1][c] = rt->Pfin[0][4 * r1 + c];
                m[2][c] = rt->Pfin[1][4 * jj + c];
                m[3][c] = rt->Pfin[2][4 * k + c];
            }
            val[h] = (16 * h + p < 27) ? ((i == 1) ? -1.0 : 1.0) * det4(m) : 0.0;
        }
        const double rs = rsqrt(row_sum16(val[0] * val[0] + val[1] * val[1]));                     // :33
        rt->T1[p] = val[0] * rs;
        if (p < 11) rt->T1[16 + p] = val[1] * rs;
        wave_sync();
    }
    // poses (row-major in LDS) -> MATLAB column-major 3x4 arrays; a non-finite entry anywhere makes the triplet ST_NONFINITE with ALL-NaN outputs
    double vRt[2], vT[2];
    bool bad = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e24 = 16 * h + p;
        vRt[h] = 0.0; vT[h] = 0.0;
        if (e24 < 24) {
            const int which = e24 / 12, e = e24 % 12, c = e / 3, r = e % 3;
            vRt[h] = rt->Rt[which][4 * r + c];
        }
        if (e24 < 27) vT[h] = rt->T1[e24];
        bad = bad || !(fabs(vRt[h]) <= 1.79e308) || !(fabs(vT[h]) <= 1.79e308);
    }
    const bool nonfinite = row_any(bad);                                     // (the ballot is the whole wavefront's: outside every per-row branch)
    if (a.reconst) {                                                         // LinearTFTPoseEstimation.m:59-60
        double n0, d0;
        // only a row that owns a live triplet stores: a tail row, a failed triplet (its Reconst is NaN) and a row that is merely carried
        // along by a neighbour's exact-tier redo (k_gh_finish_rows) leave Reconst alone -- a triplet's bits never depend on its neighbours
        double* rec = (j.valid && !j.bad_index && !nonfinite) ? a.reconst + b * 3 * (long)N : nullptr;
        const bool conv = rows_tri_pass<TRI_RECONST, EXACT>(j.src, N, rt->Pfin[0], rt->Pfin[1], rt->Pfin[2], rec, n0, d0);
        ok = ok && conv;
    }
    if (j.bad_index) status = ST_TOO_FEW;
    else if (!ok) status = ST_RETRY;                                         // redone by the exact kernel
    else if (nonfinite && status == ST_OK) status = ST_NONFINITE;            // non-finite outputs -> status 2
    if (j.bad_index || (ok && nonfinite)) {
        rows_store_nan(a, j, N);                                             // (stores only for a row that owns a triplet)
    } else {
        const bool store = j.valid && ok;                                    // per row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e24 = 16 * h + p;
            if (e24 < 24 && store) ((e24 >= 12) ? a.Rt3 : a.Rt2)[b * 12 + e24 % 12] = vRt[h];
            if (e24 < 27 && store) a.T[b * 27 + e24] = vT[h];
        }
    }
    rows_stamp(dbg, 13);
    return status;
}

// PRE: the normalisations and moment sums come from k_tft_moments (a.pre) instead of the two data passes: the kernel starts at linearTFT's solve
// and touches the correspondences in the vote pass only (and for Reconst).
template <bool PRE>
__global__ void __launch_bounds__(64, 2) k_linear_tft_pose_rows(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    if (a.retry_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.retry_zero = 0;   // (the counter of the context's next call; this call's was zeroed during the previous one)
    const int p = lane_id() & 15, row = lane_id() >> 4;
    RowLds* w = reinterpret_cast<RowLds*>(smem) + row;
    RowRt* rt = reinterpret_cast<RowRt*>(w->ov);
    for (long blk = blockIdx.x; blk * ROW_TRIPLETS < a.B; blk += gridDim.x) {
        const int N = opaque_int(a.N);
        const RowJob j = rows_begin(a, w, blk, N);
        double* dbg = j.dbg;
        rows_stamp(dbg, 0);
        int status;
        if (N < 7) {                                                         // experiments.m:99 (wave-uniform: N is the batch's)
            status = ST_TOO_FEW;
            rows_store_nan(a, j, N);
        } else {
            if constexpr (PRE) {
                rows_load_pre(a.pre, j.b, w->mom, w->nrm);
                rows_stamp(dbg, 1);
                wave_sync();
                if (dbg && p < 9) dbg[71 + p] = w->nrm[p];
            } else {
                double cen[6], nr[9];
                rows_centroids(j.src, N, cen);                               // LinearTFTPoseEstimation.m:45-47
                rows_stamp(dbg, 1);
                rows_distances_moments(j.src, N, cen, w->nrm, nr, w->mom);
                if (dbg && p < 9) dbg[71 + p] = w->nrm[p];
            }
            wave_sync();
            rows_stamp(dbg, 2);
            bool ok = rows_linear_tft_middle(w, dbg);                        // :50
            rows_stamp(dbg, 8);
            rows_transform_tft_inverse(w->t, rt->T1, rt->mats, [w](int v) { return normal_matrix(w->nrm, v); });   // :53
            ok = rows_rt_prepare(w, rt, dbg) && ok;                          // :56
            rows_stamp(dbg, 10);
            status = rows_pose_tail<false>(a, w, rt, j, N, ok);
        }
        if (p == 0 && j.valid) {
            if (a.iter) a.iter[j.b] = 0;                                     // :62
            a.status[j.b] = status;
            if (status == ST_RETRY && a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1)] = (int)j.b;   // the list the exact kernel walks
        }
    }
}

}  // namespace tff
