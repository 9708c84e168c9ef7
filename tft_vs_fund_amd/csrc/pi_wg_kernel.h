// PiPoseEstimation / PiColPoseEstimation with a workgroup of four wavefronts per triplet for the Gauss-Helmert iteration:
// the layout of gh_wg_kernel.h (k_gh_linear -> k_pi_block<Model> -> k_gh_finish) applied to the Pi models of pi_kernel.h.
// Per-correspondence passes are strided over the 256 threads, the 16 accumulation sweeps are reduced per wavefront and the four
// partials added in a fixed order (they live in the eigenvector storage V and in H, both dead at that point), the 36 x 36 KKT
// system is solved by one wavefront (wave_solve_gj) and the truncated pseudo-inverse (PiCol always, Pi when the elimination
// meets a negligible pivot) by the whole workgroup (block_pinv_solve_sym).  TFF_OPT_KERNEL = 1 selects the fused kernel.
#pragma once
#include "pi_kernel.h"
#include "gh_wg_kernel.h"
#include "wave_trid.h"

namespace tff {

// + SN N: the strong direction(s) of every weight block, see the factored weights below -- Pi (4 x 4 blocks, one near-null direction): n (4), cs,
// n'w; PiCol (5 x 5 blocks, five equations for three independent constraints: TWO near-null directions): n1, n2 (5 each), cs1, cs2, n1'w, n2'w
__host__ __device__ constexpr int pi_sn(int E) { return (E == 4) ? 6 : 14; }
// true when the symmetric n x n matrix M (augmented n x (n+1) array, untouched) has NO eigenvalue in [-4 tol, 4 tol), tol = n eps(|M|_2) MATLAB's pinv
// tolerance: pinv(M) then truncates nothing and equals inv(M).  A copy of M (V: n (n+1) doubles) is reduced to tridiagonal form (wave_trid.h: an orthogonal
// similarity; its backward error, a few u |M|, is ~0.05 tol -- a direction pinv would truncate shows up well inside the interval, and the factor 4 also covers a
// largest eigenvalue that sits within rounding of a power of two, where eps(|M|_2) is a coin toss for anyone.  Measured at 10 000 x 200, where the Schur
// complement of the constraints puts the smallest |eigenvalue| at tol-scale -- it falls like 1 / N, tol grows like N --: 52 % of the scenes are certified
// (54 % with 2 tol, 38 % with 16 x the Gershgorin-bound tolerance); at N <= 100 nearly all).  Sturm counts do the rest: the binade of
// |M|_2 from counts at the two powers of two under its Gershgorin bound (which exceeds it by at most 3x), then the counts at -4 tol and 4 tol.
// H: 192 doubles of scratch.  One wavefront; ~6 k instructions against ~55 k for the eigen-decomposition it makes unnecessary.
template <int n>
__device__ inline bool pi_spectrum_clears_tolerance(const double* M, double* V, double* H) {
    constexpr int ld = n + 1;
    const int lane = lane_id();
    for (int e = lane; e < n * ld; e += WAVE) V[e] = M[e];
    wave_sync();
    double dreg, ereg;
    wave_tridiag_burst(to_lds(V), ld, n, dreg, ereg);
    const lds_ptr dS = to_lds(H), e2S = to_lds(H) + 64, eS = to_lds(H) + 128;
    if (lane < n) { dS[lane] = dreg; eS[lane] = ereg; e2S[lane] = ereg * ereg; }
    wave_sync();
    const int j = (lane < n) ? lane : 0;
    const double rad = fabs(eS[j]) + ((j > 0) ? fabs(eS[j - 1]) : 0.0);
    const double bound = wave_max((lane < n) ? fabs(dS[j]) + rad : 0.0);       // |M|_2 <= bound <= 3 |M|_2
    if (!(bound < 1e300) || !(bound > 1e-300)) return false;                    // (wave-uniform)
    const double p0 = trid_pow2_floor(bound);                                   // 2^k0 <= bound: |M|_2 lies in binade k0, k0 - 1 or k0 - 2
    const double pw = (lane < 2) ? p0 : 0.5 * p0;
    const int cb = trid_count(dS, e2S, n, (lane & 1) ? -pw : pw);              // lanes 0 / 2: eigenvalues below +2^k; lanes 1 / 3: below -2^k
    const bool any0 = wave_bcast_i(cb, 0) < n || wave_bcast_i(cb, 1) > 0, any1 = wave_bcast_i(cb, 2) < n || wave_bcast_i(cb, 3) > 0;
    const double tol = (double)n * eps_of(any0 ? p0 : (any1 ? 0.5 * p0 : 0.25 * p0));
    const double guard = 4.0 * tol;
    const int c = trid_count(dS, e2S, n, (lane & 1) ? guard : -guard);         // eigenvalues below +guard (odd lanes) / below -guard (even lanes)
    const int c_lo = wave_bcast_i(c, 0), c_hi = wave_bcast_i(c, 1);
    wave_sync();
    return c_lo == c_hi;
}

__host__ __device__ inline int pi_wg_lds_doubles(int E, int C, int N) { return pi_lds_doubles(E, C, N, true) + pi_sn(E) * N + 16; }

// pinv(B B' + 1e-12 I) of a 5 x 5 block with TWO near-null directions, at the accuracy of the formula (the two-direction form of
// pinv_block_deflated, gh_kernel.h).  The near-null SUBSPACE is well determined (gap to the third eigenvalue ~ O(1)); inside it the two
// eigenvalues mu_k + 1e-12 (1e-12 .. 1e-9) are separated by less than the rounding of an fp64 B B', so their eigenvectors are taken from
// the 2 x 2 problem G'G, G = B'[u1 u2] (6 x 2), whose entries are squared inconsistencies formed WITHOUT cancellation:
//   u1, u2   orthonormal basis of the subspace (smallest eigenvector of W, then of W + tr(W) u1 u1')
//   G'G = R diag(mu1, mu2) R'   (one Jacobi rotation),  n_k = [u1 u2] r_k
//   K  = (W + n1 n1' + n2 n2')^-1                     Cholesky of a well conditioned matrix
//   pinv = K - sum_k n_k n_k' / (1 + mu_k + 1e-12)  (REGULAR part, returned in Wp)  + sum_k cs_k n_k n_k',  cs_k = 1 / (mu_k + 1e-12) or 0 if truncated
// false when the block does not have that structure (third-smallest eigenvalue not far above, failed factorisation, no convergence).
template <int E>
__device__ __forceinline__ bool pinv_block_deflated2(const double (&B)[E][6], const double (&W)[E][E], const double tolW, double* Wp,
                                                     double (&n1)[E], double (&n2)[E], double* cs1, double* cs2) {
    double u1[E], u2[E], W2[E][E], tr = 0.0;
    // (eight inverse iterations each, converged or not: inside the near-null subspace the iterate may keep drifting -- the two eigenvalues can
    // agree to many digits -- but its component outside decays by (mu / lambda_3)^8, and the structure test below bounds that ratio by 1e-3)
    spd_min_eigvec<E>(W, u1, 8);
#pragma unroll
    for (int a = 0; a < E; ++a) tr += W[a][a];
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int c = 0; c < E; ++c) W2[a][c] = W[a][c] + tr * u1[a] * u1[c];
    spd_min_eigvec<E>(W2, u2, 8);
    double d = 0.0, nn = 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a) d += u1[a] * u2[a];
#pragma unroll
    for (int a = 0; a < E; ++a) { u2[a] -= d * u1[a]; nn += u2[a] * u2[a]; }
    const double rn = rsqrt(nn);
#pragma unroll
    for (int a = 0; a < E; ++a) u2[a] *= rn;
    double m11 = 0.0, m12 = 0.0, m22 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double g1 = 0.0, g2 = 0.0;
#pragma unroll
        for (int a = 0; a < E; ++a) { g1 += B[a][k] * u1[a]; g2 += B[a][k] * u2[a]; }
        m11 += g1 * g1; m12 += g1 * g2; m22 += g2 * g2;
    }
    double cr = 1.0, sr = 0.0, mu1 = m11, mu2 = m22;
    if (m12 != 0.0) {
        const double tau = (m22 - m11) / (2.0 * m12);
        const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        cr = rsqrt(1.0 + t * t); sr = t * cr;
        mu1 = m11 - t * m12; mu2 = m22 + t * m12;
    }
    mu1 = (mu1 > 0.0) ? mu1 : 0.0; mu2 = (mu2 > 0.0) ? mu2 : 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a) { n1[a] = cr * u1[a] - sr * u2[a]; n2[a] = sr * u1[a] + cr * u2[a]; }
    double Wn[E][E];
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int c = 0; c < E; ++c) Wn[a][c] = W[a][c] + n1[a] * n1[c] + n2[a] * n2[c];
    if (!(nn > 1e-8) || !spd_inverse_packed<E>(Wn, Wp)) return false;
    const double l1 = mu1 + 1e-12, l2 = mu2 + 1e-12, k1 = 1.0 / (1.0 + l1), k2 = 1.0 / (1.0 + l2);
    double fro2 = 0.0;                                                       // |regular part|_F^2 = sum over the ordinary directions of 1 / (lambda_k + 1e-12)^2
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            const double v = Wp[a * (a + 1) / 2 + c] - n1[a] * n1[c] * k1 - n2[a] * n2[c] * k2;
            Wp[a * (a + 1) / 2 + c] = v;
            fro2 += (a == c) ? v * v : 2.0 * v * v;
        }
    const double mum = (mu1 > mu2) ? mu1 : mu2;
    const double l3min = 1e-5 * tr + 1e3 * mum;                              // third-smallest eigenvalue >= 1 / |.|_F: far above both small ones
    if (!(fro2 * l3min * l3min < 1.0)) return false;
    *cs1 = (l1 > tolW) ? 1.0 / l1 : 0.0;
    *cs2 = (l2 > tolW) ? 1.0 / l2 : 0.0;
    return true;
}

// pinv's tolerance E N eps(max_i lambda_max(W_i)) for the block-diagonal weight matrix (Gauss_Helmert.m:57).  Only the binade of the
// maximum enters: the eigenvalue pass is skipped when cheap upper / lower bounds agree on it.  Block-wide (contains barriers).
// pi_tolerance_from_bounds: umax / lmax are the block-wide bounds (taken along with the finite check since round 5).
template <class Model, int WV>
__device__ inline double pi_tolerance_from_bounds(const PiWork& g, const double (&pi)[27], const int N, const int tid, double* red, const double umax, const double lmax) {
    constexpr int THREADS = WV * WAVE;
    constexpr int E = Model::E;
    double smax = umax;
    if (eps_of(lmax) != eps_of(umax)) {
        smax = 0.0;
        for (int i = tid; i < N; i += THREADS) {
            double o[6], W[E][E], V[E][E];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            PiPoint<E> pt;
            pi_eval<Model, true>(pi, o, pt);
            pi_block_W<E>(pt.B, W);
            jacobi_small<E, false>(W, V);
#pragma unroll
            for (int a = 0; a < E; ++a) smax = (fabs(W[a][a]) > smax) ? fabs(W[a][a]) : smax;
        }
        smax = block_max_w<WV>(smax, red);
    }
    return (double)E * (double)N * eps_of(smax);
}

template <class Model, int WV>
__device__ inline int gauss_helmert_pi_block(PoseLds* w, PiWork& g, double* red, int own, const double* pts, int N, int* st, bool exact_pinv) {
    constexpr int THREADS = WV * WAVE;
    constexpr int E = Model::E, C = Model::C, u = 27, n = u + C, ld = n + 1, PP = pi_pp(E), NW = E * (E + 1) / 2;
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    const bool owner = wave == own;
    double objFunc = 0.0;                                                    // v0' v0, v0 = x0 - x   (:45-46)
    for (int i = tid; i < N; i += THREADS) {
        const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = g.xi[6 * i + k] - x.v[k]; objFunc += d * d; }
    }
    objFunc = block_sum_w<WV>(objFunc, red);
    int it = 0;
#pragma unroll 1
    for (it = 1; it <= GH_IT_MAX; ++it) {
        double pi[27];
        load_uniform27(g.p, pi);
        // ---- W = B B' (:52), its finite check (:53-55) and pinv(W + 1e-12 I) (:57).
        // Round 5: the finite check, max_i |W_i|_F and the cheap bounds on max_i lambda_max(W_i) behind pinv's tolerance are ONE pass and one
        // barrier pair (the bounds were a pass of their own: pi_eval + pi_block_W per correspondence again, two more reductions).  Same blocks,
        // same arithmetic.  (A speculative weight pass without the tolerance was measured first and lost: on normalised image data at
        // N = 200 the tolerance E N eps(lambda_max) exceeds the 1e-12 shift, so the pass was repeated on every problem -- profiles/r5_ab_libs.txt.)
        double f2max = 0.0, umax = 0.0, lmax = 0.0;                          // max_i |W_i|_F^2; bounds on max_i lambda_max(W_i)
        {
            bool finite = true;
            for (int i = tid; i < N; i += THREADS) {
                double o[6], W[E][E];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                PiPoint<E> pt;
                pi_eval<Model, true>(pi, o, pt);
                pi_block_W<E>(pt.B, W);
                double chk = 0.0, fro2 = 0.0;
#pragma unroll
                for (int a = 0; a < E; ++a) {
#pragma unroll
                    for (int b = 0; b < E; ++b) { chk += W[a][b]; fro2 += W[a][b] * W[a][b]; }
                }
                finite = finite && (fabs(chk) <= 1.79e308);
                f2max = (fro2 > f2max) ? fro2 : f2max;
                double up, lo;
                psd_lambda_max_bounds(W, up, lo);
                umax = (up > umax) ? up : umax;
                lmax = (lo > lmax) ? lo : lmax;
            }
            bool bad_entry = !finite;
            block_max3_any_w<WV>(f2max, umax, lmax, bad_entry, red);
            if (bad_entry || !(f2max <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :53-55
        }
        // ---- Pi (4 x 4 blocks, non-singular KKT): weights at the accuracy of the formulas, as in gh_wg_kernel.h -- the block
        //      pseudo-inverse deflated by its one small eigenvalue (pinv_block_deflated<true>: regular part only in pp), the strong
        //      direction kept apart as (n, cs, n'w) and its contributions cs a a', cs a n'w formed from a = A_i' n FIRST (a is the
        //      inconsistency of the correspondence: tiny, while cs ~ 1e12).  Sums: 27 * 28 / 2 + 27 = 405 (PiCol: two families), on the matrix
        //      core into per-wavefront slots (V, H: both dead here), combined into S = V[0 .. 405).
        bool factored = false;
        constexpr int SN = pi_sn(E);
        if (!exact_pinv && g.sn != nullptr) {
            {
                const double tolW = pi_tolerance_from_bounds<Model, WV>(g, pi, N, tid, red, umax, lmax);
                constexpr int SLOT = 406;
                double* slot = (wave < WV - 1) ? g.V + wave * SLOT : g.H;
                for (int e = lane; e < SLOT; e += WAVE) slot[e] = 0.0;
                bool bad = false;
                // the strong-direction sums on the matrix core (gh_kernel.h::StrongGram): tiles in registers across the trips; the wavefront's
                // transposition scratch lives in M (dead until the KKT matrix is assembled).  PiCol's two families keep their own accumulators
                // and meet in the slot, as their butterfly sums did.
                typedef StrongGram<27, 16> Gram;
                static_assert(WV * Gram::SCRATCH <= (27 + C) * (28 + C), "transposition scratch of the strong-direction Gram must fit M");
                Gram gram, gram2;
                gram.clear();
                if constexpr (E == 5) gram2.clear();
                double* gscratch = g.M + wave * Gram::SCRATCH;
#pragma unroll 1
                for (int base = 0; base < N; base += THREADS) {        // block-uniform trip count (the butterflies need whole wavefronts)
                    const int i = base + tid;
                    double bv[27], tv = 0.0, bv2[27], tv2 = 0.0;
#pragma unroll
                    for (int k = 0; k < 27; ++k) { bv[k] = 0.0; bv2[k] = 0.0; }
                    if (i < N) {
                        double o[6], W[E][E], Wp[NW], nn[E], nm[E], cs = 0.0, cs2 = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                        PiPoint<E> pt;
                        pi_eval<Model, true>(pi, o, pt);
                        pi_block_W<E>(pt.B, W);
                        bool ok;
                        if constexpr (E == 4) {
                            ok = pinv_block_deflated<true>(pt.B, W, tolW, Wp, nn, &cs);
#pragma unroll
                            for (int a = 0; a < E; ++a) nm[a] = 0.0;
                        } else {
                            ok = pinv_block_deflated2<E>(pt.B, W, tolW, Wp, nn, nm, &cs, &cs2);
                        }
                        bad = !ok || bad;
#pragma unroll
                        for (int a = 0; a < E; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
                        pi_store_point<E>(g, w, pts, i, o, pt, Wp);
                        const Pt6 x = premap(load_pt(pts, i), w->nrm);
                        double nw = 0.0, nw2 = 0.0;                          // n'w,  w = -f - B (x - xi)
#pragma unroll
                        for (int a = 0; a < E; ++a) {
                            double wa = -pt.f[a];
#pragma unroll
                            for (int k = 0; k < 6; ++k) wa -= pt.B[a][k] * (x.v[k] - o[k]);
                            nw += nn[a] * wa;
                            nw2 += nm[a] * wa;
                        }
                        double* sn = g.sn + SN * (long)i;
                        if constexpr (E == 4) {
                            sn[0] = nn[0]; sn[1] = nn[1]; sn[2] = nn[2]; sn[3] = nn[3]; sn[4] = ok ? cs : 0.0; sn[5] = nw;
                        } else {
#pragma unroll
                            for (int a = 0; a < E; ++a) { sn[a] = nn[a]; sn[E + a] = nm[a]; }
                            sn[2 * E] = ok ? cs : 0.0; sn[2 * E + 1] = ok ? cs2 : 0.0; sn[2 * E + 2] = nw; sn[2 * E + 3] = nw2;
                        }
                        if (ok) {
                            Model::a_quirk(pt.c);
                            const double sc = sqrt(cs), sc2 = sqrt(cs2);
#pragma unroll
                            for (int b = 0; b < 9; ++b) {                    // a[3 b + k] = (sum_r n_r c[r][b]) p_view(b)[k]
                                double an = 0.0, an2 = 0.0;
#pragma unroll
                                for (int r = 0; r < E; ++r)
                                    if (Model::nz(r, b)) { an += nn[r] * pt.c[r][b]; an2 += nm[r] * pt.c[r][b]; }
                                an *= sc; an2 *= sc2;
#pragma unroll
                                for (int k = 0; k < 3; ++k) { bv[3 * b + k] = an * hom_at(o, b / 3, k); bv2[3 * b + k] = an2 * hom_at(o, b / 3, k); }
                            }
                            tv = sc * nw; tv2 = sc2 * nw2;
                        }
                    }
                    gram.add(bv, tv, gscratch);
                    if constexpr (E == 5) gram2.add(bv2, tv2, gscratch);          // kept apart from the first: one accumulator for both families (more k-steps of the same tiles) moved a fixture scene over the 1e-9 gate, as adding the products before the butterflies did in round 3 (7e-10 -> 1.7e-9)
                }
                wave_sync();
                if constexpr (E == 5) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) { gram.t00[v] += gram2.t00[v]; gram.t10[v] += gram2.t10[v]; gram.t11[v] += gram2.t11[v]; }
                }
                gram.store(slot);
                __syncthreads();
                for (int e = tid; e < 405; e += THREADS) g.V[e] = (WV == 4) ? (g.V[e] + g.V[SLOT + e]) + (g.V[2 * SLOT + e] + g.H[e]) : g.V[e] + g.H[e];
                factored = !block_any_w<WV>(bad, red);                             // a block without the structure: the unfactored paths below for all
            }
        }
        bool fast = !exact_pinv && (double)E * (double)N * eps_of(sqrt(f2max)) < 0.9e-12;
        if (factored) {
        } else if (fast) {
            bool bad = false;
            for (int i = tid; i < N; i += THREADS) {
                double o[6], W[E][E], Wp[NW];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                PiPoint<E> pt;
                pi_eval<Model, true>(pi, o, pt);
                pi_block_W<E>(pt.B, W);
                bad = !spd_inverse_packed<E>(W, Wp) || bad;
#pragma unroll
                for (int a = 0; a < E; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
                pi_store_point<E>(g, w, pts, i, o, pt, Wp);
            }
            if (block_any_w<WV>(bad, red)) fast = false;
        }
        if (!fast && !factored) {
            const double tolW = pi_tolerance_from_bounds<Model, WV>(g, pi, N, tid, red, umax, lmax);
            for (int i = tid; i < N; i += THREADS) {
                double o[6], W[E][E], V[E][E];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                PiPoint<E> pt;
                pi_eval<Model, true>(pi, o, pt);
                pi_block_W<E>(pt.B, W);
                double Wp[NW];
                // E = 4 (Pi): one truncated direction is the generic case, no eigen-decomposition then; measured slower for PiCol's 5 x 5 blocks
                if (E == 4 && pinv_one_null_packed<E>(W, tolW, Wp)) {
#pragma unroll
                    for (int a = 0; a < E; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
                } else {
                    jacobi_small<E, true>(W, V);
                    double inv[E];
#pragma unroll
                    for (int a = 0; a < E; ++a) inv[a] = (W[a][a] > tolW) ? 1.0 / W[a][a] : 0.0;
#pragma unroll
                    for (int a = 0; a < E; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) {
                            double s = (a == b) ? 1e-12 : 0.0;
#pragma unroll
                            for (int k = 0; k < E; ++k) s += V[a][k] * inv[k] * V[b][k];
                            Wp[a * (a + 1) / 2 + b] = s;
                        }
                }
                pi_store_point<E>(g, w, pts, i, o, pt, Wp);
            }
        }
        // ---- A'WA and A'Ww: the sixteen sweeps are dealt to the four wavefronts (wave w: sweeps w, w + 4, w + 8, w + 12), each over ALL
        //      correspondences and straight into H -- a quarter of the reductions of a per-wavefront-partial layout, no combine step ----
        __syncthreads();                                                     // every correspondence's xi, W+ are in place
        if constexpr (WV == 4) {
            if (wave == 0) pi_sweeps_strided<Model, 0>(g, pi, N, lane, WAVE, g.H);
            else if (wave == 1) pi_sweeps_strided<Model, 1>(g, pi, N, lane, WAVE, g.H);
            else if (wave == 2) pi_sweeps_strided<Model, 2>(g, pi, N, lane, WAVE, g.H);
            else pi_sweeps_strided<Model, 3>(g, pi, N, lane, WAVE, g.H);
        } else {                                                             // two wavefronts: eight sweeps each
            if (wave == 0) { pi_sweeps_strided<Model, 0>(g, pi, N, lane, WAVE, g.H); pi_sweeps_strided<Model, 2>(g, pi, N, lane, WAVE, g.H); }
            else { pi_sweeps_strided<Model, 1>(g, pi, N, lane, WAVE, g.H); pi_sweeps_strided<Model, 3>(g, pi, N, lane, WAVE, g.H); }
        }
        __syncthreads();
        for (int e = tid; e < n * ld; e += THREADS) g.M[e] = 0.0;
        __syncthreads();
        for (int e = tid; e < 729 + 27; e += THREADS) {
            if (e < 729) {
                const int r = e / 27, cc = e % 27;
                const int b = r / 3, k = r % 3, bp = cc / 3, kk = cc % 3;
                const double v = (b >= bp) ? g.H[9 * (b * (b + 1) / 2 + bp) + 3 * k + kk] : g.H[9 * (bp * (bp + 1) / 2 + b) + 3 * kk + k];
                const double sv = factored ? g.V[(r >= cc) ? tri_index(r, cc) : tri_index(cc, r)] : 0.0;
                g.M[r * ld + cc] = (v + sv) + ((r == cc) ? 1e-12 : 0.0);
            } else {
                g.M[(e - 729) * ld + n] = g.H[405 + e - 729] + (factored ? g.V[378 + e - 729] : 0.0);
            }
        }
        if (tid < C) {                                                       // constraints g, C   (KKT borders)
            const int b = Model::cb(tid, 0), bp = Model::cb(tid, 1), row = u + tid;
            double gv = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double pb = g.p[3 * b + k], pbp = g.p[3 * bp + k];
                gv += pb * pbp;
                if (b == bp) { g.M[row * ld + 3 * b + k] = 2.0 * pb; g.M[(3 * b + k) * ld + row] = 2.0 * pb; }
                else {
                    g.M[row * ld + 3 * b + k] = pbp; g.M[(3 * b + k) * ld + row] = pbp;
                    g.M[row * ld + 3 * bp + k] = pb; g.M[(3 * bp + k) * ld + row] = pb;
                }
            }
            g.M[row * ld + n] = -(gv - ((b == bp) ? 1.0 : 0.0));
            g.M[row * ld + row] = 1e-12;
        }
        __syncthreads();
        double chkM = 0.0;
        for (int e = tid; e < n * ld; e += THREADS) chkM += g.M[e];
        if (!(fabs(block_sum_w<WV>(chkM, red)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        // aux = pinv(M + 1e-12 I) b   (:67)
        bool need_pinv;
        if (!Model::PINV_KKT) {
            if (owner) { const bool ok = wave_solve_gj<n>(g.M, g.dt); if (lane == 0) red[8] = ok ? 1.0 : 0.0; }
            __syncthreads();
            need_pinv = red[8] == 0.0;
        } else {
            // PiCol: eleven constraints, redundant on collinear centres -- there pinv truncates and only the eigen-decomposition reproduces it.  On every other
            // scene the 38 x 38 matrix has no eigenvalue anywhere near pinv's tolerance, pinv(M) IS inv(M), and the eigen-decomposition was 2.9 of the method's
            // 5.4 ms (profiles/r5_ab_picol_without_pinv.txt).  So: a CERTIFICATE that nothing would be truncated (pi_spectrum_clears_tolerance, on a copy of M), then the plain
            // solve (M stays intact: the elimination runs in registers); without the certificate the eigen-decomposition runs on the untouched M as before.
            if (owner) {
                bool settled = false;                                        // the certificate first: half of the N = 200 scenes do not get it, and the elimination would be wasted on them
                if (wave_uniform_i(pi_spectrum_clears_tolerance<n>(g.M, g.V, g.H) ? 1 : 0)) settled = wave_solve_gj<n>(g.M, g.dt, 1e-15);
                if (lane == 0) red[8] = settled ? 1.0 : 0.0;
            }
            __syncthreads();
            need_pinv = red[8] == 0.0;
        }
        if (need_pinv) block_pinv_solve_sym<n>(g.M, g.V, g.dt, g.H, own);     // eigenvectors -> V, scratch -> H (both dead by now)
        __syncthreads();
        double dt[27];
        load_uniform27(g.dt, dt);
        // ---- v = -B' W+ (A dt - w)   (:69) ----
        double obj = 0.0, diff = 0.0;
        for (int i = tid; i < N; i += THREADS) {
            double o[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            PiPoint<E> pt;
            pi_eval<Model, true>(pi, o, pt);
            Model::a_quirk(pt.c);
            double q[9];
#pragma unroll
            for (int b = 0; b < 9; ++b) q[b] = dt[3 * b] * o[2 * (b / 3)] + dt[3 * b + 1] * o[2 * (b / 3) + 1] + dt[3 * b + 2];
            double Ad[E];
#pragma unroll
            for (int r = 0; r < E; ++r) {
                double s = 0.0;
#pragma unroll
                for (int b = 0; b < 9; ++b)
                    if (Model::nz(r, b)) s += pt.c[r][b] * q[b];
                Ad[r] = s;
            }
            double* pw = g.pp + (long)PP * i;
            double rr[E];
#pragma unroll
            for (int a = 0; a < E; ++a) {
                double s = -pw[NW + a];
#pragma unroll
                for (int b = 0; b < E; ++b) s += sym_at<E>(pw, a, b) * Ad[b];
                rr[a] = s;
            }
            if (factored) {                                                  // + cs n (n'(A dt) - n'w): the strong direction(s) of W+
                const double* sn = g.sn + SN * (long)i;
                if constexpr (E == 4) {
                    const double st = sn[4] * ((sn[0] * Ad[0] + sn[1] * Ad[1] + sn[2] * Ad[2] + sn[3] * Ad[3]) - sn[5]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) rr[a] += sn[a] * st;
                } else {
                    double d1 = -sn[2 * E + 2], d2 = -sn[2 * E + 3];
#pragma unroll
                    for (int a = 0; a < E; ++a) { d1 += sn[a] * Ad[a]; d2 += sn[E + a] * Ad[a]; }
                    d1 *= sn[2 * E]; d2 *= sn[2 * E + 1];
#pragma unroll
                    for (int a = 0; a < E; ++a) rr[a] += sn[a] * d1 + sn[E + a] * d2;
                }
            }
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
            double vv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double s = 0.0;
#pragma unroll
                for (int a = 0; a < E; ++a) s -= pt.B[a][k] * rr[a];
                vv[k] = s;
                obj += s * s;
                const double d = o[k] - x.v[k] - s;
                diff += d * d;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) pw[k] = vv[k];
        }
        obj = block_sum_w<WV>(obj, red);
        diff = block_sum_w<WV>(diff, red);
        double ndt2 = 0.0;
#pragma unroll
        for (int k = 0; k < 27; ++k) ndt2 += dt[k] * dt[k];
        if (sqrt(ndt2) < GH_TOL && sqrt(diff) < GH_TOL) break;               // :71-73
        if (obj > objFunc) break;                                            // :75-76
        objFunc = obj;
        for (int i = tid; i < N; i += THREADS) {                       // xi = x + v; ti = ti + dt   (:80)
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) g.xi[6 * i + k] = x.v[k] + g.pp[(long)PP * i + k];
        }
        if (tid < u) g.p[tid] += g.dt[tid];
        __syncthreads();
    }
    __syncthreads();
    return (it > GH_IT_MAX) ? GH_IT_MAX : it;
}

// TWO wavefronts per workgroup (round 4; four before), four workgroups per CU, 256 registers per thread.  The wave-serial steps (the 36 x 36
// pivoted elimination of Pi, the truncated pseudo-inverse of PiCol's 38 x 38 KKT matrix) are what these kernels wait for, and what overlaps them is
// the number of WORKGROUPS per CU: with four wavefronts each that was two at 256 registers (Pi) or four at 128 registers with 767 of them spilled
// (PiCol: 1.3 KB of scratch per lane, 105x the algorithmic bytes).  Two wavefronts per workgroup give both four workgroups and 256 registers:
// Pi 3.65 -> 3.19 ms, PiCol 6.19 -> 5.46 ms per 10 000 x 200.  The per-correspondence state goes to global slices (plan_spill) as before.
template <class Model> struct pi_wg_waves { static constexpr int value = 2; };
template <class Model>
__global__ void __launch_bounds__(pi_wg_waves<Model>::value * WAVE, 2) k_pi_block(const GhWgArgs a) {   // (second argument: wavefronts per SIMD)
    constexpr int WV = pi_wg_waves<Model>::value;
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    double* ghbase = smem + base;
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        if (a.status[b] != ST_OK) continue;
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        PiWork g = pi_carve(ghbase, Model::E, Model::C, a.spill ? 0 : N, true);
        g.sn = g.pp + (long)pi_pp(Model::E) * (a.spill ? 0 : N);
        double* red = g.sn + (long)pi_sn(Model::E) * (a.spill ? 0 : N);
        if (a.spill) { g.xi = a.spill + blockIdx.x * a.spill_stride; g.pp = g.xi + 6 * (long)N; g.sn = g.pp + (long)pi_pp(Model::E) * N; }
        const int own = pick_serial_wave_w<WV>(red);
        const double* r = a.rec + b * GH_REC_DOUBLES;
        if (tid < 27) w->t[tid] = r[tid];
        if (tid < 18) w->pa[tid] = r[27 + tid];
        if (tid < 6) w->epi[tid] = r[45 + tid];
        if (tid < 9) w->nrm[tid] = r[51 + tid];
        __syncthreads();
        if (wave == own && lane == 0) red[9] = (double)Model::init(w, g.p);  // Pi matrices from the linear cameras; cameras for x_est
        __syncthreads();
        const int ist = (int)red[9];
        if (ist != ST_OK) {                                                  // 'The minimal param could not be found'
            if (tid == 0) a.status[b] = ist;
            continue;
        }
        gh_block_reproject<WV>(w, pts, N, g.xi);                             // x_est   (PiPoseEstimation.m:80-83)
        __syncthreads();
        int gst = ST_OK;
        const int iters = gauss_helmert_pi_block<Model, WV>(w, g, red, own, pts, N, &gst, (a.flags & FLAG_GH_EXACT) != 0);
        if (wave == own) {
            if (lane == 0) Model::cameras(g.p, w);                           // :94-100
            wave_sync();
            tft_from_cameras(w, w->t);                                       // T = TFT_from_P(P1,P2,P3)
            wave_sync();
            if (lane < 27) a.topt[b * 27 + lane] = w->t[lane];
            if (lane == 0) {
                if (a.iter) a.iter[b] = iters;
                if (gst != ST_OK) a.status[b] = -gst;
            }
        }
    }
}

}  // namespace tff
