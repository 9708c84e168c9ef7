// Faugeras-Papadopoulo's parameterisation (FaugPapaTFTPoseEstimation.m:48-153, all 27 tensor entries + 12 algebraic constraints) on
// Gauss_Helmert.m:38-83 at the accuracy of the reference's FORMULAS, one workgroup of four wavefronts per triplet.
//
// Why this method needs its own iteration.  pinv(B_i B_i' + 1e-12 I) (Gauss_Helmert.m:57) gives every correspondence one direction
// n_i of weight cs_i = 1 / (mu_i + 1e-12) ~ 1e9 .. 1e12.  With all 27 entries as parameters the vectors a_i = Ap_i' n_i are O(1) -- they
// span the normal space of the trifocal variety at T -- so A'WA = R + sum_i cs_i a_i a_i' carries a rank-8/9 part of size 1e13 on top of
// the regular part R of size N, in a subspace that is not aligned with any coordinate axis: every entry of the formed matrix is ~1e13
// and the regular part is known to 1e-3 absolute only.  That is the 1e-6 .. 1e-4 by which ANY fp64 evaluation of the formed 39 x 39
// KKT system (MATLAB's own included) misses the iteration the formulas define (tests/golden/gh_mp_faugpapa.npz, 50 digits).
// Here the tangential part of the strong terms never meets their normal part in one fp64 number:
//   1. weights in the deflated, factored form of gh_kernel.h (pinv_block_deflated<true>): regular part K_i, strong direction (n_i, cs_i);
//   2. Kronecker-structured sums, one wavefront sweeping all correspondences per 30 sums: R = sum Ap_i' K_i Ap_i and A'Ww's regular
//      part (297), the strong Gram matrix Hs = sum cs_i a_i a_i' and sum cs_i (n_i'w_i) a_i (270 + 27);
//   3. an orthogonal Q whose first ns columns span the dominant columns of Hs (diagonally pivoted Cholesky + Householder QR, one
//      wavefront): in the basis Q the strong subspace is axis-aligned up to ~1e-6;
//   4. the TANGENTIAL block of the strong Gram matrix again, from the rotated factors g_i = sqrt(cs_i) Q(:, 8:27)' a_i (190 + 19 sums,
//      one correspondence per thread): those components are ~1e-6 |a_i| and come out of the dot products with ~1e-10 relative error,
//      so the block is accurate at its OWN scale (size N).  The strong and cross blocks Q(:,1:8)' Hs Q are taken from the formed Hs:
//      their rounding (1e-16 of 1e13) is 1e-10 of the cross block's size and enters the Schur complement at 1e-10 relative;
//   5. M' = [Q'RQ + Q'HsQ + 1e-12 I, (CQ)'; CQ, 1e-12 I]: Cholesky elimination of the leading ns x ns block (exact algebra; the
//      truncated pseudo-inverse of the whole differs from block elimination + truncated pseudo-inverse of the Schur complement by
//      O((cross / strong)^2) ~ 1e-12), pinv's tolerance 39 eps(|M|_2) from the strong block's largest eigenvalue;
//   6. truncated pseudo-inverse of the 30 / 31-dimensional Schur complement by wave_pinv_solve_trid (wave_trid.h), back-substitution,
//      dt = Q z;  v = -B' W+ (A dt - w) with the strong term cs n n'(A dt - w) evaluated from the differences.
// tools/proto_faugpapa_factored.py is the numpy twin: 48 fixture scenes within 4e-11 of the 50-digit iteration, equal iteration counts.
//
// Per-correspondence state (estimate xi, K_i, n_i, cs_i, n_i'w_i: 22 doubles) lives in the global slices of launch_wg whenever that lets
// more workgroups share a CU (the default, plan_spill: three per CU at N = 200) or in LDS (35 KB at N = 200, two workgroups per CU;
// TFF_OPT_SPILL = 1 keeps it there whenever it fits).  A first version kept it in registers, one correspondence per thread, for
// four workgroups per CU: with 32-wide butterflies on top the register allocator spilled it around every sum (15 GB of scratch traffic
// per 10 k x 200 launch, 11 ms).  Triplets this kernel cannot take (a weight block without the one-small-eigenvalue structure) are
// handed to k_gh_block<FaugPapaModel> through the status array (ST_RETRY).
#pragma once
#include "gh_wg_kernel.h"
#include "wave_trid.h"

namespace tff {

// Two wavefronts per workgroup, four workgroups per CU (what the 39 KB of matrices allow), 256 registers per thread (round 4; before: four
// wavefronts, three workgroups, 168 registers).  The owner-only steps are 68 % of an iteration's wall time: the number of workgroups per CU is
// what overlaps them (pi_wg_kernel.h, gh_wg_kernel.h::gh_wg_waves).
constexpr int FP_WAVES = 2;
constexpr int FP_THREADS = FP_WAVES * WAVE;
constexpr int FP_WG_PER_CU = 4;
constexpr bool FP_TRID_IN_REGISTERS = true;    // the pseudo-inverse's reduction in registers (two workgroups per CU leave 256 per thread)
constexpr int FP_NS_MAX = 12;                // strong directions eliminated ahead of the pseudo-inverse (the normal space has dimension 9)
constexpr int FP_C0 = 8;                     // columns FP_C0 .. 26 of Q get the rotated (accurate) strong sums when ns >= FP_C0
constexpr int FP_XI = 6, FP_PP = 16;         // per-correspondence records: xi | W+ regular part (10), n (4), cs, n'w

struct FpLds {
    double p[28];          // parameters = tensor entries, t(j + 3k + 9i) = T(j,k,i)
    double nrm[12];        // Normalize2Ddata of the three views (9)
    double cam[3][12];     // P1, P2, P3 of the linear solution (row-major 3 x 4)
    double gneg[12];       // -g
    double rvec[28];       // regular part of A'Ww
    double rsvec[28];      // strong part of A'Ww: sum cs (n'w) a
    double red[16];
    double dt[40];         // z (basis Q), then dt at [0, 27)
    double sm[TRID_SMALL_DOUBLES];
    double flag[8];        // [0] ns  [1] |M|_2  [2] pseudo-inverse failure
    // dead during the pseudo-inverse: A1 | A2 | fin | Cm are its TRID_WORK_DOUBLES (2112 <= 2198) or the 39 x 39 eigenvectors of the fall-back
    double A1[729];        // R, then Q'RQ
    double A2[729];        // Hs, then R Q
    double fin[432];       // reflectors of the basis (12 x 27), then the rotated sums (209 / 405) | Q(:,1:8)' Hs (8 x 27, at 216)
    double Cm[324];        // C = dg/dT, 12 x 27
    double Q[729];
    double Mx[1664];       // the two families of structured sums (2 x 300) | per-wavefront slots of the rotated sums | augmented M' 39 x 40
};
constexpr int FP_LDS_DOUBLES = (int)(sizeof(FpLds) / sizeof(double));
static_assert(729 + 729 + 432 + 324 >= TRID_WORK_DOUBLES && 729 + 729 + 432 >= 39 * 39, "pseudo-inverse workspace");
__host__ __device__ inline size_t fp_lds_bytes(int N) { return sizeof(FpLds) + (size_t)(FP_XI + FP_PP) * (size_t)N * sizeof(double); }

__device__ __forceinline__ double block_sum2(double v, double* w2, double* red) {   // two sums, one barrier pair: returns sum(v), *w2 <- sum(*w2)
    v = wave_sum(v);
    const double u = wave_sum(*w2);
    if (lane_id() == 0) { red[wave_in_block()] = v; red[4 + wave_in_block()] = u; }
    __syncthreads();
    const double r = (FP_WAVES == 4) ? (red[0] + red[1]) + (red[2] + red[3]) : red[0] + red[1];
    *w2 = (FP_WAVES == 4) ? (red[4] + red[5]) + (red[6] + red[7]) : red[4] + red[5];
    __syncthreads();
    return r;
}

// s = sqrt(cs) K n (9): a_i = h1 (x) (K n_i)
__device__ __forceinline__ void fp_strong_vector(const double (&o)[6], const double (&nn)[4], const double sc, double (&sq)[9]) {
    double kr[4];
#define TFF_SQ(Q) K_row<Q>(o[2], o[3], o[4], o[5], kr); sq[Q] = sc * (kr[0] * nn[0] + kr[1] * nn[1] + kr[2] * nn[2] + kr[3] * nn[3]);
    TFF_SQ(0) TFF_SQ(1) TFF_SQ(2) TFF_SQ(3) TFF_SQ(4) TFF_SQ(5) TFF_SQ(6) TFF_SQ(7) TFF_SQ(8)
#undef TFF_SQ
}
// five entries of the lower triangle of s s' times the six products of h1: 30 of the 270 sums of Hs.
// (Entry indices are template arguments: a loop variable inside tri_row_of() is not folded early enough and would put s[] in scratch memory.)
template <int E>
__device__ __forceinline__ void fp_strong_entry(const double (&s)[9], const double (&hh)[6], double* acc) {
    const double z = s[tri_row_of(E)] * s[tri_col_of(E)];
#pragma unroll
    for (int h = 0; h < 6; ++h) acc[h] += hh[h] * z;
}
template <int CH>
__device__ __forceinline__ void fp_strong_chunk(const double (&s)[9], const double (&hh)[6], double (&acc)[32]) {
    fp_strong_entry<5 * CH + 0>(s, hh, acc + 0);
    fp_strong_entry<5 * CH + 1>(s, hh, acc + 6);
    fp_strong_entry<5 * CH + 2>(s, hh, acc + 12);
    fp_strong_entry<5 * CH + 3>(s, hh, acc + 18);
    fp_strong_entry<5 * CH + 4>(s, hh, acc + 24);
}
// Where the per-correspondence state lives decides the pointer type: LDS pointers are 32-bit and lower to ds_read / ds_write with
// immediate offsets; through generic `double*` every access is a flat load with 64-bit address arithmetic, which the optimiser hoists out
// of the loops by the hundred and then spills.
template <bool IN_LDS> struct FpState { typedef double* ptr; typedef const double* cptr; };
template <> struct FpState<true> { typedef lds_ptr ptr; typedef lds_ptr cptr; };

// One sweep of the calling wavefront over ALL correspondences: 30 (27) sums -> H[30 CH ..] (H[270 ..]).
// Regular family (STRONG = false): chunk CH < 9 of Ghat = sum (h1 h1') (x) (K W+ K'), CH = 9: ghat = sum h1 (x) (K W+ w), w recomputed.
// Strong family: chunk CH < 9 of sum cs (h1 h1') (x) (K n)(K n)', CH = 9: sum cs (n'w) h1 (x) (K n).
template <bool STRONG, int CH, class SP>
__device__ inline void fp_sweep(const FpLds& s, const SP xi, const SP pp, const double* pts, const int N, const double (&T)[27], double* H) {
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        GhPoint pt;
#pragma unroll
        for (int k = 0; k < 6; ++k) pt.o[k] = xi[FP_XI * (long)i + k];
        if constexpr (!STRONG) {
#pragma unroll
            for (int k = 0; k < 10; ++k) pt.Wp[k] = pp[FP_PP * (long)i + k];
            if constexpr (CH == 9) {
                double f[4], B[4][6], wv[4];
                tril_block(T, pt.o, f, B);
                const Pt6 x = premap(load_pt(pts, i), s.nrm);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    double sw = -f[a];
#pragma unroll
                    for (int k = 0; k < 6; ++k) sw -= B[a][k] * (x.v[k] - pt.o[k]);
                    wv[a] = sw;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    pt.ww[a] = wp_at(pt.Wp, a, 0) * wv[0] + wp_at(pt.Wp, a, 1) * wv[1] + wp_at(pt.Wp, a, 2) * wv[2] + wp_at(pt.Wp, a, 3) * wv[3];
                gh_accum_rhs(pt, acc);
            } else {
                const double hh[6] = {pt.o[0] * pt.o[0], pt.o[0] * pt.o[1], pt.o[0], pt.o[1] * pt.o[1], pt.o[1], 1.0};
                gh_accum_chunk<CH>(pt, hh, acc);
            }
        } else {
            const double nn[4] = {pp[FP_PP * (long)i + 10], pp[FP_PP * (long)i + 11], pp[FP_PP * (long)i + 12], pp[FP_PP * (long)i + 13]};
            const double cs = pp[FP_PP * (long)i + 14];
            double sq[9];
            fp_strong_vector(pt.o, nn, sqrt(cs), sq);
            if constexpr (CH == 9) {
                const double so = sqrt(cs) * pp[FP_PP * (long)i + 15];
#pragma unroll
                for (int q = 0; q < 9; ++q) { const double v = sq[q] * so; acc[q] += pt.o[0] * v; acc[9 + q] += pt.o[1] * v; acc[18 + q] += v; }
            } else {
                const double hh[6] = {pt.o[0] * pt.o[0], pt.o[0] * pt.o[1], pt.o[0], pt.o[1] * pt.o[1], pt.o[1], 1.0};
                fp_strong_chunk<CH>(sq, hh, acc);
            }
        }
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0) {
        if (CH < 9) { if (idx < 30) H[30 * CH + idx] = tot; }
        else if (idx < 27) H[270 + idx] = tot;
    }
}

// CLEN components of Q' (h1 (x) sq), columns CSTART .. CSTART + CLEN - 1 of Q.  In fenced chunks: left to itself the compiler merges the LDS
// loads of the whole product at its head, or sinks the arithmetic towards the sums that use it, and spills ~400 registers either way.
template <int CSTART, int CLEN, int OUT0, int U>
__device__ __forceinline__ void fp_rotate_chunk(const double* Q, const double (&o)[6], const double (&sq)[9], double (&bv)[U]) {
    double acc[CLEN];
#pragma unroll
    for (int c = 0; c < CLEN; ++c) acc[c] = 0.0;
#pragma unroll
    for (int i1 = 0; i1 < 3; ++i1)
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const double ar = ((i1 == 0) ? o[0] : ((i1 == 1) ? o[1] : 1.0)) * sq[m];
            const double* qr = Q + (m + 9 * i1) * 27 + CSTART;
#pragma unroll
            for (int c = 0; c < CLEN; ++c) acc[c] += qr[c] * ar;
        }
#pragma unroll
    for (int c = 0; c < CLEN; ++c) { pin_value(acc[c]); bv[OUT0 + c] = acc[c]; }   // computed here, not sunk towards the sums
    wave_sync();                                                             // a memory fence between the chunks
}

// Step 3 on one wavefront: Hs (27 x 27, LDS) -> Q (27 x 27 row-major, LDS), returns ns.  vstore: 12 x 27 doubles for the reflectors.
__device__ __forceinline__ int fp_strong_basis(const double* Hs_, double* Qout_, double* vstore_) {
    const lds_ptr Hs = to_lds(const_cast<double*>(Hs_)), Qout = to_lds(Qout_), vstore = to_lds(vstore_);   // (generic pointers after the call boundary)
    const int lane = lane_id();
    const bool row = lane < 27;
    const int rl = row ? lane : 0;
    double L[FP_NS_MAX];
#pragma unroll
    for (int k = 0; k < FP_NS_MAX; ++k) L[k] = 0.0;
    double d = row ? Hs[rl * 27 + rl] : -1.0;
    bool used = !row;
    double d0 = 0.0;
    int ns = 0;
#pragma unroll
    for (int k = 0; k < FP_NS_MAX; ++k) {
        if (ns == k) {                                                       // wave-uniform
            const double dmax = wave_max(used ? -1.0 : d);
            if (k == 0) d0 = dmax;
            const double floor_ = (1e-7 * d0 > 1e7) ? 1e-7 * d0 : 1e7;
            if (dmax > floor_) {
                int p = wave_first_lane(!used && d == dmax);
                p = (p < 27) ? p : 0;
                double col = row ? Hs[rl * 27 + p] : 0.0;
#pragma unroll
                for (int j = 0; j < k; ++j) col -= L[j] * wave_bcast(L[j], p);
                col = used ? 0.0 : col;
                const double l = col * rsqrt(dmax);
                L[k] = l;
                d -= l * l;
                used = used || lane == p;
                ns = k + 1;
            }
        }
    }
    // Householder QR of L (27 x ns): reflector k -> vstore[k * 27 + .], beta_k -> betas (registers)
    double beta[FP_NS_MAX];
#pragma unroll
    for (int k = 0; k < FP_NS_MAX; ++k) {
        beta[k] = 0.0;
        if (k < ns) {
            const double x = (row && lane >= k) ? L[k] : 0.0;
            const double sigma = wave_sum(x * x);
            const double xk = wave_bcast(x, k);
            double v = x, bk = 0.0;
            if (sigma > 0.0) {
                const double nr = sqrt(sigma);
                const double alpha = (xk > 0.0) ? -nr : nr;
                v = (lane == k) ? x - alpha : x;
                bk = 1.0 / (sigma + fabs(xk) * nr);                          // 2 / v'v
            }
            beta[k] = bk;
            if (row) vstore[k * 27 + lane] = v;
#pragma unroll
            for (int j = k + 1; j < FP_NS_MAX; ++j) {
                if (j < ns) {
                    const double sj = wave_sum(v * (row ? L[j] : 0.0));
                    L[j] -= bk * sj * v;
                }
            }
        }
    }
    wave_sync();
    // Q = H_0 H_1 ... H_(ns-1): lane r owns row r
    double q[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) q[c] = (c == lane) ? 1.0 : 0.0;
#pragma unroll 1
    for (int k = 0; k < ns; ++k) {
        const lds_ptr vk = vstore + k * 27;
        double bk = 0.0;
#pragma unroll
        for (int j = 0; j < FP_NS_MAX; ++j) bk = (j == k) ? beta[j] : bk;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 27; ++c) s += q[c] * vk[c];
        s *= bk;
#pragma unroll
        for (int c = 0; c < 27; ++c) q[c] -= s * vk[c];
    }
    if (row) {
#pragma unroll
        for (int c = 0; c < 27; ++c) Qout[lane * 27 + c] = q[c];
    }
    wave_sync();
    return ns;
}

// Step 5 on one wavefront: largest eigenvalue of the leading ns x ns block of M' (-> *nrm2), its Cholesky factor (in place, lower
// triangle) and Y = L^-1 [M12 | b1] (in place, columns ns .. 39).  M: 39 x 40 augmented.
__device__ __forceinline__ void fp_eliminate_strong(double* M_, const int ns, double* nrm2) {
    const lds_ptr M = to_lds(M_);
    constexpr int ld = 40;
    const int lane = lane_id();
    const bool row = lane < ns;
    const int rl = row ? lane : 0;
    double m[FP_NS_MAX];
#pragma unroll
    for (int c = 0; c < FP_NS_MAX; ++c) m[c] = (row && c < ns) ? M[rl * ld + c] : 0.0;
    {   // power iteration: only the binade of the norm matters (pinv's tolerance is 39 eps(|M|_2))
        double x = row ? 1.0 : 0.0, rho = 0.0;
        x *= rsqrt(wave_sum(x * x));
#pragma unroll 1
        for (int it = 0; it < 200; ++it) {
            double y = 0.0;
#pragma unroll
            for (int c = 0; c < FP_NS_MAX; ++c) y += m[c] * wave_bcast(x, c);
            rho = wave_sum(x * y);
            const double r = y - rho * x;
            const double res2 = wave_sum(r * r);
            const double up = rho + sqrt(res2);
            if ((eps_of(rho) == eps_of(up) && res2 < 1e-2 * rho * rho) || res2 <= 1e-24 * rho * rho) break;
            x = y * rsqrt(wave_sum(y * y));
        }
        *nrm2 = rho;
    }
    double myinv = 0.0;
#pragma unroll
    for (int k = 0; k < FP_NS_MAX; ++k) {                                    // right-looking Cholesky, lane r owns row r
        if (k < ns) {
            const double dk = wave_bcast(m[k], k);
            const double rs = rsqrt(dk);
            m[k] = (lane == k) ? dk * rs : m[k] * rs;
            myinv = (lane == k) ? rs : myinv;
#pragma unroll
            for (int c = k + 1; c < FP_NS_MAX; ++c) m[c] -= m[k] * wave_bcast(m[k], c);
        }
    }
    if (row) {
#pragma unroll
        for (int c = 0; c < FP_NS_MAX; ++c) if (c <= lane) M[lane * ld + c] = m[c];
    }
    wave_sync();
    const int col = ns + lane;                                               // forward substitution, one column per lane
    if (col < ld) {
#pragma unroll 1
        for (int k = 0; k < ns; ++k) {
            double y = M[k * ld + col];
            for (int j = 0; j < k; ++j) y -= M[k * ld + j] * M[j * ld + col];
            M[k * ld + col] = y / M[k * ld + k];
        }
    }
    wave_sync();
}

// The rotated strong sums: every thread takes correspondences tid, tid + 256, ...; g = sqrt(cs) Q(:, C0:27)' a, U = 27 - C0 components,
// U (U + 1) / 2 + U sums (the last U: g * sqrt(cs) n'w) in halving butterflies -> the calling wavefront's slot (zeroed here).
// Round 5: the sums are taken on the matrix core (gh_kernel.h::StrongGram, v_mfma_f64_16x16x4_f64; tiles in registers across the trips) -- the
// thirteen butterflies per trip were a tenth of an iteration.  scratch: StrongGram<U, 16>::SCRATCH doubles of this wavefront's own.
template <int C0, class SP>
__device__ inline void fp_rotated_sums(const FpLds& s, const SP xi, const SP pp, const int N, double* slot, double* scratch) {
    constexpr int U = 27 - C0;
    constexpr int total = U * (U + 1) / 2 + U;
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    for (int e = lane; e < ((total + 1) & ~1); e += WAVE) slot[e] = 0.0;
    StrongGram<U, 16> gram;
    gram.clear();
    wave_sync();
#pragma unroll 1
    for (int base = 0; base < N; base += FP_THREADS) {
        if (base + wave * WAVE >= N) break;                                  // wave-uniform: no correspondence of this chunk on this wavefront
        const int i = base + tid;
        const bool have = i < N;
        const long ii = have ? i : N - 1;                                    // (same arithmetic on finite data; the contribution is zeroed through cs)
        double o[6], nn[4];
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * ii + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) nn[k] = pp[FP_PP * ii + 10 + k];
        const double cs = have ? pp[FP_PP * ii + 14] : 0.0, om = pp[FP_PP * ii + 15];
        const double sc = sqrt(cs);
        double sq[9], bv[U];
        fp_strong_vector(o, nn, sc, sq);
        if constexpr (C0 == 8) {
            fp_rotate_chunk<8, 10, 0, U>(s.Q, o, sq, bv); fp_rotate_chunk<18, 9, 10, U>(s.Q, o, sq, bv);
        } else {
            static_assert(C0 == 0, "two variants");
            fp_rotate_chunk<0, 9, 0, U>(s.Q, o, sq, bv); fp_rotate_chunk<9, 9, 9, U>(s.Q, o, sq, bv); fp_rotate_chunk<18, 9, 18, U>(s.Q, o, sq, bv);
        }
        gram.add(bv, sc * om, scratch);
    }
    wave_sync();
    gram.store(slot);
}

// Gauss_Helmert.m:38-83 with FaugPapaTFTPoseEstimation.m:87-153 as the callback, one workgroup per problem; xi holds x0 on entry.
// Returns the iteration count (:82); *st: ST_OK, ST_NONFINITE, or ST_RETRY (not this kernel's case).
template <class SP>
__device__ inline int gauss_helmert_fp(FpLds& s, const int own, const double* pts, const int N, const SP xi, const SP pp, int* st, double* dbg) {
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    const bool owner = wave == own;
    const int waves = (N < FP_THREADS) ? (N + WAVE - 1) / WAVE : FP_WAVES;   // wavefronts that hold correspondences in the rotated pass
    double objFunc = 0.0;                                                    // v0' v0, v0 = x0 - x   (:45-46)
    for (int i = tid; i < N; i += FP_THREADS) {
        const Pt6 x = premap(load_pt(pts, i), s.nrm);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = xi[FP_XI * (long)i + k] - x.v[k]; objFunc += d * d; }
    }
    objFunc = block_sum_w<FP_WAVES>(objFunc, s.red);
    int it = 0;
#pragma unroll 1
    for (it = 1; it <= GH_IT_MAX; ++it) {
        double* sdbg = (owner && it == 1) ? dbg : nullptr;                   // phase stamps of the first iteration (debug entry point), slots 16 ..
        phase_stamp(sdbg, 16);
        // ---- func(xi, ti): constraints g, C (FaugPapaTFT...m:114-150) by the owner wavefront, through FaugPapaModel::eval into Mx ----
        if (owner) {
            GhWork g;
            g.p = s.p; g.Tc = s.rvec; g.M = s.Mx; g.u = 27; g.c = 12;
            FaugPapaModel model;
            model.eval(g);
            for (int e = lane; e < 324; e += WAVE) s.Cm[e] = s.Mx[(27 + e / 27) * 40 + e % 27];
            if (lane < 12) s.gneg[lane] = s.Mx[(27 + lane) * 40 + 39];
        }
        phase_stamp(sdbg, 17);
        double T[27];
        load_uniform27(s.p, T);
        // ---- W_i = B_i B_i' + 1e-12 I (:52), finite check (:53-55), pinv tolerance ----
        double fro2 = 0.0;
        for (int i = tid; i < N; i += FP_THREADS) {
            double o[6], f[4], B[4][6], W[4][4];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * (long)i + k];
            tril_block(T, o, f, B);
            block_W(B, W);
            double chk = 0.0, f2 = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { chk += W[a][b]; f2 += W[a][b] * W[a][b]; }
            if (!(fabs(chk) <= 1.79e308) || !(f2 <= 1.79e308)) f2 = 1e300 * 1e300;
            fro2 = (f2 > fro2) ? f2 : fro2;
        }
        const double f2max = block_max_w<FP_WAVES>(fro2, s.red);                         // (also orders the owner's constraint rows before the sums reuse Mx)
        if (!(f2max <= 1.79e308)) { *st = ST_NONFINITE; break; }
        double tolW = 0.0;
        if (!(4.0 * (double)N * eps_of(sqrt(f2max)) < 0.9e-12)) {            // the tolerance 4N eps(max lambda_max) can truncate: it is needed
            double up = 0.0, lo = 0.0;
            for (int i = tid; i < N; i += FP_THREADS) {
                double o[6], f[4], B[4][6], W[4][4], u1, l1;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * (long)i + k];
                tril_block(T, o, f, B);
                block_W(B, W);
                psd_lambda_max_bounds(W, u1, l1);
                up = (u1 > up) ? u1 : up;
                lo = (l1 > lo) ? l1 : lo;
            }
            up = block_max_w<FP_WAVES>(up, s.red);
            lo = block_max_w<FP_WAVES>(lo, s.red);
            double smax = up;
            if (eps_of(lo) != eps_of(up)) {
                smax = 0.0;
                for (int i = tid; i < N; i += FP_THREADS) {
                    double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * (long)i + k];
                    tril_block(T, o, f, B);
                    block_W(B, W);
                    jacobi4<false>(W, V);
#pragma unroll
                    for (int a = 0; a < 4; ++a) smax = (fabs(W[a][a]) > smax) ? fabs(W[a][a]) : smax;
                }
                smax = block_max_w<FP_WAVES>(smax, s.red);
            }
            tolW = 4.0 * (double)N * eps_of(smax);
        }
        phase_stamp(sdbg, 18);
        // ---- weights in the deflated, factored form -> pp: regular part of W+ (with the second + 1e-12 I of :57), n, cs, n'w ----
        bool bad = false;
        for (int i = tid; i < N; i += FP_THREADS) {
            double o[6], f[4], B[4][6], W[4][4], Wp[10], nn[4], cs = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * (long)i + k];
            tril_block(T, o, f, B);
            block_W(B, W);
            bad = !pinv_block_deflated<true>(B, W, tolW, Wp, nn, &cs) || bad;
#pragma unroll
            for (int a = 0; a < 4; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
            const Pt6 x = premap(load_pt(pts, i), s.nrm);
            double om = -(nn[0] * f[0] + nn[1] * f[1] + nn[2] * f[2] + nn[3] * f[3]);
#pragma unroll
            for (int k = 0; k < 6; ++k) om -= (B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3]) * (x.v[k] - o[k]);   // n'w without forming w
            const SP rec = pp + FP_PP * (long)i;
#pragma unroll
            for (int k = 0; k < 10; ++k) rec[k] = Wp[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) rec[10 + k] = nn[k];
            rec[14] = cs; rec[15] = om;
        }
        if (block_any_w<FP_WAVES>(bad, s.red)) { *st = ST_RETRY; break; }                // (barrier: the records of every correspondence are in place)
        phase_stamp(sdbg, 19);
        // ---- the structured sums: ten sweeps for R / ghat, ten for Hs / sum cs (n'w) a, dealt to the four wavefronts ----
        {
            double* Hr = s.Mx; double* Hq = s.Mx + 300;
            if (wave == 0) {                                                 // (two wavefronts: ten sweeps each)
                fp_sweep<false, 0, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 4, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 8, SP>(s, xi, pp, pts, N, T, Hr);
                fp_sweep<true, 2, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 6, SP>(s, xi, pp, pts, N, T, Hq);
                fp_sweep<false, 2, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 6, SP>(s, xi, pp, pts, N, T, Hr);
                fp_sweep<true, 0, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 4, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 8, SP>(s, xi, pp, pts, N, T, Hq);
            } else {
                fp_sweep<false, 1, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 5, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 9, SP>(s, xi, pp, pts, N, T, Hr);
                fp_sweep<true, 3, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 7, SP>(s, xi, pp, pts, N, T, Hq);
                fp_sweep<false, 3, SP>(s, xi, pp, pts, N, T, Hr); fp_sweep<false, 7, SP>(s, xi, pp, pts, N, T, Hr);
                fp_sweep<true, 1, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 5, SP>(s, xi, pp, pts, N, T, Hq); fp_sweep<true, 9, SP>(s, xi, pp, pts, N, T, Hq);
            }
        }
        __syncthreads();
        phase_stamp(sdbg, 20);
        for (int e = tid; e < 729 + 27; e += FP_THREADS) {                // Ghat[(q,i1),(q',i1')] = H[6 tri(q,q') + hht(i1,i1')], both families
            if (e < 729) {
                const int r = e / 27, cc = e % 27;
                const int q = r % 9, i1 = r / 9, qq = cc % 9, i1p = cc / 9;
                const int hi = (q > qq) ? q : qq, lo = (q > qq) ? qq : q;
                const int src = 6 * (hi * (hi + 1) / 2 + lo) + hht_index(i1, i1p);
                s.A1[e] = s.Mx[src];
                s.A2[e] = s.Mx[300 + src];
            } else {
                s.rvec[e - 729] = s.Mx[270 + e - 729];
                s.rsvec[e - 729] = s.Mx[300 + 270 + e - 729];
            }
        }
        __syncthreads();
        phase_stamp(sdbg, 21);
        // ---- the orthogonal basis that aligns the strong subspace ----
        if (owner) {
            const int ns_ = fp_strong_basis(s.A2, s.Q, s.fin);
            if (lane == 0) s.flag[0] = (double)ns_;
        }
        __syncthreads();
        const int ns = (int)s.flag[0];
        // (Taking only the tangential block from the rotated factors and the strong / cross blocks Q(:,1:8)' Hs Q from the formed Hs was tried:
        // 9.6e-9 instead of 7e-11 on a scene whose smallest strong eigenvalue is 1e10 -- the cross block's 1e-3 absolute rounding is not
        // small against THAT.  All 378 + 27 sums come from the rotated factors.)
        phase_stamp(sdbg, 22);
        // ---- rotated strong sums (one correspondence per thread); Y = R Q ----
        // (transposition scratch of the Gram sums: Hs in A2 is dead since the basis was built, so is the upper half of Mx)
        static_assert(FP_WAVES == 2 && StrongGram<27, 16>::SCRATCH <= 729 && 2 * 416 + StrongGram<27, 16>::SCRATCH <= 1664, "scratch of the rotated sums");
        if (wave < waves) fp_rotated_sums<0, SP>(s, xi, pp, N, s.Mx + wave * 416, (wave == 0) ? const_cast<double*>(s.A2) : const_cast<double*>(s.Mx) + 2 * 416);
        __syncthreads();
        for (int e = tid; e < 729; e += FP_THREADS) {                     // Y = R Q -> A2 (Hs is dead now)
            const int r = e / 27, c = e % 27;
            double acc = 0.0;
            for (int k = 0; k < 27; ++k) acc += s.A1[r * 27 + k] * s.Q[k * 27 + c];
            s.A2[e] = acc;
        }
        __syncthreads();
        phase_stamp(sdbg, 23);
        {
            const int nsum = 405, stride = 416;
            for (int e = tid; e < 729 + 405; e += FP_THREADS) {
                if (e < 729) {                                               // Q' (R Q) -> A1
                    const int r = e / 27, c = e % 27;
                    double acc = 0.0;
                    for (int k = 0; k < 27; ++k) acc += s.Q[k * 27 + r] * s.A2[k * 27 + c];
                    s.A1[e] = acc;
                } else if (e - 729 < nsum) {
                    double acc = 0.0;
                    for (int wv_ = 0; wv_ < waves; ++wv_) acc += s.Mx[wv_ * stride + e - 729];
                    s.fin[e - 729] = acc;
                }
            }
        }
        __syncthreads();
        phase_stamp(sdbg, 24);
        // ---- M' (39 x 40, augmented) ----
        double chkM = 0.0;
        for (int e = tid; e < 39 * 40; e += FP_THREADS) {
            const int r = e / 40, c = e % 40;
            double v;
            if (r < 27 && c < 27) {
                v = s.A1[r * 27 + c] + ((r == c) ? 1e-12 : 0.0);
                v += s.fin[(r >= c) ? tri_index(r, c) : tri_index(c, r)];
            } else if (r < 27 && c == 39) {
                double acc = s.fin[378 + r];
                for (int k = 0; k < 27; ++k) acc += s.Q[k * 27 + r] * s.rvec[k];
                v = acc;
            } else if (r >= 27 && c == 39) {
                v = s.gneg[r - 27];
            } else if (r >= 27 && c >= 27) {
                v = (r == c) ? 1e-12 : 0.0;
            } else {                                                         // C Q and its transpose
                const int j = (r >= 27) ? r - 27 : c - 27, cc = (r >= 27) ? c : r;
                double acc = 0.0;
                for (int k = 0; k < 27; ++k) acc += s.Cm[j * 27 + k] * s.Q[k * 27 + cc];
                v = acc;
            }
            chkM += v;
            s.Mx[e] = v;                                                     // (the slots that lived here were last read before the barrier above)
        }
        if (!(fabs(block_sum_w<FP_WAVES>(chkM, s.red)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        phase_stamp(sdbg, 25);
        // ---- block elimination of the strong ns x ns block ----
        if (owner) {
            double nrm2 = 0.0;
            if (ns > 0) fp_eliminate_strong(s.Mx, ns, &nrm2);
            if (lane == 0) s.flag[1] = nrm2;
        }
        __syncthreads();
        phase_stamp(sdbg, 26);
        {
            const int n2 = 39 - ns, w2 = 40 - ns;
            for (int e = tid; e < n2 * w2; e += FP_THREADS) {             // Schur complement, right-hand side included
                const int r = ns + e / w2, c = ns + e % w2;
                double acc = 0.0;
                for (int k = 0; k < ns; ++k) acc += s.Mx[k * 40 + r] * s.Mx[k * 40 + c];
                s.Mx[r * 40 + c] -= acc;
            }
        }
        __syncthreads();
        phase_stamp(sdbg, 27);
        // ---- aux = pinv(M + 1e-12 I) b (:67) ----
        if (owner) {
            const int n2 = 39 - ns;
            double* S = s.Mx + ns * 40 + ns;
            if (n2 <= TRID_MAX && ns > 0) {
                const double tol = 39.0 * eps_of(s.flag[1]);
                int kept, fail;
                wave_pinv_solve_trid<FP_TRID_IN_REGISTERS, false>(S, 40, n2, tol, s.dt + ns, s.sm, s.A1, &kept, &fail, sdbg);
                if (fail && lane == 0) s.flag[2] = 1.0;
            } else {
                // no (or a small) strong block -- every strong direction under pinv's tolerance for the weights, or a degenerate sample:
                // the eigen-decomposition of the whole remainder, its own largest eigenvalue for the tolerance
                int fail;
                const double lam = wave_eigh_ql(S, 40, s.A1, n2, n2, s.sm, &fail);
                if (fail && lane == 0) s.flag[2] = 1.0;                      // (an eigenvalue that did not converge: the triplet goes to the generic kernel)
                double amax = wave_max((lane < n2) ? fabs(lam) : 0.0);
                if (ns > 0 && s.flag[1] > amax) amax = s.flag[1];
                const double tol = 39.0 * eps_of(amax);
                if (lane < n2) {
                    const double* vk = s.A1 + eig_row(n2, lane) * n2;
                    double d = 0.0;
                    for (int r = 0; r < n2; ++r) d += vk[r] * S[r * 40 + n2];
                    s.sm[lane] = (fabs(lam) > tol) ? d / lam : 0.0;
                }
                wave_sync();
                if (lane < n2) {
                    double x = 0.0;
                    for (int k = 0; k < n2; ++k) x += s.A1[eig_row(n2, k) * n2 + lane] * s.sm[k];
                    s.dt[ns + lane] = x;
                }
                wave_sync();
            }
            phase_stamp(sdbg, 32);
            // z1 = L^-T (y_b - Y z2)
            double t = 0.0;
            if (lane < ns) {
                t = s.Mx[lane * 40 + 39];
                for (int c = ns; c < 39; ++c) t -= s.Mx[lane * 40 + c] * s.dt[c];
            }
#pragma unroll 1
            for (int k = ns - 1; k >= 0; --k) {
                const double zk = wave_bcast(t, k) / s.Mx[k * 40 + k];
                if (lane == k) t = zk;
                else if (lane < k) t -= s.Mx[k * 40 + lane] * zk;
            }
            if (lane < ns) s.dt[lane] = t;
            wave_sync();
            double dtv = 0.0;
            if (lane < 27) for (int c = 0; c < 27; ++c) dtv += s.Q[lane * 27 + c] * s.dt[c];
            wave_sync();
            if (lane < 27) s.dt[lane] = dtv;
        }
        __syncthreads();
        phase_stamp(sdbg, 33);
        double dTr[27];
        load_uniform27(s.dt, dTr);
        // ---- v = -B' W+ (A dt - w)   (:69); v overwrites the record's W+ slots (dead until the next weight pass) ----
        double obj = 0.0, diff = 0.0;
        for (int i = tid; i < N; i += FP_THREADS) {
            double o[6], f[4], B[4][6], Ad[4], wv[4], r[4], Wp[10], nn[4];
            const SP rec = pp + FP_PP * (long)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = xi[FP_XI * (long)i + k];
#pragma unroll
            for (int k = 0; k < 10; ++k) Wp[k] = rec[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) nn[k] = rec[10 + k];
            const double cs = rec[14];
            tril_block(T, o, f, B);
            {
                double m[3][3], t1[3][3], t2[3][3];
                tril_slices(dTr, o, m, t1, t2);
                tril_quad(m, o[2], o[3], o[4], o[5], Ad);                    // Ap_i dt
            }
            const Pt6 x = premap(load_pt(pts, i), s.nrm);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double sw = -f[a];
#pragma unroll
                for (int k = 0; k < 6; ++k) sw -= B[a][k] * (x.v[k] - o[k]);
                wv[a] = Ad[a] - sw;                                          // A dt - w
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
                r[a] = wp_at(Wp, a, 0) * wv[0] + wp_at(Wp, a, 1) * wv[1] + wp_at(Wp, a, 2) * wv[2] + wp_at(Wp, a, 3) * wv[3];
            const double sterm = cs * (nn[0] * wv[0] + nn[1] * wv[1] + nn[2] * wv[2] + nn[3] * wv[3]);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double bn = B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3];
                const double v = -(B[0][k] * r[0] + B[1][k] * r[1] + B[2][k] * r[2] + B[3][k] * r[3]) - bn * sterm;
                rec[k] = v;
                obj += v * v;
                const double d = o[k] - x.v[k] - v;
                diff += d * d;
            }
        }
        obj = block_sum2(obj, &diff, s.red);
        phase_stamp(sdbg, 34);
        double ndt2 = 0.0;
#pragma unroll
        for (int k = 0; k < 27; ++k) ndt2 += dTr[k] * dTr[k];
        if (sqrt(ndt2) < GH_TOL && sqrt(diff) < GH_TOL) break;               // :71-73 (dy is empty)
        if (obj > objFunc) break;                                            // :75-76, factor = 1
        objFunc = obj;                                                       // :78
        for (int i = tid; i < N; i += FP_THREADS) {                       // xi = x + v; ti = ti + dt   (:80)
            const Pt6 x = premap(load_pt(pts, i), s.nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) xi[FP_XI * (long)i + k] = x.v[k] + pp[FP_PP * (long)i + k];
        }
        if (tid < 27) s.p[tid] += s.dt[tid];
        __syncthreads();
    }
    __syncthreads();
    return (it > GH_IT_MAX) ? GH_IT_MAX : it;                                // :82
}

template <bool STATE_IN_LDS>
__global__ void __launch_bounds__(FP_THREADS, FP_WG_PER_CU * FP_WAVES / 4) k_fp_block(const GhWgArgs a) {   // (second argument: wavefronts per SIMD)
    typedef typename FpState<STATE_IN_LDS>::ptr SP;
    TFF_DYNAMIC_LDS(double, smem);
    FpLds& s = *reinterpret_cast<FpLds*>(smem);
    const int tid = thread_in_block(), lane = lane_id(), wave = wave_in_block();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        if (a.status[b] != ST_OK) continue;                                  // block-uniform
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        const double* pts = a.corresp + b * 6 * (long)N;
        SP xi, pp;                                                           // per-correspondence state: LDS, or the block's global slice
        if constexpr (STATE_IN_LDS) xi = to_lds(smem + FP_LDS_DOUBLES);
        else xi = a.spill + blockIdx.x * a.spill_stride;
        pp = xi + FP_XI * (long)N;
        const int own = pick_serial_wave_w<FP_WAVES>(s.red);
        const double* r = a.rec + b * GH_REC_DOUBLES;                        // t 27 | pa 18 | epi 6 | nrm 9   (k_gh_linear)
        if (tid < 27) s.p[tid] = r[tid];                                     // param0 = T(:)   (FaugPapaTFT...m:65)
        if (tid < 9) s.nrm[tid] = r[51 + tid];
        if (tid >= 64 && tid < 76) {                                         // P1 = [I|0], P2 = [reshape(a(1:9),3,3) e21], P3 likewise   (linearTFT.m:88-90)
            const int e = tid - 64, rr = e >> 2, c = e & 3;
            s.cam[0][e] = (rr == c) ? 1.0 : 0.0;
            s.cam[1][e] = (c < 3) ? r[27 + 3 * c + rr] : r[45 + rr];
            s.cam[2][e] = (c < 3) ? r[27 + 9 + 3 * c + rr] : r[45 + 3 + rr];
        }
        if (tid == 0) s.flag[2] = 0.0;
        __syncthreads();
        {                                                                    // x_est: reprojection of the projective triangulation   (:58-61)
            double PA[12], PB[12], PC[12];
            load_uniform12(s.cam[0], PA);
            load_uniform12(s.cam[1], PB);
            load_uniform12(s.cam[2], PC);
#pragma unroll 1
            for (int i = tid; i < N; i += FP_THREADS) {
                const Pt6 p = premap(load_pt(pts, i), s.nrm);
                double X[4];
                dlt_point<true>(PA, PB, PC, s.cam[0], s.cam[1], s.cam[2], true, p.v[0], p.v[1], p.v[2], p.v[3], p.v[4], p.v[5], X);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const double (&P)[12] = (v == 0) ? PA : ((v == 1) ? PB : PC);
                    const double aa = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
                    const double bb = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
                    const double cc = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
                    xi[FP_XI * (long)i + 2 * v] = aa / cc;
                    xi[FP_XI * (long)i + 2 * v + 1] = bb / cc;
                }
            }
        }
        __syncthreads();
        int gst = ST_OK;
        const int iters = gauss_helmert_fp<SP>(s, own, pts, N, xi, pp, &gst, a.dbg ? a.dbg + b * DBG_STRIDE : nullptr);
        phase_stamp((a.dbg && wave == own) ? a.dbg + b * DBG_STRIDE : nullptr, 35);
        if (gst == ST_OK && s.flag[2] != 0.0) gst = ST_RETRY;                // an eigenpair of the pseudo-inverse did not converge / lost its orthogonality (never observed)
        if ((a.flags & FLAG_DBG_FP_HANDOVER) && gst == ST_OK && b % 3 == 0) gst = ST_RETRY;   // test hook: exercise the hand-over
        if (wave == own) {
            if (lane < 27) a.topt[b * 27 + lane] = s.p[lane];
            if (lane == 0) {
                if (a.iter) a.iter[b] = iters;
                if (gst == ST_RETRY) a.status[b] = ST_RETRY;                 // k_gh_block<FaugPapaModel> redoes this triplet
                else if (gst != ST_OK) a.status[b] = -gst;                   // negative: reported after k_gh_finish has produced the outputs
            }
        }
    }
}

}  // namespace tff
