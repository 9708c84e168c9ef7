// Gauss-Helmert refinement of the trifocal tensor (Optimization/Gauss_Helmert.m:38-83)
// and ResslTFTPoseEstimation (TFT_methods/ResslTFTPoseEstimation.m:47-177) as one
// fused kernel, one wavefront per triplet.
//
// The reference builds dense 4N x 27 / 4N x 6N / 4N x 4N matrices and calls pinv on
// an identity (6N x 6N) and on W (4N x 4N) every iteration.  Their structure is block
// diagonal -- one 4 x 6 block B_i per correspondence -- so here
//   W = B pinv(P) B'             -> 4x4 blocks W_i = B_i B_i'                    (:52, P = I)
//   pinv(W + 1e-12 I) + 1e-12 I  -> per-lane Jacobi eigen-decomposition of each block with
//                                   pinv's GLOBAL tolerance 4N * eps(max_i lambda_max)  (:57)
//   A'WA, A'Ww                   -> A_i = Ap_i D (D = dT/dparams, wave-uniform), so
//                                   A'WA = D' Ghat D with Ghat = sum_i Ap_i' W_i Ap_i
//                                        = sum_i (h1 h1') (x) (K_i W_i K_i'),  K_i = S3 (x) S2 :
//                                   6 x 45 + 27 sums accumulated one correspondence per lane in
//                                   ten 30-accumulator sweeps (halving-butterfly reductions)  (:59-62)
//   pinv(M + 1e-12 I) b          -> pivoted elimination of the (u+c) x (u+c) KKT system; equal to
//                                   pinv whenever no singular value falls under its tolerance
//                                   (the generic case; a numerically rank-deficient system is
//                                   reported as status TFF_ST_RANK)                          (:67)
//   v = -B' W (A dt - w)         -> per lane                                                (:69)
// with the reference's stop tests, `factor = 1`, and the last step not applied when the
// objective rises (:71-80).  Nothing of size 4N x anything is ever formed.
//
// The weight blocks are evaluated at the accuracy of the reference's FORMULAS (pinv_block_deflated below; in the workgroup
// kernel also with the strong direction factored): that path agrees with a 50-digit evaluation of the iteration to ~1e-10
// (tests/test_gpu_gh_noise.py).  A dense LAPACK evaluation of the same formulas -- the numpy oracle, MATLAB itself -- carries
// 1e-6 .. 1e-3 of its own rounding noise, so agreement with IT is statistical (profiles/r2_gh_noise_mp.txt).
#pragma once
#include "tft_kernel.h"

namespace tff {

constexpr int ST_RANK = 4;            // Nordberg: P2(:,1:3) or P3(:,1:3) of rank < 2 (TFF_ST_RANK; a rank-deficient KKT matrix takes the truncated pinv path)
constexpr int GH_IT_MAX = 400;        // Gauss_Helmert.m:38
constexpr double GH_TOL = 1e-6;       // Gauss_Helmert.m:39

struct GhWork {                       // per-wave LDS carve-up for one Gauss-Helmert problem
    double* p;      // u        parameters
    double* dt;     // u + c    solution of the KKT system
    double* Tc;     // 27       tensor of the current parameters
    double* dT;     // 27       D * dt
    double* D;      // 27 x u   dT/dp, row-major
    double* G;      // 27 x 27  Ghat
    double* H;      // 270 + 27 accumulated sums: H[6 e + hh] (e: lower triangle of the 9x9 Z), then ghat[27]
    double* Y;      // 27 x u   Ghat * D
    double* M;      // (u+c) x (u+c+1) augmented KKT matrix
    double* V;      // (u+c)^2 eigenvectors of the KKT matrix (pinv path)
    double* xi;     // 6N       current estimates of the observations
    double* pp;     // 14N      per correspondence: W+ (10, packed lower) and W+ w (4); then v (6)
    double* S;      // workgroup kernel, factored weights: u(u+1)/2 + u sums of the strong-direction terms (LDS), else null
    int u, c;
};
__host__ __device__ inline int gh_lds_doubles(int u, int c, int N) {
    const int n = u + c;
    return 2 * ((u + 1) & ~1) + 2 * c + 28 + 28 + 27 * u + 729 + 298 + 27 * u + n * (n + 1) + n * n + 6 * N + 14 * N + 8;
}
__device__ inline GhWork gh_carve(double* base, int u, int c, int N) {
    GhWork g;
    const int n = u + c;
    double* q = base;
    g.p = q; q += (u + 1) & ~1;
    g.dt = q; q += ((u + 1) & ~1) + 2 * c;
    g.Tc = q; q += 28;
    g.dT = q; q += 28;
    g.D = q; q += 27 * u;
    g.G = q; q += 729;
    g.H = q; q += 298;
    g.Y = q; q += 27 * u;
    g.M = q; q += n * (n + 1);
    g.V = q; q += n * n;
    g.xi = q; q += 6 * N;
    g.pp = q; q += 14 * N;
    g.S = nullptr;
    g.u = u; g.c = c;
    return g;
}

__device__ __forceinline__ double eps_of(double x) {                        // MATLAB eps(x) for normal x > 0
    const long long e = (__double_as_longlong(x) >> 52) & 0x7ff;
    return __longlong_as_double((e > 52 ? e - 52 : 1) << 52);
}

// ---- the per-correspondence trilinearity block (ResslTFTPoseEstimation.m:141-161) ----
// quad(m) = vec(S2' m S3), S2 = [0 -1; -1 0; y2 x2], S3 likewise; entry a2 + 2 a3.
__device__ __forceinline__ void tril_quad(const double (&m)[3][3], double x2, double y2, double x3, double y3, double (&o)[4]) {
    double v0[3], v1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { v0[j] = y3 * m[j][2] - m[j][1]; v1[j] = x3 * m[j][2] - m[j][0]; }
    o[0] = y2 * v0[2] - v0[1];
    o[1] = x2 * v0[2] - v0[0];
    o[2] = y2 * v1[2] - v1[1];
    o[3] = x2 * v1[2] - v1[0];
}
__device__ __forceinline__ void tril_slices(const double (&T)[27], const double (&o)[6], double (&m)[3][3], double (&t1)[3][3], double (&t2)[3][3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            t1[j][k] = T[j + 3 * k];
            t2[j][k] = T[j + 3 * k + 9];
            m[j][k] = o[0] * t1[j][k] + o[1] * t2[j][k] + T[j + 3 * k + 18];
        }
}
// f (4) and B (4 x 6) at the observation estimate o = [x1 y1 x2 y2 x3 y3]
__device__ __forceinline__ void tril_block(const double (&T)[27], const double (&o)[6], double (&f)[4], double (&B)[4][6]) {
    double m[3][3], t1[3][3], t2[3][3], c0[4], c1[4];
    tril_slices(T, o, m, t1, t2);
    tril_quad(m, o[2], o[3], o[4], o[5], f);
    tril_quad(t1, o[2], o[3], o[4], o[5], c0);                               // :157
    tril_quad(t2, o[2], o[3], o[4], o[5], c1);                               // :158
    const double u3[2] = {o[5] * m[2][2] - m[2][1], o[4] * m[2][2] - m[2][0]};   // S3' J3' [x1;1]   (:159)
    const double u2[2] = {o[3] * m[2][2] - m[1][2], o[2] * m[2][2] - m[0][2]};   // S2' K3 [x1;1]    (:160)
#pragma unroll
    for (int a3 = 0; a3 < 2; ++a3)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int r = a2 + 2 * a3;
            B[r][0] = c0[r];
            B[r][1] = c1[r];
            B[r][2] = (a2 == 1) ? u3[a3] : 0.0;
            B[r][3] = (a2 == 0) ? u3[a3] : 0.0;
            B[r][4] = (a3 == 1) ? u2[a2] : 0.0;
            B[r][5] = (a3 == 0) ? u2[a2] : 0.0;
        }
}
__device__ __forceinline__ void load_uniform27(const double* p, double (&u)[27]) {
#pragma unroll
    for (int c = 0; c < 27; ++c) u[c] = wave_uniform(p[c]);
}

// cyclic Jacobi on a symmetric 4x4 (per lane); V accumulates eigenvectors when WITH_V
template <bool WITH_V>
__device__ __forceinline__ void jacobi4_rot(double (&A)[4][4], double (&V)[4][4], const int p, const int q) {
    const double apq = A[p][q];
    if (apq == 0.0) return;
    const double app = A[p][p], aqq = A[q][q];
    const double tau = (aqq - app) / (2.0 * apq);
    const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double c = rsqrt(1.0 + t * t), s = t * c;
    A[p][p] = app - t * apq;
    A[q][q] = aqq + t * apq;
    A[p][q] = A[q][p] = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r == p || r == q) continue;
        const double arp = A[r][p], arq = A[r][q];
        A[r][p] = A[p][r] = c * arp - s * arq;
        A[r][q] = A[q][r] = s * arp + c * arq;
    }
    if (WITH_V) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double vkp = V[k][p], vkq = V[k][q];
            V[k][p] = c * vkp - s * vkq;
            V[k][q] = s * vkp + c * vkq;
        }
    }
}
template <bool WITH_V>
__device__ __forceinline__ void jacobi4(double (&A)[4][4], double (&V)[4][4]) {
    if (WITH_V) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dg += A[i][i] * A[i][i];
#pragma unroll
            for (int j = 0; j < i; ++j) off += A[i][j] * A[i][j];
        }
        if (!(off > 1e-36 * dg)) break;
        jacobi4_rot<WITH_V>(A, V, 0, 1); jacobi4_rot<WITH_V>(A, V, 0, 2); jacobi4_rot<WITH_V>(A, V, 0, 3);
        jacobi4_rot<WITH_V>(A, V, 1, 2); jacobi4_rot<WITH_V>(A, V, 1, 3); jacobi4_rot<WITH_V>(A, V, 2, 3);
    }
}
__device__ __forceinline__ void block_W(const double (&B)[4][6], double (&W)[4][4]) {   // B B' + 1e-12 I
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double a = (i == j) ? 1e-12 : 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) a += B[i][k] * B[j][k];
            W[i][j] = W[j][i] = a;
        }
}

// K = S3 (x) S2 (9 x 4): row q = j + 3k, column a = a2 + 2 a3
template <int j>
__device__ __forceinline__ double S_el(int a, double x, double y) { return (j == 2) ? ((a == 0) ? y : x) : ((j == a) ? 0.0 : -1.0); }
template <int q>
__device__ __forceinline__ void K_row(double x2, double y2, double x3, double y3, double (&kr)[4]) {
#pragma unroll
    for (int a3 = 0; a3 < 2; ++a3)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) kr[a2 + 2 * a3] = S_el<q % 3>(a2, x2, y2) * S_el<q / 3>(a3, x3, y3);
}
__host__ __device__ constexpr int tri_row_of(int e) { int r = 0; while ((r + 1) * (r + 2) / 2 <= e) ++r; return r; }
__host__ __device__ constexpr int tri_col_of(int e) { return e - tri_row_of(e) * (tri_row_of(e) + 1) / 2; }

struct GhPoint { double o[6]; double Wp[10]; double ww[4]; };
__device__ __forceinline__ double wp_at(const double (&Wp)[10], int a, int b) { return (a >= b) ? Wp[a * (a + 1) / 2 + b] : Wp[b * (b + 1) / 2 + a]; }

// five entries of Z = K W+ K' (x) six products h_a h_b -> 30 accumulators
template <int E>
__device__ __forceinline__ void gh_accum_entry(const GhPoint& pt, const double (&hh)[6], double* acc) {
    constexpr int q = tri_row_of(E), qq = tri_col_of(E);
    double kr[4], kc[4];
    K_row<q>(pt.o[2], pt.o[3], pt.o[4], pt.o[5], kr);
    K_row<qq>(pt.o[2], pt.o[3], pt.o[4], pt.o[5], kc);
    double z = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double y = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) y += kr[a] * wp_at(pt.Wp, a, b);
        z += y * kc[b];
    }
#pragma unroll
    for (int h = 0; h < 6; ++h) acc[h] += hh[h] * z;
}
template <int CH>
__device__ __forceinline__ void gh_accum_chunk(const GhPoint& pt, const double (&hh)[6], double (&acc)[32]) {
    if constexpr (CH < 9) {
        gh_accum_entry<5 * CH + 0>(pt, hh, acc + 0);
        gh_accum_entry<5 * CH + 1>(pt, hh, acc + 6);
        gh_accum_entry<5 * CH + 2>(pt, hh, acc + 12);
        gh_accum_entry<5 * CH + 3>(pt, hh, acc + 18);
        gh_accum_entry<5 * CH + 4>(pt, hh, acc + 24);
    }
}
// ghat[q + 9 i1] += h[i1] * (K[q] . ww)
__device__ __forceinline__ void gh_accum_rhs(const GhPoint& pt, double (&acc)[32]) {
    double kq[9];
    {
        double kr[4];
#define TFF_KQ(Q) K_row<Q>(pt.o[2], pt.o[3], pt.o[4], pt.o[5], kr); kq[Q] = kr[0] * pt.ww[0] + kr[1] * pt.ww[1] + kr[2] * pt.ww[2] + kr[3] * pt.ww[3];
        TFF_KQ(0) TFF_KQ(1) TFF_KQ(2) TFF_KQ(3) TFF_KQ(4) TFF_KQ(5) TFF_KQ(6) TFF_KQ(7) TFF_KQ(8)
#undef TFF_KQ
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) { acc[q] += pt.o[0] * kq[q]; acc[9 + q] += pt.o[1] * kq[q]; acc[18 + q] += kq[q]; }
}

template <int CH>
__device__ inline void gh_sweep(const GhWork& g, int N) {
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = lane; i < N; i += WAVE) {
        GhPoint pt;
#pragma unroll
        for (int k = 0; k < 6; ++k) pt.o[k] = g.xi[6 * i + k];
#pragma unroll
        for (int k = 0; k < 10; ++k) pt.Wp[k] = g.pp[14 * i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) pt.ww[k] = g.pp[14 * i + 10 + k];
        if constexpr (CH < 9) {
            const double hh[6] = {pt.o[0] * pt.o[0], pt.o[0] * pt.o[1], pt.o[0], pt.o[1] * pt.o[1], pt.o[1], 1.0};
            gh_accum_chunk<CH>(pt, hh, acc);
        } else {
            gh_accum_rhs(pt, acc);
        }
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0) {
        if (CH < 9) { if (idx < 30) g.H[30 * CH + idx] = tot; }              // H[6 e + hh], e = 5 CH + idx/6
        else if (idx < 27) g.H[270 + idx] = tot;
    }
}

// Solve the n x n system held as an augmented n x (n+1) matrix in LDS (x -> sol[0..n), false when a pivot is
// negligible).  The matrix lives in registers: lane r owns row r (compile-time n <= 64), Gauss-Jordan elimination
// with partial pivoting and no physical row exchange -- the pivot row of column k is the not-yet-used lane with
// the largest |a[k]|, broadcast by v_readlane; at the end the lane that served as pivot of column k holds x_k.
// ~n^2 readlanes + n^2/2 FMAs per lane and no LDS traffic inside the loop.
// min_pivot_rel: a pivot at or below this fraction of the largest entry makes the solve report failure (callers whose fall-back is the truncated
// pseudo-inverse: 1e-10; PiCol, which certifies the spectrum separately -- pi_wg_kernel.h --: 1e-15).
template <int n>
__device__ inline bool wave_solve_gj(const double* M, double* sol, const double min_pivot_rel = 1e-10) {
    constexpr int ld = n + 1;
    const int lane = lane_id();
    const bool valid = lane < n;
    double a[ld];
    {
        const double* row = M + (valid ? lane : 0) * ld;
#pragma unroll
        for (int c = 0; c < ld; ++c) a[c] = row[c];
    }
    double amax = 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) amax = (valid && fabs(a[c]) > amax) ? fabs(a[c]) : amax;
    amax = wave_max(amax);
    bool ok = amax > 0.0 && amax < 1e300;
    bool active = valid;
    double x = 0.0;
    int mycol = 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double v = active ? fabs(a[k]) : -1.0;
        const double best = (n <= 32) ? wave_max32_finite(v) : wave_max(v);  // rows live in lanes 0..n-1
        if (!(best > min_pivot_rel * amax)) ok = false;
        int p = wave_first_lane(active && v == best);                        // wave-uniform
        p = (p < 64) ? p : 0;
        const double ipiv = 1.0 / wave_bcast(a[k], p);
        const bool elim = valid && lane != p;
        const double f = elim ? a[k] * ipiv : 0.0;
#pragma unroll
        for (int c = k + 1; c < ld; ++c) a[c] -= f * wave_bcast(a[c], p);
        if (lane == p) { active = false; mycol = k; x = ipiv; }             // x_k = rhs / pivot once the other columns are cleared
    }
    if (valid) sol[mycol] = a[n] * x;
    wave_sync();
    return ok;
}

// cameras of the initial projective reconstruction: P1 = [I|0] -> Pfin[0], P2 -> P[0], P3 -> P[1] (row-major)
__device__ inline void gh_linear_cameras(PoseLds* w) {
    const int lane = lane_id();
    if (lane < 12) {
        const int r = lane >> 2, c = lane & 3;
        w->Pfin[0][lane] = (r == c) ? 1.0 : 0.0;
        w->P[0][lane] = (c < 3) ? w->pa[3 * c + r] : w->epi[r];              // P2 = [reshape(a(1:9),3,3) e21]    (linearTFT.m:89)
        w->P[1][lane] = (c < 3) ? w->pa[9 + 3 * c + r] : w->epi[3 + r];      // P3 = [reshape(a(10:18),3,3) e31]  (linearTFT.m:90)
    }
    wave_sync();
}

__device__ __forceinline__ double det3(const double (&A)[3][3]) {
    return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
         + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}
// signed cofactor, FaugPapaTFTPoseEstimation.m:156-159 (`minor`), i and j static
template <int i, int j>
__device__ __forceinline__ double cof3(const double (&A)[3][3]) {
    constexpr int r0 = (i == 0) ? 1 : 0, r1 = (i == 2) ? 1 : 2, c0 = (j == 0) ? 1 : 0, c1 = (j == 2) ? 1 : 2;
    const double d = A[r0][c0] * A[r1][c1] - A[r0][c1] * A[r1][c0];
    return ((i + j) % 2 == 0) ? d : -d;
}

// ---- Faugeras-Papadopoulo: all 27 entries, 12 algebraic constraints (FaugPapaTFTPoseEstimation.m:48-153) ----
struct FaugPapaModel {
    static constexpr int U = 27, C = 12;
    static constexpr bool IDENTITY_D = true;
    static constexpr int WG_PER_CU = 4;                                      // workgroups per CU the block kernel is compiled for (measured best)
    static constexpr bool SPARSE_DT = false;
    __device__ __forceinline__ void share(double*) const {}
    __device__ __forceinline__ void adopt(const double*) {}
    static constexpr bool REDUNDANT_CONSTRAINTS = true;     // trifocal tensors have codimension 9 < 12: singular KKT, pinv truncates
    __device__ inline void init(PoseLds* w, GhWork& g) {                    // param0 = T(:)   (:65)
        const int lane = lane_id();
        if (lane < 27) g.p[lane] = w->t[lane];
        wave_sync();
        gh_linear_cameras(w);
    }
    __device__ inline void eval(GhWork& g) const {
        const int lane = lane_id();
        constexpr int u = U, n = U + C, ld = n + 1;
        for (int e = lane; e < n * ld; e += WAVE) g.M[e] = 0.0;
        if (lane < 27) g.Tc[lane] = g.p[lane];
        wave_sync();
        const double* T = g.Tc;
        if (lane < 3) {                                                      // det(T_i) = 0   (:117-124)
            const int i = lane;
            double A[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) A[j][k] = T[j + 3 * k + 9 * i];
            const int row = u + i;
            g.M[row * ld + n] = -det3(A);
#define TFF_C1(J, K) { const double v = cof3<J, K>(A); const int col = J + 3 * K + 9 * i; g.M[row * ld + col] = v; g.M[col * ld + row] = v; }
            TFF_C1(0, 0) TFF_C1(0, 1) TFF_C1(0, 2) TFF_C1(1, 0) TFF_C1(1, 1) TFF_C1(1, 2) TFF_C1(2, 0) TFF_C1(2, 1) TFF_C1(2, 2)
#undef TFF_C1
        } else if (lane < 12) {                                              // extended rank constraints   (:126-150)
            const int c9 = lane - 3;
            // (k2,k3,l2,l3) in the nesting order of the reference's loops
            const int k2 = (c9 >= 6) ? 1 : 0;
            const int k3 = (c9 == 4 || c9 == 5 || c9 == 8) ? 1 : 0;
            const int l2 = (c9 == 0 || c9 == 1 || c9 == 4) ? 1 : 2;
            const int l3 = (c9 == 0 || c9 == 2 || c9 == 6) ? 1 : 2;
            const int pk2k3 = k2 + 3 * k3, pk2l3 = k2 + 3 * l3, pl2l3 = l2 + 3 * l3, pl2k3 = l2 + 3 * k3;
            double A1[3][3], A2[3][3], A3[3][3], A4[3][3];                   // rows: tensor positions, columns: slices
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double a = T[pk2k3 + 9 * i], b = T[pk2l3 + 9 * i], c = T[pl2l3 + 9 * i], d = T[pl2k3 + 9 * i];
                A1[0][i] = a; A1[1][i] = b; A1[2][i] = c;
                A2[0][i] = a; A2[1][i] = d; A2[2][i] = c;
                A3[0][i] = d; A3[1][i] = b; A3[2][i] = c;
                A4[0][i] = a; A4[1][i] = d; A4[2][i] = b;
            }
            const double d1 = det3(A1), d2 = det3(A2), d3 = det3(A3), d4 = det3(A4);
            const int row = u + lane;
            g.M[row * ld + n] = -(d1 * d2 - d3 * d4);
#define TFF_C9(I1) { \
            const double v0 = cof3<I1, 0>(A1) * d2 + d1 * cof3<I1, 0>(A2) - d3 * cof3<I1, 0>(A4); \
            const double v1 = cof3<I1, 1>(A1) * d2 - cof3<I1, 1>(A3) * d4 - d3 * cof3<I1, 2>(A4); \
            const double v2 = cof3<I1, 2>(A1) * d2 + d1 * cof3<I1, 2>(A2) - cof3<I1, 2>(A3) * d4; \
            const double v3 = d1 * cof3<I1, 1>(A2) - cof3<I1, 0>(A3) * d4 - d3 * cof3<I1, 1>(A4); \
            const int c0 = pk2k3 + 9 * I1, c1 = pk2l3 + 9 * I1, c2 = pl2l3 + 9 * I1, c3 = pl2k3 + 9 * I1; \
            g.M[row * ld + c0] = v0; g.M[c0 * ld + row] = v0; g.M[row * ld + c1] = v1; g.M[c1 * ld + row] = v1; \
            g.M[row * ld + c2] = v2; g.M[c2 * ld + row] = v2; g.M[row * ld + c3] = v3; g.M[c3 * ld + row] = v3; }
            TFF_C9(0) TFF_C9(1) TFF_C9(2)
#undef TFF_C9
        }
        wave_sync();
    }
};

// x = pinv(M) b for the symmetric n x n matrix held (with b as column n) in the augmented n x (n+1) array:
// eigen-decomposition M = V L V' (wave_eigh_ql), singular values |lambda_k|, MATLAB's pinv tolerance n * eps(max |lambda|),
// x = sum_{|lambda_k| > tol} v_k (v_k' b) / lambda_k.  Needed when the constraints are redundant (KKT singular).
// ZT: n * n doubles (eigenvectors, one per row), scr: 2 n doubles.  One wavefront.  INLINE: the workgroup kernels, whose register
// budget (launch bounds) must cover the solver.
template <bool INLINE = false>
__device__ inline void wave_pinv_solve_sym(double* M, double* ZT, int n, double* sol, double* scr) {
    const int lane = lane_id();
    const int ld = n + 1;
    int fail;
    const double lam = INLINE ? wave_eigh_ql(M, ld, ZT, n, n, scr, &fail) : wave_eigh_ql_call(M, ld, ZT, n, n, scr, &fail);
    const double amax = wave_max((lane < n) ? fabs(lam) : 0.0);
    const double tol = (double)n * eps_of(amax);
    if (lane < n) {
        const double* vk = ZT + eig_row(n, lane) * n;
        double d = 0.0;
        for (int r = 0; r < n; ++r) d += vk[r] * M[r * ld + n];
        scr[lane] = (fabs(lam) > tol) ? d / lam : 0.0;
    }
    wave_sync();
    if (lane < n) {
        double x = 0.0;
        for (int k = 0; k < n; ++k) x += ZT[eig_row(n, k) * n + lane] * scr[k];
        sol[lane] = x;
    }
    wave_sync();
}

// ---- Nordberg: three orthogonal matrices (axis-angle) + 10-entry sparse tensor, 19 parameters, 1 constraint
//      (NordbergTFTPoseEstimation.m:47-222) ----
struct NordbergModel {
    static constexpr int U = 19, C = 1;
    static constexpr int WG_PER_CU = 3;
    static constexpr bool IDENTITY_D = false;
    static constexpr bool SPARSE_DT = false;
    __device__ __forceinline__ void share(double*) const {}
    __device__ __forceinline__ void adopt(const double*) {}
    static constexpr bool REDUNDANT_CONSTRAINTS = false;
    int bad;                                                                 // P2 or P3 of rank < 2: H(4,1:3) = null(.)' is an error in the reference (:56-62)

    __device__ static __forceinline__ void sparse_pos(int k, int& c, int& d, int& j) {
        // param_ind = [1,7,10,12,16,19:22,25] (1-based, column-major 3x3x3)   (:82)
        const int idx = (k == 0) ? 0 : (k == 1) ? 6 : (k == 2) ? 9 : (k == 3) ? 11 : (k == 4) ? 15 : (k == 5) ? 18 : (k == 6) ? 19
                      : (k == 7) ? 20 : (k == 8) ? 21 : 24;
        c = idx % 3; d = (idx / 3) % 3; j = idx / 9;
    }
    // M (M'M)^(-1/2), then sign(det) * M   (:68-70)
    __device__ static inline Mat3 orthogonalise(const Mat3& M) {
        double S[3][3], Q[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] = M.m[0][i] * M.m[0][j] + M.m[1][i] * M.m[1][j] + M.m[2][i] * M.m[2][j];
        jacobi3(S, Q);
        Mat3 P;                                                              // (M'M)^(-1/2) = Q diag(lambda^-1/2) Q'
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            P.m[i][j] = Q[i][0] * rsqrt(S[0][0]) * Q[j][0] + Q[i][1] * rsqrt(S[1][1]) * Q[j][1] + Q[i][2] * rsqrt(S[2][2]) * Q[j][2];
        Mat3 R = mat3_mul(M, P);
        const double sg = sgn(mat3_det(R));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.m[i][j] *= sg;
        return R;
    }
    __device__ static inline void crossm(const double* v, Mat3& X) {
        X.m[0][0] = 0; X.m[0][1] = -v[2]; X.m[0][2] = v[1];
        X.m[1][0] = v[2]; X.m[1][1] = 0; X.m[1][2] = -v[0];
        X.m[2][0] = -v[1]; X.m[2][1] = v[0]; X.m[2][2] = 0;
    }
    __device__ static inline void matvec(const Mat3& A, const double* x, double* y) {
        for (int i = 0; i < 3; ++i) y[i] = A.m[i][0] * x[0] + A.m[i][1] * x[1] + A.m[i][2] * x[2];
    }
    // axis-angle of a rotation: vec = null vector of (R - I), angle by atan2   (:73-78)
    __device__ static inline void axis_angle(const Mat3& R, double* out3) {
        Mat3 D = R;
        D.m[0][0] -= 1.0; D.m[1][1] -= 1.0; D.m[2][2] -= 1.0;
        double v[3];
        null3(D, v);
        const double sn = (v[0] * (R.m[2][1] - R.m[1][2]) + v[1] * (R.m[0][2] - R.m[2][0]) + v[2] * (R.m[1][0] - R.m[0][1])) * 0.5;
        const double cs = (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1.0) * 0.5;
        const double o = atan2(sn, cs);
        out3[0] = v[0] * o; out3[1] = v[1] * o; out3[2] = v[2] * o;
    }

    // The serial part of the initial parameters (:56-78): projective fix-up of P2, P3, r = A\a, s = B\b, the three orthogonalised frames and
    // their axis-angle vectors.  One LANE's work -- 58 k cycles of svd3 / jacobi3 / null3 / atan2, an eighth of the block kernel's time when the
    // owner wavefront's lane 0 runs it while 255 threads wait.  k_nordberg_init (gh_wg_kernel.h) runs it for 64 triplets per wavefront, one per
    // lane, ahead of the block kernel, which then only loads the result (`pre`).  P2 / P3: row-major 3 x 4, updated in place when deficient.
    static constexpr bool PREINIT = true;
    static constexpr int PRE_DOUBLES = 64;                                   // P2 12 | P3 12 | U, V, W 27 | axis-angles 9 | deficient 1 | pad
    const double* pre = nullptr;
    __device__ static inline void init_serial(double* P2, double* P3, double* rot, double* p9, int* deficient_out) {
        Mat3 A, B;
        double a[3], b[3];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { A.m[r][c] = P2[4 * r + c]; B.m[r][c] = P3[4 * r + c]; } a[r] = P2[4 * r + 3]; b[r] = P3[4 * r + 3]; }
        // H = eye(4); H(4,1:3) = null(P3(:,1:3))' if rank(P3(:,1:3)) < 3, else the same with P2   (:56-62).
        // rank(X) < 3 <=> sigma_3 <= 3 eps(sigma_1), with sigma_3 = |det X| / (sigma_1 sigma_2) (the squared-matrix
        // Jacobi of svd3 cannot resolve a singular value that small).  P*H = [P(:,1:3) + P(:,4) n', P(:,4)]; P1*H = P1.
        int deficient = 0;
        {
            Mat3 Us, Vs;
            double svB[3], svA[3];
            svd3(B, Us, Vs, svB);
            const bool defB = !(fabs(mat3_det(B)) > 3.0 * eps_of(svB[0]) * svB[0] * svB[1]);
            svd3(A, Us, Vs, svA);
            const bool defA = !(fabs(mat3_det(A)) > 3.0 * eps_of(svA[0]) * svA[0] * svA[1]);
            if (defB || defA) {
                double nv[3];
                null3(defB ? B : A, nv);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { A.m[r][c] += a[r] * nv[c]; B.m[r][c] += b[r] * nv[c]; }
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { P2[4 * r + c] = A.m[r][c]; P3[4 * r + c] = B.m[r][c]; }
                // a null space of dimension > 1 makes H(4,1:3) = null(.)' a MATLAB error: reported
                const double s2 = defB ? svB[1] : svA[1], s1 = defB ? svB[0] : svA[0];
                deficient = !(s2 > 3.0 * eps_of(s1) * s1) ? 1 : 0;
            }
        }
        double r3[3], s3[3];
        matvec(mat3_inv(A), a, r3);                                          // r = A\a   (:65)
        matvec(mat3_inv(B), b, s3);                                          // s = B\b   (:66)
        Mat3 Xr, Xa, Xb, M0;
        double t1[3], t2[3], t3[3];
        crossm(r3, Xr); crossm(a, Xa); crossm(b, Xb);
        // U = [r, crossM(r)^2 s, crossM(r) s]   (:68)
        matvec(Xr, s3, t1); matvec(Xr, t1, t2);
        for (int i = 0; i < 3; ++i) { M0.m[i][0] = r3[i]; M0.m[i][1] = t2[i]; M0.m[i][2] = t1[i]; }
        const Mat3 Um = orthogonalise(M0);
        // V = [a, crossM(a) A s, crossM(a)^2 A s]   (:69)
        matvec(A, s3, t3); matvec(Xa, t3, t1); matvec(Xa, t1, t2);
        for (int i = 0; i < 3; ++i) { M0.m[i][0] = a[i]; M0.m[i][1] = t1[i]; M0.m[i][2] = t2[i]; }
        const Mat3 Vm = orthogonalise(M0);
        // W = [b, crossM(b) B r, crossM(b)^2 B r]   (:70)
        matvec(B, r3, t3); matvec(Xb, t3, t1); matvec(Xb, t1, t2);
        for (int i = 0; i < 3; ++i) { M0.m[i][0] = b[i]; M0.m[i][1] = t1[i]; M0.m[i][2] = t2[i]; }
        const Mat3 Wm = orthogonalise(M0);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { rot[3 * i + j] = Um.m[i][j]; rot[9 + 3 * i + j] = Vm.m[i][j]; rot[18 + 3 * i + j] = Wm.m[i][j]; }
        axis_angle(Um, p9 + 0);                                              // :73-78, :94
        axis_angle(Vm, p9 + 3);
        axis_angle(Wm, p9 + 6);
        *deficient_out = deficient;
    }

    __device__ inline void init(PoseLds* w, GhWork& g) {
        const int lane = lane_id();
        gh_linear_cameras(w);
        double* rot = g.V;                                                   // scratch: U, V, W row-major (27)
        bad = 0;
        if (pre) {                                                           // k_nordberg_init has run
            if (lane < 12) { w->P[0][lane] = pre[lane]; w->P[1][lane] = pre[12 + lane]; }
            if (lane < 27) rot[lane] = pre[24 + lane];
            if (lane < 9) g.p[lane] = pre[51 + lane];
            if (lane == 0) rot[27] = pre[60];
        } else if (lane == 0) {
            int deficient = 0;
            init_serial(w->P[0], w->P[1], rot, g.p, &deficient);
            rot[27] = (double)deficient;
        }
        wave_sync();
        bad = (int)rot[27];
        // Ts = transf_t(T,U,V,W): Ts(a,b,i) = sum V(c,a) U(j,i) T(c,d,j) W(d,b); keep the 10 sparse entries, normalised   (:81-83)
        double ts = 0.0;
        if (lane < 10) {
            int a, b, i;
            sparse_pos(lane, a, b, i);
            for (int j = 0; j < 3; ++j)
                for (int c = 0; c < 3; ++c)
                    for (int d = 0; d < 3; ++d) ts += rot[9 + 3 * c + a] * rot[3 * j + i] * w->t[c + 3 * d + 9 * j] * rot[18 + 3 * d + b];
        }
        const double nn = rsqrt(wave_sum(ts * ts));
        wave_sync();
        if (lane < 10) g.p[9 + lane] = ts * nn;
        wave_sync();
    }

    // a = D' q for q[j + 3k + 9i] = h1[i] gm[j][k] (workgroup kernel).  The ten columns of the tensor parameters are Kronecker products,
    // D[(a,b,i)][9 + k] = V(a,c_k) W(b,d_k) U(i,j_k) (eval below), so their part of D'q is (U'h1)(j_k) * (V' gm W)(c_k,d_k): 73 multiply-adds
    // instead of 270 and 27 LDS reads instead of 135.  The nine rotation columns stay dense (243).
    static constexpr bool KRONECKER_DT = true;
    __device__ __forceinline__ void apply_Dt_kronecker(const GhWork& g, const double (&h1)[3], const double (&gm)[3][3], double (&a)[U]) const {
        const double* rot = g.V;                                             // U, V, W (row-major), written by eval()
#pragma unroll
        for (int c = 0; c < 9; ++c) a[c] = 0.0;
#pragma unroll
        for (int i1 = 0; i1 < 3; ++i1)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double q = h1[i1] * gm[j][k];
                    const double* row = g.D + (j + 3 * k + 9 * i1) * U;
#pragma unroll
                    for (int c = 0; c < 9; ++c) a[c] += row[c] * q;
                }
        double uh[3], t[3][3], Gp[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) uh[j] = rot[j] * h1[0] + rot[3 + j] * h1[1] + rot[6 + j] * h1[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) t[r][d] = gm[r][0] * rot[18 + d] + gm[r][1] * rot[18 + 3 + d] + gm[r][2] * rot[18 + 6 + d];   // gm W
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) Gp[c][d] = rot[9 + c] * t[0][d] + rot[9 + 3 + c] * t[1][d] + rot[9 + 6 + c] * t[2][d];        // V' (gm W)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            int c, d, j;
            sparse_pos(k, c, d, j);
            a[9 + k] = uh[j] * Gp[c][d];
        }
    }

    __device__ inline void eval(GhWork& g) const {
        const int lane = lane_id();
        constexpr int u = U, n = U + C, ld = n + 1;
        double* rot = g.V;                                                   // R_k (3 x 9) then dR_k/dx_m (3 x 3 x 9), row-major 3x3 each
        for (int e = lane; e < n * ld; e += WAVE) g.M[e] = 0.0;
        if (lane < 3) {                                                      // Rodrigues and its derivative   (:131-136, :181-198)
            const double* x = g.p + 3 * lane;
            const double o = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const double v[3] = {x[0] / o, x[1] / o, x[2] / o};
            const double so = sin(o), co = cos(o);
            Mat3 X, X2;
            crossm(v, X);
            X2 = mat3_mul(X, X);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rot[9 * lane + 3 * r + c] = ((r == c) ? 1.0 : 0.0) + so * X.m[r][c] + (1.0 - co) * X2.m[r][c];
            for (int i = 0; i < 3; ++i) {
                double ei[3] = {0, 0, 0};
                ei[i] = 1.0;
                Mat3 Xe;
                crossm(ei, Xe);
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
                    const double vv = v[r] * v[c];
                    rot[27 + 27 * lane + 9 * i + 3 * r + c] =
                        -v[i] * so * ((r == c) ? 1.0 : 0.0) + v[i] * co * X.m[r][c] + so * (1.0 / o) * (Xe.m[r][c] - v[i] * X.m[r][c])
                        + v[i] * so * vv + (1.0 - co) * (1.0 / o) * (v[r] * ei[c] + ei[r] * v[c] - 2.0 * v[i] * vv);
                }
            }
        }
        wave_sync();
        const double* Ur = rot; const double* Vr = rot + 9; const double* Wr = rot + 18;
        // T = transf_t(Ts, U', V', W'): T(a,b,i) = sum_k V(a,c_k) W(b,d_k) U(i,j_k) Ts_k   (:145)
        for (int e = lane; e < 27 * (u + 1); e += WAVE) {
            const int r = e % 27, col = e / 27;                              // col 0..18: J columns; col 19: T itself
            const int i = r / 9, b = (r % 9) / 3, a = r % 3;
            double acc = 0.0;
            if (col >= 9 && col < 19) {                                      // d/dparamT_k   (:176-179)
                int c, d, j;
                sparse_pos(col - 9, c, d, j);
                acc = Vr[3 * a + c] * Wr[3 * b + d] * Ur[3 * i + j];
            } else {
                const int which = (col < 9) ? col / 3 : -1, m = col % 3;     // derivative w.r.t. rotation `which`, component m   (:199-203)
                const double* dR = rot + 27 + 27 * ((which < 0) ? 0 : which) + 9 * m;
                for (int k = 0; k < 10; ++k) {
                    int c, d, j;
                    sparse_pos(k, c, d, j);
                    const double fu = (which == 0) ? dR[3 * i + j] : Ur[3 * i + j];
                    const double fv = (which == 1) ? dR[3 * a + c] : Vr[3 * a + c];
                    const double fw = (which == 2) ? dR[3 * b + d] : Wr[3 * b + d];
                    acc += fv * fw * fu * g.p[9 + k];
                }
            }
            if (col < 19) g.D[r * u + col] = acc; else g.Tc[r] = acc;
        }
        // g = |paramT|^2 - 1, C(1,10:19) = 2 paramT'   (:208-210)
        const double pk = (lane < 10) ? g.p[9 + lane] : 0.0;
        const double g0 = wave_sum(pk * pk) - 1.0;
        if (lane < 10) { g.M[u * ld + 9 + lane] = 2.0 * pk; g.M[(9 + lane) * ld + u] = 2.0 * pk; }
        if (lane == 0) g.M[u * ld + n] = -g0;
        wave_sync();
    }
};

// ---- Ressl's minimal parameterisation (ResslTFTPoseEstimation.m:56-68,79,110-135,164-170) ----
struct ResslModel {
    int Ind;                                                                 // argmax |e21|, 0-based
    static constexpr int U = 20, C = 2;
    static constexpr int WG_PER_CU = 2;                                      // 256 registers: at three the factored weight pass spills (3.5 vs 3.0 ms per 10k x 200)
    static constexpr bool IDENTITY_D = false;
    static constexpr bool SPARSE_DT = true;
    // workgroup kernel: the wavefront that ran init() publishes the per-triplet model state the other wavefronts need (apply_Dt)
    __device__ __forceinline__ void share(double* slot) const { slot[0] = (double)Ind; }
    __device__ __forceinline__ void adopt(const double* slot) { Ind = (int)slot[0]; }
    // a = D' q for q[j + 3k + 9i] = h1[i] gm[j][k], from the structure of D (eval below): T(j,k,i) = e21(j) S(k,i) + mn(i,j) e31(k)
    __device__ __forceinline__ void apply_Dt(const GhWork& g, const double (&h1)[3], const double (&gm)[3][3], double (&a)[U]) const {
        const int i2a = (Ind == 0) ? 1 : 0, i2b = (Ind == 2) ? 1 : 2;
        double e21[3], S[3][3], e31[3], mn[3][3];                           // wave-uniform parameters (scalar registers)
#pragma unroll
        for (int j = 0; j < 3; ++j) e21[j] = (j == Ind) ? 1.0 : wave_uniform(g.p[(j == i2a) ? 9 : 10]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            e31[k] = wave_uniform(g.p[17 + k]);
#pragma unroll
            for (int i = 0; i < 3; ++i) S[k][i] = wave_uniform(g.p[k + 3 * i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) mn[i][j] = (j == Ind) ? 0.0 : wave_uniform(g.p[11 + i + 3 * ((j == i2a) ? 0 : 1)]);
        double bj[3] = {0.0, 0.0, 0.0}, cij[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) a[17 + k] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double sk = 0.0;                                             // d/dS(k,i): sum_j e21(j) q(j,k,i)
#pragma unroll
                for (int j = 0; j < 3; ++j) sk += e21[j] * gm[j][k];
                a[k + 3 * i] = h1[i] * sk;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double sS = 0.0, sE = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) { sS += S[k][i] * gm[j][k]; sE += e31[k] * gm[j][k]; }
                bj[j] += h1[i] * sS;                                         // d/de21(j): sum_{k,i} S(k,i) q(j,k,i)
                cij[i][j] = h1[i] * sE;                                      // d/dmn(i,j): sum_k e31(k) q(j,k,i)
#pragma unroll
                for (int k = 0; k < 3; ++k) a[17 + k] += mn[i][j] * h1[i] * gm[j][k];   // d/de31(k): sum_{i,j} mn(i,j) q(j,k,i)
            }
        }
        a[9] = (i2a == 0) ? bj[0] : bj[1];
        a[10] = (i2b == 1) ? bj[1] : bj[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[11 + i] = (i2a == 0) ? cij[i][0] : cij[i][1];
            a[14 + i] = (i2b == 1) ? cij[i][1] : cij[i][2];
        }
    }
    static constexpr bool REDUNDANT_CONSTRAINTS = false;
    // initial parameters from linearTFT's output (w->t constrained tensor, w->epi)
    __device__ inline void init(PoseLds* w, GhWork& g) {
        const int lane = lane_id();
        const double* e21 = w->epi; const double* e31 = w->epi + 3;
        const double a0 = fabs(e21[0]), a1 = fabs(e21[1]), a2 = fabs(e21[2]);
        Ind = (a0 >= a1 && a0 >= a2) ? 0 : ((a1 >= a2) ? 1 : 2);              // first maximum, as MATLAB max   (:57)
        const double escale = 1.0 / e21[Ind];
        const double n31 = rsqrt(e31[0] * e31[0] + e31[1] * e31[1] + e31[2] * e31[2]);
        const int i2a = (Ind == 0) ? 1 : 0, i2b = (Ind == 2) ? 1 : 2;         // Ind2
        // S(k,i) = T(Ind,k,i); aux = |S|_F; S /= aux; T /= aux   (:60-62)
        double sv = (lane < 9) ? w->t[Ind + 3 * (lane % 3) + 9 * (lane / 3)] : 0.0;
        const double aux = rsqrt(wave_sum(sv * sv));
        sv *= aux;
        if (lane < 9) g.p[lane] = sv;                                        // S(:) at k + 3i
        if (lane == 0) { g.p[9] = e21[i2a] * escale; g.p[10] = e21[i2b] * escale; }
        if (lane < 3) g.p[17 + lane] = e31[lane] * n31;
        wave_sync();
        if (lane < 6) {                                                      // mn(i, Ind2(m)) = e31' (T_i' - S(:,i) e21')(:, Ind2(m))   (:65-68)
            const int i = lane % 3, m = lane / 3, j = (m == 0) ? i2a : i2b;
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += e31[k] * n31 * (w->t[j + 3 * k + 9 * i] * aux - g.p[k + 3 * i] * e21[j] * escale);
            g.p[11 + i + 3 * m] = acc;
        }
        wave_sync();
        gh_linear_cameras(w);
    }
    // p -> Tc, D, and the constraint rows / right-hand side of the KKT matrix
    __device__ inline void eval(GhWork& g) const {
        const int lane = lane_id();
        const int u = U, n = U + C, ld = n + 1;
        const int i2a = (Ind == 0) ? 1 : 0, i2b = (Ind == 2) ? 1 : 2;
        for (int e = lane; e < 27 * u; e += WAVE) g.D[e] = 0.0;
        for (int e = lane; e < n * ld; e += WAVE) g.M[e] = 0.0;
        wave_sync();
        if (lane < 27) {
            const int i = lane / 9, k = (lane % 9) / 3, j = lane % 3;
            const double e21j = (j == Ind) ? 1.0 : g.p[(j == i2a) ? 9 : 10];
            const double mnij = (j == Ind) ? 0.0 : g.p[11 + i + 3 * ((j == i2a) ? 0 : 1)];
            const double S_ki = g.p[k + 3 * i], e31k = g.p[17 + k];
            g.Tc[lane] = e21j * S_ki + mnij * e31k;                          // T(j,k,i)   (:118-120)
            double* Dr = g.D + lane * u;
            Dr[k + 3 * i] = e21j;                                            // d/dS(k,i)
            if (j != Ind) { const int m = (j == i2a) ? 0 : 1; Dr[9 + m] = S_ki; Dr[11 + i + 3 * m] = e31k; }
            Dr[17 + k] = mnij;                                               // d/de31(k)
        }
        // g = [|e31|^2 - 1; |S|^2 - 1], C rows 2 e31', 2 S(:)'   (:129-135)
        double s31 = (lane < 3) ? g.p[17 + lane] : 0.0, sS = (lane < 9) ? g.p[lane] : 0.0;
        const double g0 = wave_sum(s31 * s31) - 1.0, g1 = wave_sum(sS * sS) - 1.0;
        if (lane < 3) { g.M[u * ld + 17 + lane] = 2.0 * s31; g.M[(17 + lane) * ld + u] = 2.0 * s31; }
        if (lane < 9) { g.M[(u + 1) * ld + lane] = 2.0 * sS; g.M[lane * ld + u + 1] = 2.0 * sS; }
        if (lane == 0) { g.M[u * ld + n] = -g0; g.M[(u + 1) * ld + n] = -g1; }
        wave_sync();
    }
};

// inverse of a symmetric positive definite E x E block through its Cholesky factor (per lane), packed lower triangle out.
// false when a pivot is not positive.
template <int E>
__device__ __forceinline__ bool spd_inverse_packed(const double (&W)[E][E], double* Wp) {
    double L[E][E], id[E];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        double d = W[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        ok = ok && (d > 0.0);
        id[j] = rsqrt(d);
#pragma unroll
        for (int i = j + 1; i < E; ++i) {
            double s = W[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * id[j];
        }
    }
    double Li[E][E];                                                         // inv(L), lower triangular
#pragma unroll
    for (int j = 0; j < E; ++j) {
        Li[j][j] = id[j];
#pragma unroll
        for (int i = j + 1; i < E; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s += L[i][k] * Li[k][j];
            Li[i][j] = -s * id[i];
        }
    }
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {                                       // inv(W) = inv(L)' inv(L)
            double s = 0.0;
#pragma unroll
            for (int k = a; k < E; ++k) s += Li[k][a] * Li[k][b];
            Wp[a * (a + 1) / 2 + b] = s;
        }
    return ok;
}

// pinv(W) under MATLAB's tolerance tolW for a positive definite block W (= B B' + 1e-12 I ALREADY: the callers pass the shifted
// matrix) whose ONLY eigenvalue at or below the tolerance is its shifted null direction n -- the generic case of the rank-(E-1)
// weight blocks once E N eps(|W|) exceeds 1e-12.  With lam = n'Wn that eigenvalue and Wn = W + n n' (well conditioned, so its
// Cholesky inverse is accurate)
//     pinv = Wn^-1 - n n' / (1 + lam)
// exactly, at a fifth of the cost of the Jacobi eigen-decomposition.  Returns false when the structure does not hold (no eigenvalue
// under the tolerance, a second one, or a failed factorisation): the caller then takes the eigen-decomposition.
template <int E>
__device__ __forceinline__ bool pinv_one_null_packed(const double (&W)[E][E], const double tolW, double* Wp) {
    double n[E];
    spd_min_eigvec<E>(W, n, 40);
    double lam = 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a) {
        double wn = 0.0;
#pragma unroll
        for (int c = 0; c < E; ++c) wn += W[a][c] * n[c];
        lam += n[a] * wn;
    }
    if (!(lam <= tolW)) return false;                                        // nothing is truncated here (or NaN)
    double Wn[E][E];
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int c = 0; c < E; ++c) Wn[a][c] = W[a][c] + n[a] * n[c];
    if (!spd_inverse_packed<E>(Wn, Wp)) return false;
    const double kn = 1.0 / (1.0 + lam);
    double fro2 = 0.0;
#pragma unroll
    for (int a = 0; a < E; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            const double v = Wp[a * (a + 1) / 2 + c] - n[a] * n[c] * kn;
            Wp[a * (a + 1) / 2 + c] = v;
            fro2 += (a == c) ? v * v : 2.0 * v * v;
        }
    return fro2 * tolW * tolW < 1.0;                                         // second-smallest eigenvalue >= 1 / |pinv|_F > tolW
}

// pinv(B B' + 1e-12 I) (Gauss_Helmert.m:52,57) of one 4 x 4 trilinearity block, evaluated at the accuracy of the FORMULA instead of
// the accuracy of an fp64 B B'.  A point triplet carries three independent constraints, so B B' has one eigenvalue mu that is zero
// for consistent observations and ~1e-12 .. 1e-9 (the squared inconsistency) during the iteration -- the size of the 1e-12 shift.
// pinv gives that direction the weight 1 / (mu + 1e-12) ~ 1e9 .. 1e12, and an fp64 B B' knows mu only to ~1e-16 absolute: the
// weights of a plain Cholesky / eigen / LAPACK evaluation carry 1e-4 relative noise (the reference's own dense pinv included), which is
// what made the Gauss-Helmert results agree to 1e-6 .. 1e-4 only.  Here, with n the eigenvector of the smallest eigenvalue of W = B B' + 1e-12 I:
//   mu = |B' n|^2                      the small eigenvalue WITHOUT the cancellation (relative error ~1e-9)
//   K  = (W + n n')^-1                 Cholesky of a well conditioned matrix: eigenvalues lambda_k + 1e-12 and 1 + mu + 1e-12
//   pinv = K + n n' (1/(mu + 1e-12) - 1/(1 + mu + 1e-12))      if mu + 1e-12 > tolW (kept),     K - n n' / (1 + mu + 1e-12) otherwise (truncated)
// agrees with a 50-digit evaluation of the reference's formulas to ~1e-9 (profiles/r2_gh_noise_mp.txt).
// Returns false when the block does not have that structure (second-smallest eigenvalue not far above mu, failed factorisation,
// non-finite data): the caller takes the eigen-decomposition path.  Wp: packed lower triangle, WITHOUT the trailing + 1e-12 I.
// FACTORED = false: Wp is the full pseudo-inverse.  FACTORED = true: Wp is its REGULAR part only, K - n n' / (1 + mu + 1e-12) (weights
// 1 / (lambda_k + 1e-12) of the three ordinary directions), and the strong direction comes back separately as (n, *cs) with
// cs = 1 / (mu + 1e-12) (kept) or 0 (truncated): pinv = Wp + cs n n'.  The caller then evaluates A' pinv A = A' Wp A + cs (A'n)(A'n)'
// with a = A'n formed FIRST: for the minimal parameterisations n is (nearly) orthogonal to the columns of A -- |a| is the
// inconsistency of the correspondence, 1e-15 .. 1e-6 -- so the product with the 1e12-size matrix entries cancels ~10 digits
// (the noise of every fp64 evaluation that forms pinv(W) explicitly, the reference's included), while cs a a' is accurate to ~1e-9.
template <bool FACTORED>
__device__ __forceinline__ bool pinv_block_deflated(const double (&B)[4][6], const double (&W)[4][4], const double tolW, double (&Wp)[10],
                                                    double (&n)[4], double* cs) {
    bool conv;
    spd_min_eigvec<4>(W, n, 40, &conv);
    double mu = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double bn = B[0][k] * n[0] + B[1][k] * n[1] + B[2][k] * n[2] + B[3][k] * n[3];
        mu += bn * bn;
    }
    double Wn[4][4], tr = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        tr += W[a][a];
#pragma unroll
        for (int c = 0; c < 4; ++c) Wn[a][c] = W[a][c] + n[a] * n[c];
    }
    if (!conv || !spd_inverse_packed<4>(Wn, Wp)) return false;
    const double lam = mu + 1e-12, kn = 1.0 / (1.0 + lam);
    double fro2 = 0.0;                                                       // |K - n n' / (1 + lam)|_F^2 = sum_k 1 / (lambda_k + 1e-12)^2
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            const double v = Wp[a * (a + 1) / 2 + c] - n[a] * n[c] * kn;
            Wp[a * (a + 1) / 2 + c] = v;
            fro2 += (a == c) ? v * v : 2.0 * v * v;
        }
    // second-smallest eigenvalue lambda_3 >= 1 / |.|_F: require lambda_3 > 1e-5 trace (n is then determined to ~1e-11) and > 1e3 mu
    const double l3min = 1e-5 * tr + 1e3 * mu;
    if (!(fro2 * l3min * l3min < 1.0)) return false;
    const double c1 = (lam > tolW) ? 1.0 / lam : 0.0;
    *cs = c1;
    if (!FACTORED) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c <= a; ++c) Wp[a * (a + 1) / 2 + c] += c1 * n[a] * n[c];
    }
    return true;
}

// gradient of n' f with respect to the 27 tensor entries, q = Ap' n (Ap: ResslTFTPoseEstimation.m:156), for the observation estimate o:
// f = tril_quad(m), m = x1 T1 + y1 T2 + T3, so q[j + 3k + 9i] = h1[i] gm[j][k], h1 = (x1, y1, 1).
__device__ __forceinline__ void tril_grad_n(const double (&o)[6], const double (&n)[4], double (&gm)[3][3]) {
    const double x2 = o[2], y2 = o[3], x3 = o[4], y3 = o[5];
    const double a0[3] = {-n[1], -n[0], n[0] * y2 + n[1] * x2};              // d(n'f)/d v0[j]
    const double a1[3] = {-n[3], -n[2], n[2] * y2 + n[3] * x2};              // d(n'f)/d v1[j]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gm[j][0] = -a1[j];
        gm[j][1] = -a0[j];
        gm[j][2] = y3 * a0[j] + x3 * a1[j];
    }
}

// w = -f - B (x - xi) (Gauss_Helmert.m:58); stores W+ (10) and W+ w (4) of correspondence i
__device__ __forceinline__ void gh_store_point(const GhWork& g, const PoseLds* w, const double* pts, int i, const double (&o)[6],
                                               const double (&f)[4], const double (&B)[4][6], const double (&Wp)[10]) {
    const Pt6 x = premap(load_pt(pts, i), w->nrm);
    double wv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = -f[a];
#pragma unroll
        for (int k = 0; k < 6; ++k) s -= B[a][k] * (x.v[k] - o[k]);
        wv[a] = s;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) g.pp[14 * i + k] = Wp[k];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        g.pp[14 * i + 10 + a] = wp_at(Wp, a, 0) * wv[0] + wp_at(Wp, a, 1) * wv[1] + wp_at(Wp, a, 2) * wv[2] + wp_at(Wp, a, 3) * wv[3];
}

// ---- factored strong-direction terms (see pinv_block_deflated<true>) -------------------------------------------------------------
// Per correspondence the thread holds b = sqrt(cs) a (U doubles, a = D' Ap' n) and t = sqrt(cs) n'w in registers; the sums
//   N_s[r][c] = sum_i b_i[r] b_i[c]  (lower triangle, U (U + 1) / 2 entries)   and   r_s[r] = sum_i b_i[r] t_i  (U entries)
// are taken 32 at a time with the halving butterfly (compile-time entry indices, so b stays in registers) and added to the calling
// wavefront's own partial-sum slot.
template <int U, int E>
__device__ __forceinline__ double strong_entry(const double (&b)[U], const double t) {
    constexpr int ntri = U * (U + 1) / 2;
    if constexpr (E < ntri) return b[tri_row_of(E)] * b[tri_col_of(E)];
    else if constexpr (E < ntri + U) return b[E - ntri] * t;
    else return 0.0;
}
template <int U, int SW>
__device__ __forceinline__ void strong_sweep(const double (&b)[U], const double t, double* slot) {
    constexpr int total = U * (U + 1) / 2 + U;
    if constexpr (32 * SW < total) {
        double acc[32];
        acc[0] = strong_entry<U, 32 * SW + 0>(b, t);   acc[1] = strong_entry<U, 32 * SW + 1>(b, t);   acc[2] = strong_entry<U, 32 * SW + 2>(b, t);   acc[3] = strong_entry<U, 32 * SW + 3>(b, t);
        acc[4] = strong_entry<U, 32 * SW + 4>(b, t);   acc[5] = strong_entry<U, 32 * SW + 5>(b, t);   acc[6] = strong_entry<U, 32 * SW + 6>(b, t);   acc[7] = strong_entry<U, 32 * SW + 7>(b, t);
        acc[8] = strong_entry<U, 32 * SW + 8>(b, t);   acc[9] = strong_entry<U, 32 * SW + 9>(b, t);   acc[10] = strong_entry<U, 32 * SW + 10>(b, t); acc[11] = strong_entry<U, 32 * SW + 11>(b, t);
        acc[12] = strong_entry<U, 32 * SW + 12>(b, t); acc[13] = strong_entry<U, 32 * SW + 13>(b, t); acc[14] = strong_entry<U, 32 * SW + 14>(b, t); acc[15] = strong_entry<U, 32 * SW + 15>(b, t);
        acc[16] = strong_entry<U, 32 * SW + 16>(b, t); acc[17] = strong_entry<U, 32 * SW + 17>(b, t); acc[18] = strong_entry<U, 32 * SW + 18>(b, t); acc[19] = strong_entry<U, 32 * SW + 19>(b, t);
        acc[20] = strong_entry<U, 32 * SW + 20>(b, t); acc[21] = strong_entry<U, 32 * SW + 21>(b, t); acc[22] = strong_entry<U, 32 * SW + 22>(b, t); acc[23] = strong_entry<U, 32 * SW + 23>(b, t);
        acc[24] = strong_entry<U, 32 * SW + 24>(b, t); acc[25] = strong_entry<U, 32 * SW + 25>(b, t); acc[26] = strong_entry<U, 32 * SW + 26>(b, t); acc[27] = strong_entry<U, 32 * SW + 27>(b, t);
        acc[28] = strong_entry<U, 32 * SW + 28>(b, t); acc[29] = strong_entry<U, 32 * SW + 29>(b, t); acc[30] = strong_entry<U, 32 * SW + 30>(b, t); acc[31] = strong_entry<U, 32 * SW + 31>(b, t);
        const double tot = wave_reduce_scatter<32>(acc);
        const int lane = lane_id(), e = 32 * SW + reduce32_index(lane);
        if ((lane & 1) == 0 && e < total) slot[e] += tot;
        sched_fence();                                                       // keep the sweeps apart: interleaved, their 32 accumulators each would spill
    }
}
template <int U>
__device__ inline void strong_accumulate(const double (&b)[U], const double t, double* slot) {
    static_assert(U * (U + 1) / 2 + U <= 13 * 32, "thirteen sweeps");
    strong_sweep<U, 0>(b, t, slot); strong_sweep<U, 1>(b, t, slot); strong_sweep<U, 2>(b, t, slot); strong_sweep<U, 3>(b, t, slot);
    strong_sweep<U, 4>(b, t, slot); strong_sweep<U, 5>(b, t, slot); strong_sweep<U, 6>(b, t, slot); strong_sweep<U, 7>(b, t, slot);
    strong_sweep<U, 8>(b, t, slot); strong_sweep<U, 9>(b, t, slot); strong_sweep<U, 10>(b, t, slot); strong_sweep<U, 11>(b, t, slot);
    strong_sweep<U, 12>(b, t, slot);
}
// The same sums on the matrix core (round 5).  They are a genuine dense contraction: with e_i = [b_i; t_i] (U + 1 entries per correspondence)
//   [N_s  r_s; r_s'  .] = sum_i e_i e_i' = E' E,   E = [e_1' ; ... ; e_N']   (N x (U + 1))
// -- the Gram matrix of a tall matrix, K = N.  The butterflies above cost 32 products + 93 exchange / add instructions per 32 sums and
// correspondence trip (8 sweeps for U = 20, 13 for U = 27: 1.0 k / 1.7 k VALU instructions per trip and wavefront, a fifth to a third of a
// Gauss-Helmert iteration); v_mfma_f64_16x16x4_f64 takes four correspondences and a 16 x 16 tile per instruction: three tiles ((0,0), (1,0), (1,1):
// U + 1 <= 32) x 16 k-steps = 48 instructions per trip of 64 correspondences, accumulators in registers across the trips.
// The operands want the TRANSPOSED layout (lane = component, k = correspondence), so the wavefront passes its vectors through a scratch area of
// its own in LDS, CH correspondences at a time (CH x LDE doubles, LDE odd: rows of 16 lanes read consecutive doubles, the four k-rows land on
// different banks).  Summation order differs from the butterflies' (k order inside the matrix core, then over the steps): same accuracy, other
// rounding.  add(): every lane of the wavefront calls it (a lane without a correspondence passes zeros); store(): tiles -> the slot layout of
// strong_accumulate (lower triangle, then the U entries of r_s).
template <int U, int CH>
struct StrongGram {
    static_assert(U + 1 > 16 && U + 1 <= 32 && (CH == 16 || CH == 32 || CH == 64), "two blocks of 16 components");
    static constexpr int UE = U + 1, LDE = UE | 1, SCRATCH = CH * LDE;
    double t00[4], t10[4], t11[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int v = 0; v < 4; ++v) { t00[v] = 0.0; t10[v] = 0.0; t11[v] = 0.0; }
    }
    __device__ __forceinline__ void add(const double (&b)[U], const double t, double* scratch) {
        const int lane = lane_id();
#pragma unroll 1
        for (int c0 = 0; c0 < WAVE; c0 += CH) {
            wave_sync();                                                     // (the previous chunk's reads are done)
            if (lane >= c0 && lane < c0 + CH) {
                double* row = scratch + (lane - c0) * LDE;
#pragma unroll
                for (int k = 0; k < U; ++k) row[k] = b[k];
                row[U] = t;
            }
            wave_sync();
            const double* src = scratch + (lane >> 4) * LDE + (lane & 15);
            const bool hi = 16 + (lane & 15) < UE;
#pragma unroll 1
            for (int s = 0; s < CH / 4; ++s) {                               // k-step s: correspondences 4 s .. 4 s + 3 of the chunk
                const double a0 = src[4 * s * LDE];
                const double a1 = hi ? src[4 * s * LDE + 16] : 0.0;
                mfma_f64_16x16x4(a0, a0, t00);
                mfma_f64_16x16x4(a1, a0, t10);
                mfma_f64_16x16x4(a1, a1, t11);
            }
        }
    }
    __device__ __forceinline__ void store(double* slot) const {
        constexpr int ntri = U * (U + 1) / 2;
        const int lane = lane_id(), col = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = rg + 4 * v;
            if (col <= row) slot[tri_index(row, col)] = t00[v];              // rows / columns 0 .. 15
            const int r1 = 16 + row, c1 = 16 + col;
            if (r1 < U) slot[tri_index(r1, col)] = t10[v];
            else if (r1 == U) slot[ntri + col] = t10[v];                     // the t row: r_s[col]
            if (r1 < U && c1 <= r1) slot[tri_index(r1, c1)] = t11[v];
            else if (r1 == U && c1 < U) slot[ntri + c1] = t11[v];
        }
    }
};

// a = D' q for q = h1 (x) vec(gm) (27): through the model's sparse form when it has one, else the dense 27 x U matrix in LDS
template <class M, class = void> struct gh_has_kronecker_dt { static constexpr bool value = false; };
template <class M> struct gh_has_kronecker_dt<M, decltype((void)M::KRONECKER_DT)> { static constexpr bool value = M::KRONECKER_DT; };
// ROLLED: a loop over the columns of D with a select chain to place each sum -- twice the instructions, a tenth of the code and no
// register spills; for the fused single-wavefront kernel, where the unrolled form spilled 100 registers (6.2 ms per 10 k Nordberg triplets).
template <class Model, bool ROLLED = false>
__device__ __forceinline__ void strong_apply_Dt(const Model& model, const GhWork& g, const double (&h1)[3], const double (&gm)[3][3], double (&a)[Model::U]) {
    if constexpr (Model::SPARSE_DT) {
        model.apply_Dt(g, h1, gm, a);
    } else if constexpr (!ROLLED && gh_has_kronecker_dt<Model>::value) {
        model.apply_Dt_kronecker(g, h1, gm, a);
    } else if constexpr (ROLLED) {
        constexpr int u = Model::U;
#pragma unroll 1
        for (int pcol = 0; pcol < u; ++pcol) {
            double acc = 0.0;
#pragma unroll
            for (int i1 = 0; i1 < 3; ++i1) {
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) part += g.D[(j + 3 * k + 9 * i1) * u + pcol] * gm[j][k];
                acc += h1[i1] * part;
            }
#pragma unroll
            for (int c = 0; c < u; ++c) a[c] = (c == pcol) ? acc : a[c];
        }
    } else {
        constexpr int u = Model::U;
        // row by row of D (every lane reads the same addresses: LDS broadcasts, consecutive entries pair up in ds_read2_b64); a[] is
        // indexed statically -- a loop over the columns instead needs a 2 u-instruction select chain per column to place its sum
#pragma unroll
        for (int c = 0; c < u; ++c) a[c] = 0.0;
#pragma unroll
        for (int i1 = 0; i1 < 3; ++i1)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double q = h1[i1] * gm[j][k];
                    const double* row = g.D + (j + 3 * k + 9 * i1) * u;
#pragma unroll
                    for (int c = 0; c < u; ++c) a[c] += row[c] * q;
                }
    }
}

// Gauss_Helmert.m:38-83 for a trilinearity model.  xi holds x0 on entry.  Returns iterations; status via *st.
template <class Model>
__device__ inline int gauss_helmert_wave(PoseLds* w, GhWork& g, Model& model, const double* pts, int N, int* st, double* dbg, bool exact_pinv) {
    const int lane = lane_id();
    const int u = g.u, c = g.c, n = u + c, ld = n + 1;
    // objFunc = v0' v0, v0 = x0 - x   (:45-46)
    double objFunc = 0.0;
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = g.xi[6 * i + k] - x.v[k]; objFunc += d * d; }
    }
    objFunc = wave_sum(objFunc);
    int it = 0;
#pragma unroll 1
    for (it = 1; it <= GH_IT_MAX; ++it) {
        if (it == 1) phase_stamp(dbg, 40);
        model.eval(g);                                                       // func(xi, ti, yi)   (:50)
        if (it == 1) phase_stamp(dbg, 41);
        double T[27];
        load_uniform27(g.Tc, T);
        // ---- W = B B' (:52): finite check and a bound on its largest eigenvalue (lambda_max <= |W|_F) ----
        double f2max = 0.0;
        bool finite = true;
        for (int i = lane; i < N; i += WAVE) {
            double o[6], f[4], B[4][6], W[4][4];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            tril_block(T, o, f, B);
            block_W(B, W);
            double chk = 0.0, fro2 = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { chk += W[a][b]; fro2 += W[a][b] * W[a][b]; }
            finite = finite && (fabs(chk) <= 1.79e308);
            f2max = (fro2 > f2max) ? fro2 : f2max;
        }
        f2max = wave_max(f2max);
        if (wave_any(!finite) || !(f2max <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :53-55
        // pinv(W + 1e-12 I) (:57) truncates singular values <= 4N eps(lambda_max).  Every eigenvalue of W + 1e-12 I is
        // >= 1e-12, so while that tolerance is safely below 1e-12 (N up to a few hundred) nothing can be truncated and the
        // tolerance itself is not needed; otherwise the per-block eigenvalue pass finds lambda_max.  The blocks themselves are
        // inverted in the deflated form (pinv_block_deflated); the Jacobi eigen-decompositions remain as the fall-back for
        // blocks without the one-small-eigenvalue structure and as TFF_OPT_GH_EXACT.
        const bool may_truncate = !(4.0 * (double)N * eps_of(sqrt(f2max)) < 0.9e-12);
        if (it == 1) phase_stamp(dbg, 42);
        double tolW = 0.0;
        if (may_truncate || exact_pinv) {
            double smax = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                tril_block(T, o, f, B);
                block_W(B, W);
                jacobi4<false>(W, V);
#pragma unroll
                for (int a = 0; a < 4; ++a) smax = (fabs(W[a][a]) > smax) ? fabs(W[a][a]) : smax;
            }
            smax = wave_max(smax);
            tolW = 4.0 * (double)N * eps_of(smax);
        }
        bool jacobi = exact_pinv;
        // Minimal parameterisations (Ressl, Nordberg): the factored strong direction of gh_wg_kernel.h -- pp holds the regular part of
        // W+ only, the sums of cs a a' and cs a n'w (a = D' Ap' n formed first) go to S = g.V (dead until a pseudo-inverse fall-back).
        bool factored = false;
        if (!jacobi) {
            constexpr bool want_factored = !Model::IDENTITY_D;
            constexpr int NS = Model::U * (Model::U + 1) / 2 + Model::U;
            if (want_factored) for (int e = lane; e < NS; e += WAVE) g.V[e] = 0.0;
            bool bad = false;
#pragma unroll 1
            for (int base = 0; base < N; base += WAVE) {                     // wave-uniform trip count (the butterflies need the whole wavefront)
                const int i = base + lane;
                double bv[Model::U], tv = 0.0;
#pragma unroll
                for (int k = 0; k < Model::U; ++k) bv[k] = 0.0;
                if (i < N) {
                    double o[6], f[4], B[4][6], W[4][4], Wp[10];
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                    tril_block(T, o, f, B);
                    block_W(B, W);
                    double nn[4], cs = 0.0;
                    const bool ok = want_factored ? pinv_block_deflated<true>(B, W, tolW, Wp, nn, &cs) : pinv_block_deflated<false>(B, W, tolW, Wp, nn, &cs);
                    bad = !ok || bad;
#pragma unroll
                    for (int a = 0; a < 4; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
                    gh_store_point(g, w, pts, i, o, f, B, Wp);
                    if constexpr (!Model::IDENTITY_D) if (ok) {
                        double gm[3][3];
                        tril_grad_n(o, nn, gm);
                        const double h1[3] = {o[0], o[1], 1.0};
                        strong_apply_Dt<Model, true>(model, g, h1, gm, bv);
                        const double sc = sqrt(cs);
#pragma unroll
                        for (int k = 0; k < Model::U; ++k) bv[k] *= sc;
                        const Pt6 x = premap(load_pt(pts, i), w->nrm);
                        double nw = -(nn[0] * f[0] + nn[1] * f[1] + nn[2] * f[2] + nn[3] * f[3]);
#pragma unroll
                        for (int k = 0; k < 6; ++k) nw -= (B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3]) * (x.v[k] - o[k]);
                        tv = sc * nw;
                    }
                }
                if constexpr (!Model::IDENTITY_D) strong_accumulate<Model::U>(bv, tv, g.V);
            }
            if (wave_any(bad)) jacobi = true;                                // a block without the structure: eigen-decompositions for all
            else factored = want_factored;
        }
        if (jacobi) {
            if (!(may_truncate || exact_pinv)) {                             // the tolerance was not needed so far
                double smax = 0.0;
                for (int i = lane; i < N; i += WAVE) {
                    double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                    tril_block(T, o, f, B);
                    block_W(B, W);
                    jacobi4<false>(W, V);
#pragma unroll
                    for (int a = 0; a < 4; ++a) smax = (fabs(W[a][a]) > smax) ? fabs(W[a][a]) : smax;
                }
                smax = wave_max(smax);
                tolW = 4.0 * (double)N * eps_of(smax);
            }
            // per block: W+ = pinv(W + 1e-12 I) + 1e-12 I   (:57)
            for (int i = lane; i < N; i += WAVE) {
                double o[6], f[4], B[4][6], W[4][4], V[4][4];
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                tril_block(T, o, f, B);
                block_W(B, W);
                jacobi4<true>(W, V);
                double inv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) inv[a] = (W[a][a] > tolW) ? 1.0 / W[a][a] : 0.0;
                double Wp[10];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b)
                        Wp[a * (a + 1) / 2 + b] = V[a][0] * inv[0] * V[b][0] + V[a][1] * inv[1] * V[b][1] + V[a][2] * inv[2] * V[b][2]
                                                  + V[a][3] * inv[3] * V[b][3] + ((a == b) ? 1e-12 : 0.0);
                gh_store_point(g, w, pts, i, o, f, B, Wp);
            }
        }
        wave_sync();
        if (it == 1) phase_stamp(dbg, 43);
        // ---- Ghat = sum Ap' W+ Ap and ghat = sum Ap' W+ w   (:59-62) ----
        gh_sweep<0>(g, N); gh_sweep<1>(g, N); gh_sweep<2>(g, N); gh_sweep<3>(g, N); gh_sweep<4>(g, N);
        gh_sweep<5>(g, N); gh_sweep<6>(g, N); gh_sweep<7>(g, N); gh_sweep<8>(g, N); gh_sweep<9>(g, N);
        wave_sync();
        if (it == 1) phase_stamp(dbg, 44);
        for (int e = lane; e < 729; e += WAVE) {                             // Ghat[(q,i1),(q',i1')] = H[6 tri(q,q') + hht(i1,i1')]
            const int r = e / 27, cc = e % 27;
            const int q = r % 9, i1 = r / 9, qq = cc % 9, i1p = cc / 9;
            const int hi = (q > qq) ? q : qq, lo = (q > qq) ? qq : q;
            g.G[e] = g.H[6 * (hi * (hi + 1) / 2 + lo) + hht_index(i1, i1p)];
        }
        wave_sync();
        if (Model::IDENTITY_D) {                                             // A = Ap: A'WA = Ghat, A'Ww = ghat
            for (int e = lane; e < 729 + 27; e += WAVE) {
                if (e < 729) g.M[(e / 27) * ld + e % 27] = g.G[e] + ((e / 27 == e % 27) ? 1e-12 : 0.0);
                else g.M[(e - 729) * ld + n] = g.H[270 + e - 729];
            }
        } else {
            for (int e = lane; e < 27 * u; e += WAVE) {                      // Y = Ghat D
                const int r = e / u, pcol = e % u;
                double a = 0.0;
                for (int k = 0; k < 27; ++k) a += g.G[r * 27 + k] * g.D[k * u + pcol];
                g.Y[e] = a;
            }
            wave_sync();
            for (int e = lane; e < u * u + u; e += WAVE) {                   // M = [D'Y + 1e-12 I ...], b = [D' ghat; -g]
                const int pr = e / u, pc = e % u;
                double a = 0.0;
                if (e < u * u) {
                    for (int k = 0; k < 27; ++k) a += g.D[k * u + pr] * g.Y[k * u + pc];
                    if (factored) a += g.V[(pr >= pc) ? tri_index(pr, pc) : tri_index(pc, pr)];
                    g.M[pr * ld + pc] = a + ((pr == pc) ? 1e-12 : 0.0);
                } else {
                    for (int k = 0; k < 27; ++k) a += g.D[k * u + pc] * g.H[270 + k];
                    if (factored) a += g.V[u * (u + 1) / 2 + pc];
                    g.M[pc * ld + n] = a;
                }
            }
        }
        if (lane < c) g.M[(u + lane) * ld + u + lane] = 1e-12;
        wave_sync();
        double chkM = 0.0;
        for (int e = lane; e < n * ld; e += WAVE) chkM += g.M[e];
        if (!(fabs(wave_sum(chkM)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        if (it == 1) phase_stamp(dbg, 45);
        // aux = pinv(M + 1e-12 I) * b   (:67)
        if (Model::REDUNDANT_CONSTRAINTS) wave_pinv_solve_sym(g.M, g.V, n, g.dt, g.Y);
        else if (!wave_solve_gj<Model::U + Model::C>(g.M, g.dt)) wave_pinv_solve_sym(g.M, g.V, n, g.dt, g.Y);   // singular: pinv truncates
        wave_sync();
        if (lane < 27) {                                                     // dT = D dt
            double a = 0.0;
            if (Model::IDENTITY_D) a = g.dt[lane];
            else for (int k = 0; k < u; ++k) a += g.D[lane * u + k] * g.dt[k];
            g.dT[lane] = a;
        }
        wave_sync();
        double dTr[27];
        load_uniform27(g.dT, dTr);
        if (it == 1) phase_stamp(dbg, 46);
        // ---- v = -B' W+ (A dt - w)   (:69) ----
        double obj = 0.0, diff = 0.0;
        for (int i = lane; i < N; i += WAVE) {
            double o[6], f[4], B[4][6], Ad[4];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            tril_block(T, o, f, B);
            {
                double m[3][3], t1[3][3], t2[3][3];
                tril_slices(dTr, o, m, t1, t2);
                tril_quad(m, o[2], o[3], o[4], o[5], Ad);                    // Ap_i (D dt)
            }
            double Wp[10], r[4];
#pragma unroll
            for (int k = 0; k < 10; ++k) Wp[k] = g.pp[14 * i + k];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                r[a] = wp_at(Wp, a, 0) * Ad[0] + wp_at(Wp, a, 1) * Ad[1] + wp_at(Wp, a, 2) * Ad[2] + wp_at(Wp, a, 3) * Ad[3] - g.pp[14 * i + 10 + a];
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
            double bn[6] = {0, 0, 0, 0, 0, 0}, sterm = 0.0;                  // strong direction: -cs (B'n) n'(A dt - w)
            if (factored) {
                double W[4][4], Wq[10], nn[4], cs = 0.0;
                block_W(B, W);
                pinv_block_deflated<true>(B, W, tolW, Wq, nn, &cs);
                double nwv = 0.0;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    double sw = -f[a];
#pragma unroll
                    for (int k = 0; k < 6; ++k) sw -= B[a][k] * (x.v[k] - o[k]);
                    nwv += nn[a] * (Ad[a] - sw);                             // n'(A dt - w)
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) bn[k] = B[0][k] * nn[0] + B[1][k] * nn[1] + B[2][k] * nn[2] + B[3][k] * nn[3];
                sterm = cs * nwv;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double v = -(B[0][k] * r[0] + B[1][k] * r[1] + B[2][k] * r[2] + B[3][k] * r[3]) - bn[k] * sterm;
                g.pp[14 * i + k] = v;
                obj += v * v;
                const double d = o[k] - x.v[k] - v;
                diff += d * d;
            }
        }
        obj = wave_sum(obj);
        diff = wave_sum(diff);
        const double dtk = (lane < u) ? g.dt[lane] : 0.0;
        const double ndt2 = wave_sum(dtk * dtk);
        if (it == 1) phase_stamp(dbg, 47);
        if (dbg && lane == 0 && it <= 8) { dbg[96 + 3 * (it - 1)] = obj; dbg[97 + 3 * (it - 1)] = ndt2; dbg[98 + 3 * (it - 1)] = diff; }
        if (sqrt(ndt2) < GH_TOL && sqrt(diff) < GH_TOL) break;               // :71-73 (dy is empty)
        if (obj > objFunc) break;                                            // :75-76, factor = 1
        objFunc = obj;                                                       // :78
        for (int i = lane; i < N; i += WAVE) {                               // xi = x + v; ti = ti + dt   (:80)
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) g.xi[6 * i + k] = x.v[k] + g.pp[14 * i + k];
        }
        if (lane < u) g.p[lane] += dtk;
        wave_sync();
    }
    return (it > GH_IT_MAX) ? GH_IT_MAX : it;                                // :82
}

template <class Model> __device__ __forceinline__ int gh_model_bad(const Model&) { return 0; }
template <> __device__ __forceinline__ int gh_model_bad<NordbergModel>(const NordbergModel& m) { return m.bad; }

template <class Model, bool JAC>
__global__ void __launch_bounds__(64, 1) k_gh_tft_pose(const LinearTftArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    PoseLds* w = reinterpret_cast<PoseLds*>(smem);
    constexpr int base = (POSE_LDS_DOUBLES + 1) & ~1;
    JacobiLds* jw = JAC ? reinterpret_cast<JacobiLds*>(smem + base) : nullptr;
    double* ghbase = smem + base + (JAC ? ((JACOBI_LDS_DOUBLES + 1) & ~1) : 0);
    const int lane = lane_id();
    for (long b = blockIdx.x; b < a.B; b += gridDim.x) {
        if ((a.flags & FLAG_ONLY_RETRY) && a.status[b] != ST_RETRY) continue;
        const int N = opaque_int(a.N);                                       // (not hoisted out of the one-trip triplet loop: tft_kernel.h)
        double* dbg = a.dbg ? a.dbg + b * DBG_STRIDE : nullptr;
        const double* pts = a.corresp + b * 6 * (long)N;                     // re-read through L2 (LDS is taken by the GH workspace)
        wave_sync();
        GhWork g = gh_carve(ghbase, Model::U, Model::C, a.spill ? 0 : N);
        if (a.spill) { g.xi = a.spill + blockIdx.x * a.spill_stride; g.pp = g.xi + 6 * (long)N; }   // large N: per-correspondence state in global memory
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        int status = ST_OK, iters = 0;
        if (N < 7) {
            status = ST_TOO_FEW;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        } else {
            normalise3(pts, N, w->nrm);                                      // ResslTFTPoseEstimation.m:48-50
            const bool ok = linear_tft_wave<JAC>(w, jw, pts, N, true, dbg);  // :53
            if (!ok) {
                status = ST_RETRY;
            } else {
                Model model;
                model.init(w, g);                                            // initial parameters; cameras P1, P2, P3 of the linear solution
                const int model_bad = gh_model_bad(model);
                // x_est: reprojection of the projective triangulation with P1, P2, P3   (ResslTFT...m:72-75)
                tri_pass(w, pts, N, TRI_REPROJECT, 1, w->P[0], w->P[1], g.xi, w->nrm);
                wave_sync();
                int gst = ST_OK;
                iters = gauss_helmert_wave(w, g, model, pts, N, &gst, dbg, (a.flags & FLAG_GH_EXACT) != 0);  // :84
                wave_sync();
                model.eval(g);                                               // T from p_opt   (:87-94)
                if (lane < 27) w->t[lane] = g.Tc[lane];
                wave_sync();
                transform_tft_inverse(w->t, w->T1, w->Lp, [w](int v) { return normal_matrix(w->nrm, v); });   // :96
                status = rt_from_tft_wave(w, pts, N, dbg);                   // :99
                if (gst != ST_OK) status = gst;
                if (model_bad) status = ST_RANK;
                write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
                if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
                if (a.reconst) final_reconst(w, pts, N, a.reconst + b * 3 * (long)N);   // :102-103
                double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
                const bool bad = !(fabs(chk) <= 1.79e308);
                if (wave_any(bad)) { if (status == ST_OK) status = ST_NONFINITE; wave_nan_outputs(a.Rt2, a.Rt3, a.T, a.reconst, b, N); }
            }
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = iters;
            a.status[b] = status;
        }
    }
}

}  // namespace tff
