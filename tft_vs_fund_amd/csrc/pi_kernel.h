// PiPoseEstimation / PiColPoseEstimation (TFT_methods/PiPoseEstimation.m:50-182,
// TFT_methods/PiColPoseEstimation.m:50-218): Gauss-Helmert refinement of the Ponce-Hebert
// Pi-matrix parameterisation of three views, one wavefront per triplet.  SURVEY 8(f) rank 2.
//
// Parameters: nine 3-vectors pi_b (27 doubles), block b multiplies the homogeneous point of view
// v(b) = b / 3.  Every equation of both callbacks is multilinear in the nine scalars
// a_b = pi_b . p_v(b), so for correspondence i
//     A_i[r][3b + k] = c[r][b] * p_v(b)[k],      B_i[r][2v + j] = sum_{b in view v} c[r][b] * pi_b[j],
// with c[r][b] = df_r / da_b (E x 9, sparse).  Gauss_Helmert.m's products become
//     A'WA[(b,k),(b',k')] = sum_i Z_i[b][b'] p_v(b)[k] p_v(b')[k'],   Z_i = c' W_i^+ c   (45 block pairs x 9 sums),
//     A'Ww[(b,k)]         = sum_i (c' W_i^+ w_i)[b] p_v(b)[k]                           (27 sums),
// accumulated three block pairs (27 accumulators) per sweep, one correspondence per lane.
// W_i = B_i B_i' is E x E (E = 4 or 5): per-lane Jacobi, pinv with the global tolerance E N eps(max lambda).
// The (27 + C) x (27 + C) KKT system is solved by pivoted elimination in LDS.
//
// PiColPoseEstimation.m:186 carries the wrong sign for dA(ind2+4)/dpi21 (its B, :198, has the right
// one); the restatement keeps it (Model::a_quirk).  The null-space bases / signs the reference takes
// from svd (null(P), null(M.')) are gauge choices of the projective frame: any choice gives the same
// problem in exact arithmetic (checked against the oracle by flipping/rotating them).
#pragma once
#include "gh_kernel.h"
#include "f_kernel.h"

namespace tff {

constexpr int ST_NO_PARAM = 5;        // PiColPoseEstimation.m:84-89: error('The minimal param could not be found')

struct PiWork {
    double* p;      // 27 (+1)  parameters
    double* dt;     // n        KKT solution
    double* H;      // 405 + 27 accumulated sums: H[9 e + 3 k + k'], e = tri(b, b'); then rhs[27]
    double* M;      // n x (n+1) augmented KKT matrix
    double* V;      // n x n    eigenvectors of the KKT matrix, then n coefficients (pseudo-inverse path)
    double* xi;     // 6N       current estimates of the observations
    double* pp;     // PP * N   per correspondence: W+ (packed lower), W+ w; later v (6)
    double* sn;     // workgroup kernel (Pi): 6N, per correspondence the strong direction n (4), its weight cs and n'w; else null
};
__host__ __device__ constexpr int pi_pp(int E) { return E * (E + 1) / 2 + E; }
__host__ __device__ inline int pi_lds_doubles(int E, int C, int N, bool pinv_kkt) {
    const int n = 27 + C;
    (void)pinv_kkt;                                                          // V is always there: the fall-back of the elimination needs it too
    return 28 + ((n + 1) & ~1) + 432 + ((n * (n + 1) + 1) & ~1) + ((n * n + n + 1) & ~1) + 6 * N + pi_pp(E) * N + 8;
}
__device__ inline PiWork pi_carve(double* base, int E, int C, int N, bool pinv_kkt) {
    PiWork g;
    const int n = 27 + C;
    double* q = base;
    g.p = q; q += 28;
    g.dt = q; q += (n + 1) & ~1;
    g.H = q; q += 432;
    g.M = q; q += (n * (n + 1) + 1) & ~1;
    (void)pinv_kkt;
    g.V = q; q += (n * n + n + 1) & ~1;
    g.xi = q; q += 6 * N;
    g.pp = q;
    g.sn = nullptr;
    return g;
}

// ---- small dense helpers for the lane-0 set-up code -------------------------------------------------------
__device__ __forceinline__ double det3v(const double* a, const double* b, const double* c, int i0, int i1, int i2) {
    return a[i0] * (b[i1] * c[i2] - b[i2] * c[i1]) - a[i1] * (b[i0] * c[i2] - b[i2] * c[i0]) + a[i2] * (b[i0] * c[i1] - b[i1] * c[i0]);
}
// unit vector orthogonal to three 4-vectors (generalised cross product); for the rows of a 3x4 camera: its null vector
__device__ __forceinline__ void cross4(const double* r0, const double* r1, const double* r2, double (&n)[4]) {
    n[0] = det3v(r0, r1, r2, 1, 2, 3);
    n[1] = -det3v(r0, r1, r2, 0, 2, 3);
    n[2] = det3v(r0, r1, r2, 0, 1, 3);
    n[3] = -det3v(r0, r1, r2, 0, 1, 2);
    const double s = rsqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) n[k] *= s;
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double dot4(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
// P (3x4) <- P * M (4x4, M[row][col])
__device__ __forceinline__ void cam_mul(double (&P)[3][4], const double (&M)[4][4]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double a0 = P[r][0], a1 = P[r][1], a2 = P[r][2], a3 = P[r][3];
#pragma unroll
        for (int c = 0; c < 4; ++c) P[r][c] = a0 * M[0][c] + a1 * M[1][c] + a2 * M[2][c] + a3 * M[3][c];
    }
}
// inverse of the 3x3 made of columns (c0, c1, c2) of a 3x4 camera
__device__ __forceinline__ Mat3 cam_sub_inv(const double (&P)[3][4], int c0, int c1, int c2) {
    Mat3 S;
#pragma unroll
    for (int r = 0; r < 3; ++r) { S.m[r][0] = P[r][c0]; S.m[r][1] = P[r][c1]; S.m[r][2] = P[r][c2]; }
    return mat3_inv(S);
}
// cameras of the linear solution: P1 = [I|0], P2 = [reshape(a(1:9),3,3) e21], P3 = [reshape(a(10:18),3,3) e31]   (linearTFT.m:88-90)
__device__ __forceinline__ void pi_linear_cameras(const PoseLds* w, double (&P1)[3][4], double (&P2)[3][4], double (&P3)[3][4]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            P1[r][c] = (r == c) ? 1.0 : 0.0;
            P2[r][c] = (c < 3) ? w->pa[3 * c + r] : w->epi[r];
            P3[r][c] = (c < 3) ? w->pa[9 + 3 * c + r] : w->epi[3 + r];
        }
}
__device__ __forceinline__ void pi_store_cameras(PoseLds* w, const double (&P1)[3][4], const double (&P2)[3][4], const double (&P3)[3][4]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { w->Pfin[0][4 * r + c] = P1[r][c]; w->P[0][4 * r + c] = P2[r][c]; w->P[1][4 * r + c] = P3[r][c]; }
}
// camera from a Pi matrix: the three columns (c0,c1,c2) of P are inv(Pi), the fourth is zero
__device__ __forceinline__ void pi_camera(const double* p9, int c0, int c1, int c2, double* Pout /*row-major 12*/) {
    Mat3 S;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) S.m[r][c] = p9[3 * r + c];            // Pi = reshape(pi(1:9),3,3).' : rows are the pi vectors
    const Mat3 I = mat3_inv(S);
#pragma unroll
    for (int k = 0; k < 12; ++k) Pout[k] = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) { Pout[4 * r + c0] = I.m[r][0]; Pout[4 * r + c1] = I.m[r][1]; Pout[4 * r + c2] = I.m[r][2]; }
}

// ---- PiPoseEstimation ------------------------------------------------------------------------------------
struct PiModel {
    static constexpr int E = 4, C = 9;
    static constexpr bool PINV_KKT = false;      // 27 - 9 = 18 = dim of the trifocal variety: the KKT matrix is regular
    // constraint c: |pi_b|^2 = 1 when CB[c][0] == CB[c][1], else pi_b . pi_b' = 0   (PiPoseEstimation.m:123-137)
    __device__ static constexpr int cb(int c, int s) {
        constexpr int t[9][2] = {{2, 2}, {5, 5}, {8, 8}, {0, 0}, {4, 4}, {6, 6}, {0, 2}, {4, 5}, {6, 8}};
        return t[c][s];
    }
    // sparsity of c[r][b]
    __host__ __device__ static constexpr bool nz(int r, int b) {
        constexpr int m[4] = {(1 << 1) | (1 << 2) | (1 << 4) | (1 << 5), (1 << 0) | (1 << 2) | (1 << 7) | (1 << 8),
                              (1 << 3) | (1 << 5) | (1 << 6) | (1 << 8), (1 << 0) | (1 << 1) | (1 << 3) | (1 << 4) | (1 << 6) | (1 << 7)};
        return (m[r] >> b) & 1;
    }
    // a = [a21 a31 a41 a12 a32 a42 a13 a23 a43]: f (:152-153) and c = df/da (:156-164)
    __device__ static __forceinline__ void coeffs(const double (&a)[9], double (&f)[4], double (&c)[4][9]) {
        f[0] = a[2] * a[4] - a[1] * a[5];
        c[0][1] = -a[5]; c[0][2] = a[4]; c[0][4] = a[2]; c[0][5] = -a[1];
        f[1] = a[2] * a[7] - a[0] * a[8];
        c[1][0] = -a[8]; c[1][2] = a[7]; c[1][7] = a[2]; c[1][8] = -a[0];
        f[2] = a[5] * a[6] - a[3] * a[8];
        c[2][3] = -a[8]; c[2][5] = a[6]; c[2][6] = a[5]; c[2][8] = -a[3];
        const double a04 = a[0] * a[4], a13 = a[1] * a[3];
        f[3] = a04 * a[6] - a13 * a[7];
        c[3][0] = a[4] * a[6]; c[3][1] = -a[3] * a[7]; c[3][3] = -a[1] * a[7]; c[3][4] = a[0] * a[6]; c[3][6] = a04; c[3][7] = -a13;
    }
    __device__ static __forceinline__ void a_quirk(double (&)[4][9]) {}
    // lane 0: PiPoseEstimation.m:61-86.  Cameras P1,P2,P3 (for the initial triangulation) -> w->Pfin[0], w->P[0], w->P[1]; parameters -> p.
    __device__ static inline int init(PoseLds* w, double* p) {
        double P1[3][4], P2[3][4], P3[3][4];
        pi_linear_cameras(w, P1, P2, P3);
        double M[4][4];
        {
            double n1[4], n2[4], n3[4], n4[4];
            cross4(P1[0], P1[1], P1[2], n1);                                 // null(P1), null(P2), null(P3)   (:61)
            cross4(P2[0], P2[1], P2[2], n2);
            cross4(P3[0], P3[1], P3[2], n3);
            cross4(n1, n2, n3, n4);                                          // null(M.')   (:62)
#pragma unroll
            for (int r = 0; r < 4; ++r) { M[r][0] = n1[r]; M[r][1] = n2[r]; M[r][2] = n3[r]; M[r][3] = n4[r]; }
        }
        cam_mul(P1, M); cam_mul(P2, M); cam_mul(P3, M);                       // :63
        double Pi[3][4][3];                                                  // :66-69
        {
            const Mat3 I1 = cam_sub_inv(P1, 1, 2, 3), I2 = cam_sub_inv(P2, 0, 2, 3), I3 = cam_sub_inv(P3, 0, 1, 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pi[0][0][c] = 0.0; Pi[0][1][c] = I1.m[0][c]; Pi[0][2][c] = I1.m[1][c]; Pi[0][3][c] = I1.m[2][c];
                Pi[1][0][c] = I2.m[0][c]; Pi[1][1][c] = 0.0; Pi[1][2][c] = I2.m[1][c]; Pi[1][3][c] = I2.m[2][c];
                Pi[2][0][c] = I3.m[0][c]; Pi[2][1][c] = I3.m[1][c]; Pi[2][2][c] = 0.0; Pi[2][3][c] = I3.m[2][c];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                        // Pi_k / norm(Pi_k(4,:))   (:72)
            const double s = rsqrt(dot3(Pi[k][3], Pi[k][3]));
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Pi[k][r][c] *= s;
        }
        double qd[3], q4[3];                                                 // Q(i,i), Q(i,4)   (:73-76)
        {
            const int src[3] = {2, 0, 1};                                    // Q(1,:) from Pi3(1,:), Q(2,:) from Pi1(2,:), Q(3,:) from Pi2(3,:)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double* ri = Pi[src[i]][i];
                const double* r4 = Pi[src[i]][3];
                const double d = dot3(ri, r4);
                const double e0 = ri[0] - d * r4[0], e1 = ri[1] - d * r4[1], e2 = ri[2] - d * r4[2];
                qd[i] = 1.0 / sqrt(e0 * e0 + e1 * e1 + e2 * e2);
                q4[i] = -qd[i] * d;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)                                          // Pi_k = Q * Pi_k   (:77)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) Pi[k][i][c] = qd[i] * Pi[k][i][c] + q4[i] * Pi[k][3][c];
        auto inv_q = [&](double (&P)[3][4]) {                                // P * inv(Q)   (:79)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double last = P[r][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { last -= P[r][i] * q4[i] / qd[i]; P[r][i] = P[r][i] / qd[i]; }
                P[r][3] = last;
            }
        };
        inv_q(P1); inv_q(P2); inv_q(P3);
        pi_store_cameras(w, P1, P2, P3);
        // pi = [Pi1(2:4,:)'(:); Pi2([1 3 4],:)'(:); Pi3([1 2 4],:)'(:)]   (:86)
        const int rows[3][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) p[9 * k + 3 * j + c] = Pi[k][rows[k][j]][c];
        return ST_OK;
    }
    // lane 0: cameras from the optimised parameters   (:94-100)
    __device__ static inline void cameras(const double* p, PoseLds* w) {
        pi_camera(p, 1, 2, 3, w->Pfin[0]);
        pi_camera(p + 9, 0, 2, 3, w->Pfin[1]);
        pi_camera(p + 18, 0, 1, 3, w->Pfin[2]);
    }
};

// ---- PiColPoseEstimation (collinear camera centres) ---------------------------------------------------------
struct PiColModel {
    static constexpr int E = 5, C = 11;
    // the KKT matrix of this parameterisation is numerically singular (two singular values under pinv's
    // tolerance on typical scenes): Gauss_Helmert.m:67's pinv truncates them, so does the eigen-decomposition path here
    static constexpr bool PINV_KKT = true;
    __device__ static constexpr int cb(int c, int s) {                       // PiColPoseEstimation.m:149-160
        constexpr int t[11][2] = {{0, 0}, {3, 3}, {6, 6}, {7, 7}, {8, 8}, {0, 1}, {0, 2}, {1, 2}, {3, 4}, {3, 5}, {4, 5}};
        return t[c][s];
    }
    __host__ __device__ static constexpr bool nz(int r, int b) {
        constexpr int m[5] = {(1 << 1) | (1 << 2) | (1 << 4) | (1 << 5), (1 << 1) | (1 << 2) | (1 << 7) | (1 << 8),
                              (1 << 4) | (1 << 5) | (1 << 7) | (1 << 8), (1 << 0) | (1 << 1) | (1 << 3) | (1 << 4) | (1 << 6) | (1 << 7),
                              (1 << 0) | (1 << 2) | (1 << 3) | (1 << 5) | (1 << 6) | (1 << 8)};
        return (m[r] >> b) & 1;
    }
    // a = [a21 a31 a41 a12 a32 a42 aw3 a33 a43]: f (:175-177), c = df/da (the coefficients of B, :195-205)
    __device__ static __forceinline__ void coeffs(const double (&a)[9], double (&f)[5], double (&c)[5][9]) {
        f[0] = a[2] * a[4] - a[1] * a[5];
        c[0][1] = -a[5]; c[0][2] = a[4]; c[0][4] = a[2]; c[0][5] = -a[1];
        f[1] = a[2] * a[7] - a[1] * a[8];
        c[1][1] = -a[8]; c[1][2] = a[7]; c[1][7] = a[2]; c[1][8] = -a[1];
        f[2] = a[5] * a[7] - a[4] * a[8];
        c[2][4] = -a[8]; c[2][5] = a[7]; c[2][7] = a[5]; c[2][8] = -a[4];
        const double t3 = a[1] * a[3] - a[0] * a[4], t4 = a[2] * a[3] - a[0] * a[5];
        f[3] = a[1] * a[4] * a[6] + t3 * a[7];
        c[3][0] = -a[4] * a[7]; c[3][1] = a[4] * a[6] + a[3] * a[7]; c[3][3] = a[1] * a[7]; c[3][4] = a[1] * a[6] - a[0] * a[7];
        c[3][6] = a[1] * a[4]; c[3][7] = t3;
        f[4] = a[2] * a[5] * a[6] + t4 * a[8];
        c[4][0] = -a[5] * a[8]; c[4][2] = a[5] * a[6] + a[3] * a[8]; c[4][3] = a[2] * a[8]; c[4][5] = a[2] * a[6] - a[0] * a[8];
        c[4][6] = a[2] * a[5]; c[4][8] = t4;
    }
    // the reference's A(ind2+4,1:3) = +p1' (pi32'p2)(pi33'p3)   (:186), opposite to the derivative used in its B (:198)
    __device__ static __forceinline__ void a_quirk(double (&c)[5][9]) { c[3][0] = -c[3][0]; }
    // lane 0: PiColPoseEstimation.m:61-113
    __device__ static __forceinline__ int init(PoseLds* w, double* p) {
        double P1[3][4], P2[3][4], P3[3][4];
        pi_linear_cameras(w, P1, P2, P3);
        double M[4][4];
        {
            double n1[4], n2[4], n3[4];
            cross4(P1[0], P1[1], P1[2], n1);                                 // M = [null(P1), null(P2)]   (:61)
            cross4(P2[0], P2[1], P2[2], n2);
            cross4(P3[0], P3[1], P3[2], n3);
            // coeff = M \ null(P3): 4x2 least squares through the normal equations   (:62)
            const double g12 = dot4(n1, n2), r1 = dot4(n1, n3), r2 = dot4(n2, n3);
            const double det = 1.0 - g12 * g12;
            const double c1 = (r1 - g12 * r2) / det, c2 = (r2 - g12 * r1) / det;
            // null(M.'): an orthonormal basis of the complement of span(n1, n2) (basis choice is a gauge of the frame)
            double q2[4], v3[4], v4[4];
            {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { q2[k] = n2[k] - g12 * n1[k]; s += q2[k] * q2[k]; }
                s = rsqrt(s);
#pragma unroll
                for (int k = 0; k < 4; ++k) q2[k] *= s;
                double best = -1.0;
#pragma unroll
                for (int ax = 0; ax < 4; ++ax) {                             // the coordinate axis farthest from the plane
                    double e[4], nn = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { e[k] = ((k == ax) ? 1.0 : 0.0) - n1[ax] * n1[k] - q2[ax] * q2[k]; nn += e[k] * e[k]; }
                    if (nn > best) {
                        best = nn;
                        const double is = rsqrt(nn);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v3[k] = e[k] * is;
                    }
                }
                cross4(n1, q2, v3, v4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { M[r][0] = c1 * n1[r]; M[r][1] = c2 * n2[r]; M[r][2] = v3[r]; M[r][3] = v4[r]; }   // :63
        }
        cam_mul(P1, M); cam_mul(P2, M); cam_mul(P3, M);                       // :64
        double Pi[3][4][3];                                                  // :67-70
        {
            const Mat3 I1 = cam_sub_inv(P1, 1, 2, 3), I2 = cam_sub_inv(P2, 0, 2, 3), I3 = cam_sub_inv(P3, 1, 2, 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Pi[0][0][c] = 0.0; Pi[0][1][c] = I1.m[0][c]; Pi[0][2][c] = I1.m[1][c]; Pi[0][3][c] = I1.m[2][c];
                Pi[1][0][c] = I2.m[0][c]; Pi[1][1][c] = 0.0; Pi[1][2][c] = I2.m[1][c]; Pi[1][3][c] = I2.m[2][c];
                Pi[2][0][c] = 0.0; Pi[2][1][c] = I3.m[0][c]; Pi[2][2][c] = I3.m[1][c]; Pi[2][3][c] = I3.m[2][c];
            }
        }
        auto scale = [&](int k, double s) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Pi[k][r][c] *= s;
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) scale(k, rsqrt(dot3(Pi[k][3], Pi[k][3])));   // :73
        double q02, q03, q12, q13, q23, q32;
        {
            double u1[3], v1[3], u2[3], v2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { u1[c] = Pi[0][2][c]; v1[c] = Pi[0][3][c]; u2[c] = Pi[1][2][c]; v2[c] = Pi[1][3][c]; }
            const double u1u1 = dot3(u1, u1), u1v1 = dot3(u1, v1), v1v1 = dot3(v1, v1);
            const double u2u2 = dot3(u2, u2), u2v2 = dot3(u2, v2), v2v2 = dot3(v2, v2);
            const double A = v1v1 * u2v2 - u1v1 * v2v2, B = v1v1 * u2u2 - u1u1 * v2v2, Cq = u1v1 * u2u2 - u1u1 * u2v2;   // :81-83
            const double disc = B * B - 4.0 * A * Cq;
            if (!(fabs(A) > 1e-10 && disc >= 0.0 && fabs(Cq) > 1e-10)) return ST_NO_PARAM;   // :84-89
            q23 = (-B + sqrt(disc)) / (2.0 * A);
            q32 = (-B + sqrt(disc)) / (2.0 * Cq);
            // A = u1 v1' - v1 u1', B = u2 v2' - v2 u2'   (:90); the denominators of the Pi2 rows use u1 as the reference does (:93-94)
            double Au1[3], Av1[3], Bu2[3], Bv2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Au1[c] = u1[c] * u1v1 - v1[c] * u1u1;
                Av1[c] = u1[c] * v1v1 - v1[c] * u1v1;
                Bu2[c] = u2[c] * u2v2 - v2[c] * u2u2;
                Bv2[c] = u2[c] * v2v2 - v2[c] * u2v2;
            }
            const double dA = dot3(u1, Av1), dB = dot3(u1, Bv2);
            q13 = dot3(Pi[0][1], Au1) / dA;                                  // Q1(2,4)   (:91)
            q12 = -dot3(Pi[0][1], Av1) / dA;                                 // Q1(2,3): A.' v1 = -A v1   (:92)
            q03 = dot3(Pi[1][0], Bu2) / dB;                                  // Q1(1,4)   (:93)
            q02 = -dot3(Pi[1][0], Bv2) / dB;                                 // Q1(1,3)   (:94)
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)                                          // Pi_k = Q1 * Pi_k   (:96)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double r2 = Pi[k][2][c], r3 = Pi[k][3][c];
                Pi[k][0][c] += q02 * r2 + q03 * r3;
                Pi[k][1][c] += q12 * r2 + q13 * r3;
                Pi[k][2][c] = r2 + q23 * r3;
                Pi[k][3][c] = q32 * r2 + r3;
            }
        scale(0, rsqrt(dot3(Pi[0][1], Pi[0][1])));                           // :97-99
        scale(1, rsqrt(dot3(Pi[1][0], Pi[1][0])));
        {
            const double d0 = Pi[2][1][0] - Pi[2][0][0], d1 = Pi[2][1][1] - Pi[2][0][1], d2 = Pi[2][1][2] - Pi[2][0][2];
            scale(2, rsqrt(d0 * d0 + d1 * d1 + d2 * d2));
        }
        const double n2r = sqrt(dot3(Pi[2][2], Pi[2][2])), n3r = sqrt(dot3(Pi[2][3], Pi[2][3]));   // Q2 = diag(1,1,1/n2r,1/n3r)   (:100-102)
#pragma unroll
        for (int k = 0; k < 3; ++k)                                          // :103
#pragma unroll
            for (int c = 0; c < 3; ++c) { Pi[k][2][c] /= n2r; Pi[k][3][c] /= n3r; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { Pi[2][1][c] -= Pi[2][0][c]; Pi[2][0][c] = 0.0; }   // Pi3(1:2,:) -= Pi3([1 1],:)   (:104)
        // P * inv(Q2 Q1) = P * inv(Q1) * diag(1,1,n2r,n3r);  Q1 = [I R; 0 S], inv = [I -R inv(S); 0 inv(S)]   (:106)
        {
            const double ds = 1.0 / (1.0 - q23 * q32);
            const double s00 = ds, s01 = -q23 * ds, s10 = -q32 * ds, s11 = ds;   // inv(S)
            const double t02 = -(q02 * s00 + q03 * s10), t03 = -(q02 * s01 + q03 * s11);
            const double t12 = -(q12 * s00 + q13 * s10), t13 = -(q12 * s01 + q13 * s11);
            auto inv_q = [&](double (&P)[3][4]) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double a0 = P[r][0], a1 = P[r][1], a2 = P[r][2], a3 = P[r][3];
                    P[r][2] = (a0 * t02 + a1 * t12 + a2 * s00 + a3 * s10) * n2r;
                    P[r][3] = (a0 * t03 + a1 * t13 + a2 * s01 + a3 * s11) * n3r;
                }
            };
            inv_q(P1); inv_q(P2); inv_q(P3);
        }
        pi_store_cameras(w, P1, P2, P3);
        // pi = [Pi1(2:4,:)'(:); Pi2([1 3 4],:)'(:); Pi3(2:4,:)'(:)]   (:113)
        const int rows[3][3] = {{1, 2, 3}, {0, 2, 3}, {1, 2, 3}};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) p[9 * k + 3 * j + c] = Pi[k][rows[k][j]][c];
        return ST_OK;
    }
    // lane 0: cameras from the optimised parameters   (:121-127)
    __device__ static inline void cameras(const double* p, PoseLds* w) {
        pi_camera(p, 1, 2, 3, w->Pfin[0]);
        pi_camera(p + 9, 0, 2, 3, w->Pfin[1]);
        pi_camera(p + 18, 1, 2, 3, w->Pfin[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) w->Pfin[2][4 * r + 0] = -w->Pfin[2][4 * r + 1];   // P3(:,1) = -P3(:,2)
    }
};

// ---- per-correspondence evaluation ----------------------------------------------------------------------------
template <int E>
struct PiPoint {
    double f[E];
    double c[E][9];      // df/da (true derivatives: B); Model::a_quirk turns it into the coefficients of A
    double B[E][6];
};
// p_v(b)[k] of the observation o = [x1 y1 x2 y2 x3 y3]
__device__ __forceinline__ double hom_at(const double (&o)[6], int v, int k) { return (k == 2) ? 1.0 : o[2 * v + k]; }

template <class Model, bool WITH_B>
__device__ __forceinline__ void pi_eval(const double (&pi)[27], const double (&o)[6], PiPoint<Model::E>& pt) {
    constexpr int E = Model::E;
    double a[9];
#pragma unroll
    for (int b = 0; b < 9; ++b) a[b] = pi[3 * b] * o[2 * (b / 3)] + pi[3 * b + 1] * o[2 * (b / 3) + 1] + pi[3 * b + 2];
    Model::coeffs(a, pt.f, pt.c);
    if (WITH_B) {
#pragma unroll
        for (int r = 0; r < E; ++r)
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb)
                        if (Model::nz(r, 3 * v + bb)) s += pt.c[r][3 * v + bb] * pi[3 * (3 * v + bb) + j];
                    pt.B[r][2 * v + j] = s;
                }
    }
}

// cyclic Jacobi on a symmetric E x E matrix held per lane
template <int E, bool WITH_V>
__device__ __forceinline__ void jacobi_small(double (&A)[E][E], double (&V)[E][E]) {
    if (WITH_V) {
#pragma unroll
        for (int i = 0; i < E; ++i)
#pragma unroll
            for (int j = 0; j < E; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            dg += A[i][i] * A[i][i];
#pragma unroll
            for (int j = 0; j < i; ++j) off += A[i][j] * A[i][j];
        }
        if (!(off > 1e-36 * dg)) break;
#pragma unroll
        for (int p = 0; p < E - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < E; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double app = A[p][p], aqq = A[q][q];
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double cs = rsqrt(1.0 + t * t), sn = t * cs;
                    A[p][p] = app - t * apq;
                    A[q][q] = aqq + t * apq;
                    A[p][q] = A[q][p] = 0.0;
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        if (r == p || r == q) continue;
                        const double arp = A[r][p], arq = A[r][q];
                        A[r][p] = A[p][r] = cs * arp - sn * arq;
                        A[r][q] = A[q][r] = sn * arp + cs * arq;
                    }
                    if (WITH_V) {
#pragma unroll
                        for (int k = 0; k < E; ++k) {
                            const double vkp = V[k][p], vkq = V[k][q];
                            V[k][p] = cs * vkp - sn * vkq;
                            V[k][q] = sn * vkp + cs * vkq;
                        }
                    }
                }
            }
    }
}
template <int E>
__device__ __forceinline__ void pi_block_W(const double (&B)[E][6], double (&W)[E][E]) {       // B B' + 1e-12 I
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double a = (i == j) ? 1e-12 : 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) a += B[i][k] * B[j][k];
            W[i][j] = W[j][i] = a;
        }
}
template <int E>
__device__ __forceinline__ double sym_at(const double* Wp, int a, int b) { return (a >= b) ? Wp[a * (a + 1) / 2 + b] : Wp[b * (b + 1) / 2 + a]; }

// one accumulation sweep: S < 15: block pairs e = 3S .. 3S+2 of A'WA; S == 15: A'Ww
template <class Model, int S>
__device__ inline void pi_sweep(const PiWork& g, const double (&pi)[27], int N, int first, int step, double* Hp) {
    constexpr int E = Model::E, PP = pi_pp(E), NW = E * (E + 1) / 2;
    const int lane = lane_id();
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll 1
    for (int i = first; i < N; i += step) {
        double o[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
        PiPoint<E> pt;
        pi_eval<Model, false>(pi, o, pt);
        Model::a_quirk(pt.c);
        const double* pw = g.pp + (long)PP * i;
        if constexpr (S < 15) {
            double Wp[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) Wp[k] = pw[k];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                constexpr int e0 = 3 * S;
                const int e = e0 + j;
                const int b = tri_row_of(e), bp = tri_col_of(e);             // folded: e is a constant after unrolling
                double z = 0.0;
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if (!Model::nz(r, b)) continue;
                    double y = 0.0;
#pragma unroll
                    for (int rr = 0; rr < E; ++rr)
                        if (Model::nz(rr, bp)) y += sym_at<E>(Wp, r, rr) * pt.c[rr][bp];
                    z += pt.c[r][b] * y;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double zk = z * hom_at(o, b / 3, k);
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) acc[9 * j + 3 * k + kk] += zk * hom_at(o, bp / 3, kk);
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < 9; ++b) {
                double y = 0.0;
#pragma unroll
                for (int r = 0; r < E; ++r)
                    if (Model::nz(r, b)) y += pt.c[r][b] * pw[NW + r];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[3 * b + k] += y * hom_at(o, b / 3, k);
            }
        }
    }
    const double tot = wave_reduce_scatter<32>(acc);
    const int idx = reduce32_index(lane);
    if ((lane & 1) == 0 && idx < 27) Hp[27 * S + idx] = tot;
}
// all 16 sweeps over the correspondences first, first + step, ...; sums of this wavefront -> Hp[0..431]
template <class Model, int S>
__device__ __forceinline__ void pi_sweeps_strided(const PiWork& g, const double (&pi)[27], int N, int first, int step, double* Hp) {
    pi_sweep<Model, S>(g, pi, N, first, step, Hp);
    if constexpr (S + 4 < 16) pi_sweeps_strided<Model, S + 4>(g, pi, N, first, step, Hp);
}
template <class Model, int S>
__device__ __forceinline__ void pi_sweeps(const PiWork& g, const double (&pi)[27], int N, int first, int step, double* Hp) {
    pi_sweep<Model, S>(g, pi, N, first, step, Hp);
    if constexpr (S < 15) pi_sweeps<Model, S + 1>(g, pi, N, first, step, Hp);
}

// w = -f - B (x - xi) (Gauss_Helmert.m:58); stores W+ (packed) and W+ w of correspondence i
template <int E>
__device__ __forceinline__ void pi_store_point(const PiWork& g, const PoseLds* w, const double* pts, int i, const double (&o)[6],
                                               const PiPoint<E>& pt, const double* Wp) {
    constexpr int PP = pi_pp(E), NW = E * (E + 1) / 2;
    const Pt6 x = premap(load_pt(pts, i), w->nrm);
    double wv[E];
#pragma unroll
    for (int a = 0; a < E; ++a) {
        double s = -pt.f[a];
#pragma unroll
        for (int k = 0; k < 6; ++k) s -= pt.B[a][k] * (x.v[k] - o[k]);
        wv[a] = s;
    }
    double* pw = g.pp + (long)PP * i;
#pragma unroll
    for (int k = 0; k < NW; ++k) pw[k] = Wp[k];
#pragma unroll
    for (int a = 0; a < E; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < E; ++b) s += sym_at<E>(Wp, a, b) * wv[b];
        pw[NW + a] = s;
    }
}

// Gauss_Helmert.m:38-83 for a Pi model.  xi holds x0 on entry.  Returns iterations; status via *st.
template <class Model>
__device__ inline int gauss_helmert_pi_wave(PoseLds* w, PiWork& g, const double* pts, int N, int* st, double* dbg, bool exact_pinv) {
    constexpr int E = Model::E, C = Model::C, u = 27, n = u + C, ld = n + 1, PP = pi_pp(E), NW = E * (E + 1) / 2;
    const int lane = lane_id();
    double objFunc = 0.0;                                                    // v0' v0, v0 = x0 - x   (:45-46)
    for (int i = lane; i < N; i += WAVE) {
        const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double d = g.xi[6 * i + k] - x.v[k]; objFunc += d * d; }
    }
    objFunc = wave_sum(objFunc);
    if (dbg && lane == 0) dbg[95] = objFunc;
    int it = 0;
#pragma unroll 1
    for (it = 1; it <= GH_IT_MAX; ++it) {
        double pi[27];
        load_uniform27(g.p, pi);
        // ---- W = B B' (:52): finite check and a bound on its largest eigenvalue; fast / exact pinv as in gh_kernel.h ----
        double f2max = 0.0;
        bool finite = true;
        for (int i = lane; i < N; i += WAVE) {
            double o[6], W[E][E];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            PiPoint<E> pt;
            pi_eval<Model, true>(pi, o, pt);
            pi_block_W<E>(pt.B, W);
            double chk = 0.0, fro2 = 0.0;
#pragma unroll
            for (int a = 0; a < E; ++a)
#pragma unroll
                for (int b = 0; b < E; ++b) { chk += W[a][b]; fro2 += W[a][b] * W[a][b]; }
            finite = finite && (fabs(chk) <= 1.79e308);
            f2max = (fro2 > f2max) ? fro2 : f2max;
        }
        f2max = wave_max(f2max);
        if (wave_any(!finite) || !(f2max <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :53-55
        // pinv's tolerance E N eps(max_i lambda_max(W_i)); only the binade of the maximum enters: the eigenvalue pass is skipped when
        // cheap bounds agree on it
        auto tolerance = [&]() {
            double umax = 0.0, lmax = 0.0;
            for (int i = lane; i < N; i += WAVE) {
                double o[6], W[E][E], up, lo;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                PiPoint<E> pt;
                pi_eval<Model, true>(pi, o, pt);
                pi_block_W<E>(pt.B, W);
                psd_lambda_max_bounds(W, up, lo);
                umax = (up > umax) ? up : umax;
                lmax = (lo > lmax) ? lo : lmax;
            }
            umax = wave_max(umax);
            lmax = wave_max(lmax);
            double smax = umax;
            if (eps_of(lmax) != eps_of(umax)) {
                smax = 0.0;
                for (int i = lane; i < N; i += WAVE) {
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
                smax = wave_max(smax);
            }
            return (double)E * (double)N * eps_of(smax);
        };
        // Pi (4 x 4 blocks): the deflated + factored weights of pi_wg_kernel.h; the 405 strong-direction sums go to S = g.V (dead until a
        // pseudo-inverse fall-back), (n, cs) are recomputed in the v update
        bool factored = false;
        double tolF = 0.0;
        if constexpr (E == 4 && !Model::PINV_KKT) {
            if (!exact_pinv) {
                tolF = tolerance();
                for (int e = lane; e < 405; e += WAVE) g.V[e] = 0.0;
                bool bad = false;
#pragma unroll 1
                for (int base = 0; base < N; base += WAVE) {                 // wave-uniform trip count (the butterflies need the whole wavefront)
                    const int i = base + lane;
                    double bv[27], tv = 0.0;
#pragma unroll
                    for (int k = 0; k < 27; ++k) bv[k] = 0.0;
                    if (i < N) {
                        double o[6], W[E][E], Wp[NW], nn[4], cs = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
                        PiPoint<E> pt;
                        pi_eval<Model, true>(pi, o, pt);
                        pi_block_W<E>(pt.B, W);
                        const bool ok = pinv_block_deflated<true>(pt.B, W, tolF, Wp, nn, &cs);
                        bad = !ok || bad;
#pragma unroll
                        for (int a = 0; a < E; ++a) Wp[a * (a + 1) / 2 + a] += 1e-12;
                        pi_store_point<E>(g, w, pts, i, o, pt, Wp);
                        if (ok) {
                            const Pt6 x = premap(load_pt(pts, i), w->nrm);
                            double nw = 0.0;                                 // n'w,  w = -f - B (x - xi)
#pragma unroll
                            for (int a = 0; a < E; ++a) {
                                double wa = -pt.f[a];
#pragma unroll
                                for (int k = 0; k < 6; ++k) wa -= pt.B[a][k] * (x.v[k] - o[k]);
                                nw += nn[a] * wa;
                            }
                            Model::a_quirk(pt.c);
                            const double sc = sqrt(cs);
#pragma unroll
                            for (int b = 0; b < 9; ++b) {                    // a[3 b + k] = (sum_r n_r c[r][b]) p_view(b)[k]
                                double an = 0.0;
#pragma unroll
                                for (int r = 0; r < E; ++r)
                                    if (Model::nz(r, b)) an += nn[r] * pt.c[r][b];
                                an *= sc;
#pragma unroll
                                for (int k = 0; k < 3; ++k) bv[3 * b + k] = an * hom_at(o, b / 3, k);
                            }
                            tv = sc * nw;
                        }
                    }
                    strong_accumulate<27>(bv, tv, g.V);
                }
                factored = !wave_any(bad);
            }
        }
        bool fast = !exact_pinv && (double)E * (double)N * eps_of(sqrt(f2max)) < 0.9e-12;
        if (factored) {
        } else if (fast) {
            bool bad = false;
            for (int i = lane; i < N; i += WAVE) {
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
            if (wave_any(bad)) fast = false;
        }
        if (!fast && !factored) {
            const double tolW = tolerance();
            // per block: W+ = pinv(W + 1e-12 I) + 1e-12 I   (:57)
            for (int i = lane; i < N; i += WAVE) {
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
        wave_sync();
        // ---- A'WA and A'Ww   (:59-62) ----
        pi_sweeps<Model, 0>(g, pi, N, lane, WAVE, g.H);
        wave_sync();
        for (int e = lane; e < n * ld; e += WAVE) g.M[e] = 0.0;
        wave_sync();
        for (int e = lane; e < 729 + 27; e += WAVE) {
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
        if (lane < C) {                                                      // constraints g, C   (callback; KKT borders :59-62)
            const int b = Model::cb(lane, 0), bp = Model::cb(lane, 1), row = u + lane;
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
        wave_sync();
        double chkM = 0.0;
        for (int e = lane; e < n * ld; e += WAVE) chkM += g.M[e];
        if (!(fabs(wave_sum(chkM)) <= 1.79e308)) { *st = ST_NONFINITE; break; }   // :63-65
        // aux = pinv(M + 1e-12 I) b   (:67)
        // Gauss-Jordan when the matrix is regular (then pinv is the inverse); the eigen-decomposition path reproduces the
        // truncation otherwise (degenerate geometry, e.g. collinear centres under the generic parameterisation)
        if (Model::PINV_KKT || !wave_solve_gj<n>(g.M, g.dt)) wave_pinv_solve_sym(g.M, g.V, n, g.dt, g.H);   // H is dead by now
        wave_sync();
        double dt[27];
        load_uniform27(g.dt, dt);
        // ---- v = -B' W+ (A dt - w)   (:69) ----
        double obj = 0.0, diff = 0.0;
        for (int i = lane; i < N; i += WAVE) {
            double o[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = g.xi[6 * i + k];
            PiPoint<E> pt;
            pi_eval<Model, true>(pi, o, pt);
            Model::a_quirk(pt.c);                                            // B is already formed from the true derivatives
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
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
            if constexpr (E == 4) {
                if (factored) {                                              // + cs n (n'(A dt) - n'w): the strong direction of W+
                    double W[E][E], Wq[NW], nn[4], cs = 0.0;
                    pi_block_W<E>(pt.B, W);
                    pinv_block_deflated<true>(pt.B, W, tolF, Wq, nn, &cs);
                    double nwv = 0.0;
#pragma unroll
                    for (int a = 0; a < E; ++a) {
                        double wa = -pt.f[a];
#pragma unroll
                        for (int k = 0; k < 6; ++k) wa -= pt.B[a][k] * (x.v[k] - o[k]);
                        nwv += nn[a] * (Ad[a] - wa);
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) rr[a] += nn[a] * (cs * nwv);
                }
            }
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
        obj = wave_sum(obj);
        diff = wave_sum(diff);
        const double dtk = (lane < u) ? g.dt[lane] : 0.0;
        const double ndt2 = wave_sum(dtk * dtk);
        if (dbg && lane == 0 && it <= 8) { dbg[96 + 3 * (it - 1)] = obj; dbg[97 + 3 * (it - 1)] = ndt2; dbg[98 + 3 * (it - 1)] = diff; }
        if (sqrt(ndt2) < GH_TOL && sqrt(diff) < GH_TOL) break;               // :71-73
        if (obj > objFunc) break;                                            // :75-76
        objFunc = obj;
        for (int i = lane; i < N; i += WAVE) {                               // xi = x + v; ti = ti + dt   (:80)
            const Pt6 x = premap(load_pt(pts, i), w->nrm);
#pragma unroll
            for (int k = 0; k < 6; ++k) g.xi[6 * i + k] = x.v[k] + g.pp[(long)PP * i + k];
        }
        if (lane < u) g.p[lane] += dtk;
        wave_sync();
    }
    return (it > GH_IT_MAX) ? GH_IT_MAX : it;
}

template <class Model, bool JAC>
__global__ void __launch_bounds__(64, 1) k_pi_tft_pose(const LinearTftArgs a) {
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
        const double* pts = a.corresp + b * 6 * (long)N;
        wave_sync();
        PiWork g = pi_carve(ghbase, Model::E, Model::C, a.spill ? 0 : N, Model::PINV_KKT);
        if (a.spill) { g.xi = a.spill + blockIdx.x * a.spill_stride; g.pp = g.xi + 6 * (long)N; }   // large N: per-correspondence state in global memory
        if (lane < 27) w->calm[lane] = a.calm[b * a.calm_stride + lane];
        int status = ST_OK, iters = 0;
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        if (N < 7) {
            status = ST_TOO_FEW;
        } else {
            normalise3(pts, N, w->nrm);                                      // PiPoseEstimation.m:53-56
            const bool ok = linear_tft_wave<JAC>(w, jw, pts, N, true, dbg);  // :59
            if (!ok) {
                status = ST_RETRY;
            } else {
                if (lane == 0) g.H[0] = (double)Model::init(w, g.p);
                wave_sync();
                const int ist = (int)g.H[0];
                if (ist != ST_OK) {
                    status = ist;
                } else {
                    tri_pass(w, pts, N, TRI_REPROJECT, 1, w->P[0], w->P[1], g.xi, w->nrm);   // x_est   (:80-83)
                    wave_sync();
                    int gst = ST_OK;
                    if (a.init_p) {                                          // debug/building-block output: the start of the iteration
                        if (lane < 27) a.init_p[b * 27 + lane] = g.p[lane];
                        for (int e = lane; e < 6 * N; e += WAVE) a.init_x[b * 6 * (long)N + e] = g.xi[e];
                    }
                    iters = gauss_helmert_pi_wave<Model>(w, g, pts, N, &gst, dbg, (a.flags & FLAG_GH_EXACT) != 0);  // :90
                    wave_sync();
                    if (lane == 0) Model::cameras(g.p, w);                   // :94-100
                    wave_sync();
                    tft_from_cameras(w, w->t);                               // T = TFT_from_P(P1,P2,P3)
                    wave_sync();
                    transform_tft_inverse(w->t, w->T1, w->Lp, [w](int v) { return normal_matrix(w->nrm, v); });   // :103
                    status = rt_from_tft_wave(w, pts, N, dbg);               // :106
                    if (gst != ST_OK) status = gst;
                    write_poses(w, a.Rt2 + b * 12, a.Rt3 + b * 12);
                    if (lane < 27) a.T[b * 27 + lane] = w->T1[lane];
                    if (a.reconst) final_reconst(w, pts, N, a.reconst + b * 3 * (long)N);   // :109-110
                    double chk = (lane < 12) ? w->Rt[0][lane] : ((lane < 24) ? w->Rt[1][lane - 12] : ((lane < 51) ? w->T1[lane - 24] : 0.0));
                    const bool bad = !(fabs(chk) <= 1.79e308);
                    if (wave_any(bad)) { if (status == ST_OK) status = ST_NONFINITE; wave_nan_outputs(a.Rt2, a.Rt3, a.T, a.reconst, b, N); }
                }
            }
        }
        if (status == ST_TOO_FEW || status == ST_NO_PARAM) {
            if (lane < 12) { a.Rt2[b * 12 + lane] = qnan; a.Rt3[b * 12 + lane] = qnan; }
            if (lane < 27) a.T[b * 27 + lane] = qnan;
            if (a.reconst) for (int i = lane; i < 3 * N; i += WAVE) a.reconst[b * 3 * (long)N + i] = qnan;
        }
        if (lane == 0) {
            if (a.iter) a.iter[b] = iters;
            a.status[b] = status;
        }
    }
}

}  // namespace tff
