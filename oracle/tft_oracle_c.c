/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of LinearTFTPoseEstimation
 * and LinearFPoseEstimation, used (a) as a second, LAPACK-free checker and
 * (b) as the "port" CPU baseline timed by bench.py on the GPU box's host cores.
 * Never linked into libtftfund.so; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg load it.
 *
 * PARITY UNPINNED (see oracle/tft_oracle.py): the MATLAB reference cannot run
 * here and has no golden vectors; this file is validated against the numpy /
 * LAPACK restatement and the committed fixtures (tests/test_oracle_c.py).
 *
 * It follows the reference's *algorithm*, not the GPU's: the 4N x 27 design
 * matrix is built row by row (linearTFT.m:36-62), every [~,~,V]=svd(.) is an
 * actual SVD (one-sided Jacobi, Hestenes) of that matrix, E (27x18) is formed
 * with Kronecker products and decomposed (linearTFT.m:82-86), every
 * triangulation is an SVD of the 2M x 4 system (triangulation3D.m:51-63), and
 * recover_R_t evaluates all four candidates (R_t_from_TFT.m:92-104).
 * Matrices are column-major like MATLAB.  File:line citations are into the
 * reference tree.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(r, c, ld) ((r) + (size_t)(c) * (ld))

/* ---- [U*S, V] = svd(A) by one-sided Jacobi; A is m x n (m >= n or not), column-major,
 * overwritten by U*diag(s); V n x n; columns sorted by descending singular value. ---- */
static void svd_jacobi(double* A, int m, int n, double* V, double* s) {
    for (int i = 0; i < n * n; ++i) V[i] = 0.0;
    for (int i = 0; i < n; ++i) V[IDX(i, i, n)] = 1.0;
    for (int sweep = 0; sweep < 80; ++sweep) {
        int rot = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double al = 0, be = 0, ga = 0;
                const double *ap = A + (size_t)p * m, *aq = A + (size_t)q * m;
                for (int i = 0; i < m; ++i) { al += ap[i] * ap[i]; be += aq[i] * aq[i]; ga += ap[i] * aq[i]; }
                if (!(fabs(ga) > 1e-16 * sqrt(al * be)) || ga == 0.0) continue;
                ++rot;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                double *bp = A + (size_t)p * m, *bq = A + (size_t)q * m;
                for (int i = 0; i < m; ++i) { double x = bp[i], y = bq[i]; bp[i] = c * x - sn * y; bq[i] = sn * x + c * y; }
                double *vp = V + (size_t)p * n, *vq = V + (size_t)q * n;
                for (int i = 0; i < n; ++i) { double x = vp[i], y = vq[i]; vp[i] = c * x - sn * y; vq[i] = sn * x + c * y; }
            }
        if (!rot) break;
    }
    for (int j = 0; j < n; ++j) { double a = 0; for (int i = 0; i < m; ++i) a += A[IDX(i, j, m)] * A[IDX(i, j, m)]; s[j] = sqrt(a); }
    for (int j = 0; j < n - 1; ++j) {                      /* selection sort, descending */
        int k = j;
        for (int l = j + 1; l < n; ++l) if (s[l] > s[k]) k = l;
        if (k != j) {
            double ts = s[j]; s[j] = s[k]; s[k] = ts;
            for (int i = 0; i < m; ++i) { double x = A[IDX(i, j, m)]; A[IDX(i, j, m)] = A[IDX(i, k, m)]; A[IDX(i, k, m)] = x; }
            for (int i = 0; i < n; ++i) { double x = V[IDX(i, j, n)]; V[IDX(i, j, n)] = V[IDX(i, k, n)]; V[IDX(i, k, n)] = x; }
        }
    }
}

/* last right singular vector of an m x n matrix (copy is decomposed) */
static void svd_last_v(const double* A, int m, int n, double* v, double* work /* m*n + n*n + n */) {
    double* B = work; double* V = work + (size_t)m * n; double* s = V + n * n;
    memcpy(B, A, sizeof(double) * (size_t)m * n);
    svd_jacobi(B, m, n, V, s);
    for (int i = 0; i < n; ++i) v[i] = V[IDX(i, n - 1, n)];
}

static void mat3_mul(const double* a, const double* b, double* c) {       /* column-major 3x3 */
    double t[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double x = 0; for (int k = 0; k < 3; ++k) x += a[IDX(i, k, 3)] * b[IDX(k, j, 3)]; t[IDX(i, j, 3)] = x; }
    memcpy(c, t, sizeof t);
}
static void mat3_T(const double* a, double* c) { double t[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[IDX(j, i, 3)] = a[IDX(i, j, 3)]; memcpy(c, t, sizeof t); }
static double mat3_det(const double* a) {
    return a[0] * (a[4] * a[8] - a[7] * a[5]) - a[3] * (a[1] * a[8] - a[7] * a[2]) + a[6] * (a[1] * a[5] - a[4] * a[2]);
}
static void mat3_inv(const double* a, double* c) {                          /* inv() */
    double d = mat3_det(a), t[9];
    t[0] = (a[4] * a[8] - a[7] * a[5]) / d; t[3] = -(a[3] * a[8] - a[6] * a[5]) / d; t[6] = (a[3] * a[7] - a[6] * a[4]) / d;
    t[1] = -(a[1] * a[8] - a[7] * a[2]) / d; t[4] = (a[0] * a[8] - a[6] * a[2]) / d; t[7] = -(a[0] * a[7] - a[6] * a[1]) / d;
    t[2] = (a[1] * a[5] - a[4] * a[2]) / d; t[5] = -(a[0] * a[5] - a[3] * a[2]) / d; t[8] = (a[0] * a[4] - a[3] * a[1]) / d;
    memcpy(c, t, sizeof t);
}
static double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

/* Normalize2Ddata.m:33-39: pts 2 x N (column-major, stride 6 between points of the 6xN Corresp) */
static void normalize2d(const double* corresp, int view, int N, double* xn /*2N*/, double* Nm /*3x3*/) {
    double cx = 0, cy = 0;
    for (int i = 0; i < N; ++i) { cx += corresp[6 * i + 2 * view]; cy += corresp[6 * i + 2 * view + 1]; }
    cx /= N; cy /= N;
    double d = 0;
    for (int i = 0; i < N; ++i) { double dx = corresp[6 * i + 2 * view] - cx, dy = corresp[6 * i + 2 * view + 1] - cy; d += sqrt(dx * dx + dy * dy); }
    d /= N;
    memset(Nm, 0, 9 * sizeof(double));
    Nm[0] = sqrt(2.0) / d; Nm[4] = sqrt(2.0) / d; Nm[8] = 1.0;
    Nm[6] = -sqrt(2.0) * cx / d; Nm[7] = -sqrt(2.0) * cy / d;
    for (int i = 0; i < N; ++i) {
        xn[2 * i] = Nm[0] * corresp[6 * i + 2 * view] + Nm[6];
        xn[2 * i + 1] = Nm[4] * corresp[6 * i + 2 * view + 1] + Nm[7];
    }
}

/* epipoles of a tensor (linearTFT.m:71-79, R_t_from_TFT.m:47-55) */
static void epipoles(const double* t, double* e21, double* e31, int fix_sign, double* work) {
    double v[9], M[9], x[3];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
                double val = t[j + 3 * k + 9 * i];
                if (pass == 0) M[IDX(j, k, 3)] = val; else M[IDX(k, j, 3)] = val;
            }
            svd_last_v(M, 3, 3, x, work);
            for (int k = 0; k < 3; ++k) v[IDX(i, k, 3)] = x[k];              /* [v1 v2 v3].' */
        }
        svd_last_v(v, 3, 3, x, work);
        if (fix_sign) { double sg = sgn(x[2]); x[0] *= sg; x[1] *= sg; x[2] *= sg; }
        memcpy(pass == 0 ? e31 : e21, x, 3 * sizeof(double));
    }
}

/* transform_TFT.m:42-49 (inverse = 1) */
static void transform_tft_inv(const double* to, const double* M1, const double* M2, const double* M3, double* tn) {
    double M2i[9], M3i[9], M3it[9], mix[9], tmp[9];
    mat3_inv(M2, M2i); mat3_inv(M3, M3i); mat3_T(M3i, M3it);
    for (int i = 0; i < 3; ++i) {
        for (int e = 0; e < 9; ++e) mix[e] = M1[IDX(0, i, 3)] * to[e] + M1[IDX(1, i, 3)] * to[9 + e] + M1[IDX(2, i, 3)] * to[18 + e];
        mat3_mul(M2i, mix, tmp); mat3_mul(tmp, M3it, tn + 9 * i);
    }
    double nn = 0; for (int e = 0; e < 27; ++e) nn += tn[e] * tn[e];
    nn = sqrt(nn); for (int e = 0; e < 27; ++e) tn[e] /= nn;
}

/* triangulation3D.m:51-63 for one point; P: M cameras 3x4 column-major; xy: 2M image coords */
static void triangulate1(const double* const* P, const double* xy, int M, double* X, double* work) {
    double ls[24];
    for (int i = 0; i < M; ++i) for (int c = 0; c < 4; ++c) {
        ls[IDX(2 * i, c, 2 * M)] = -P[i][IDX(1, c, 3)] + xy[2 * i + 1] * P[i][IDX(2, c, 3)];
        ls[IDX(2 * i + 1, c, 2 * M)] = P[i][IDX(0, c, 3)] - xy[2 * i] * P[i][IDX(2, c, 3)];
    }
    svd_last_v(ls, 2 * M, 4, X, work);
}

static void compose_cam(const double* K, const double* R, const double* t, double* P) {   /* K [R t], all column-major */
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { double x = 0; for (int k = 0; k < 3; ++k) x += K[IDX(r, k, 3)] * R[IDX(k, c, 3)]; P[IDX(r, c, 3)] = x; }
        double x = 0; for (int k = 0; k < 3; ++k) x += K[IDX(r, k, 3)] * t[k]; P[IDX(r, 3, 3)] = x;
    }
}

/* recover_R_t: R_t_from_TFT.m:82-106.  view = 1 or 2 selects columns (3:4) or (5:6) of Corresp. */
static int recover_R_t(const double* E, const double* K1, const double* K2, const double* corresp, int view, int N,
                       double* R_f, double* t_f, double* work) {
    double US[9], V[9], s[3], U[9];
    memcpy(US, E, sizeof US);
    svd_jacobi(US, 3, 3, V, s);
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 3; ++r) U[IDX(r, c, 3)] = US[IDX(r, c, 3)] / s[c];
    U[6] = U[1] * U[5] - U[2] * U[4]; U[7] = U[2] * U[3] - U[0] * U[5]; U[8] = U[0] * U[4] - U[1] * U[3];   /* u3 = u1 x u2 */
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};                        /* [0 -1 0; 1 0 0; 0 0 1] column-major */
    double Wt[9], Vt[9], R[9], Rp[9], tmp[9], t[3];
    mat3_T(W, Wt); mat3_T(V, Vt);
    mat3_mul(U, W, tmp); mat3_mul(tmp, Vt, R);
    mat3_mul(U, Wt, tmp); mat3_mul(tmp, Vt, Rp);
    double d = sgn(mat3_det(R)); for (int e = 0; e < 9; ++e) R[e] *= d;
    d = sgn(mat3_det(Rp)); for (int e = 0; e < 9; ++e) Rp[e] *= d;
    t[0] = U[6]; t[1] = U[7]; t[2] = U[8];
    double P1[12] = {0}, P2[12];
    memcpy(P1, K1, 9 * sizeof(double));
    const double* Ps[2] = {P1, P2};
    double seen = 0; int assigned = 0;
    for (int k = 1; k <= 4; ++k) {
        if (k == 2 || k == 4) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
        else if (k == 3) memcpy(R, Rp, sizeof R);
        compose_cam(K2, R, t, P2);
        double score = 0;
        for (int i = 0; i < N; ++i) {
            double xy[4] = {corresp[6 * i], corresp[6 * i + 1], corresp[6 * i + 2 * view], corresp[6 * i + 2 * view + 1]}, X[4];
            triangulate1(Ps, xy, 2, X, work);
            double x0 = X[0] / X[3], x1 = X[1] / X[3], x2 = X[2] / X[3];
            double z2 = R[IDX(2, 0, 3)] * x0 + R[IDX(2, 1, 3)] * x1 + R[IDX(2, 2, 3)] * x2 + t[2];
            score += sgn(x2) + sgn(z2);
        }
        if (score >= seen) { memcpy(R_f, R, 9 * sizeof(double)); memcpy(t_f, t, 3 * sizeof(double)); seen = score; assigned = 1; }
    }
    return assigned;
}

static void t3_scale_and_reconst(const double* K1, const double* K2, const double* K3, const double* R2, const double* t2,
                                 const double* R3, double* t3, const double* corresp, int N, double* reconst, double* work) {
    double P1[12] = {0}, P2[12], P3[12], u3[3], K3R3[9];
    memcpy(P1, K1, 9 * sizeof(double));
    compose_cam(K2, R2, t2, P2);
    for (int r = 0; r < 3; ++r) { u3[r] = 0; for (int k = 0; k < 3; ++k) u3[r] += K3[IDX(r, k, 3)] * t3[k]; }
    mat3_mul(K3, R3, K3R3);
    const double* Ps[3] = {P1, P2, P3};
    double num = 0, den = 0;
    for (int i = 0; i < N; ++i) {
        double X[4], X3[3], c1[3], c2[3];
        triangulate1(Ps, corresp + 6 * i, 2, X, work);
        double xd[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
        for (int r = 0; r < 3; ++r) X3[r] = K3R3[IDX(r, 0, 3)] * xd[0] + K3R3[IDX(r, 1, 3)] * xd[1] + K3R3[IDX(r, 2, 3)] * xd[2];
        double p[3] = {corresp[6 * i + 4], corresp[6 * i + 5], 1.0};
        c1[0] = p[1] * X3[2] - p[2] * X3[1]; c1[1] = p[2] * X3[0] - p[0] * X3[2]; c1[2] = p[0] * X3[1] - p[1] * X3[0];
        c2[0] = p[1] * u3[2] - p[2] * u3[1]; c2[1] = p[2] * u3[0] - p[0] * u3[2]; c2[2] = p[0] * u3[1] - p[1] * u3[0];
        num += c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
        den += c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
    }
    double lam = -num / den;
    t3[0] *= lam; t3[1] *= lam; t3[2] *= lam;
    if (reconst) {
        compose_cam(K3, R3, t3, P3);
        for (int i = 0; i < N; ++i) {
            double X[4];
            triangulate1(Ps, corresp + 6 * i, 3, X, work);
            reconst[3 * i] = X[0] / X[3]; reconst[3 * i + 1] = X[1] / X[3]; reconst[3 * i + 2] = X[2] / X[3];
        }
    }
}

static void write_pose(const double* R, const double* t, double* Rt) { memcpy(Rt, R, 9 * sizeof(double)); memcpy(Rt + 9, t, 3 * sizeof(double)); }

/* LinearTFTPoseEstimation.m:44-62 for one triplet.  Returns status (0 ok, 1 too few, 3 no pose). */
static int linear_tft_pose_one(const double* corresp, const double* calm, int N, double* Rt2, double* Rt3, double* Tout, double* reconst) {
    if (N < 7) return 1;
    const int m = 4 * N;
    double* A = (double*)calloc((size_t)m * 27, sizeof(double));
    double* AU = (double*)malloc(sizeof(double) * (size_t)m * 15);
    double* work = (double*)malloc(sizeof(double) * ((size_t)m * 27 + 27 * 27 + 27 + 64));
    double* xn = (double*)malloc(sizeof(double) * 6 * (size_t)N);
    double Nm[3][9];
    for (int v = 0; v < 3; ++v) normalize2d(corresp, v, N, xn + 2 * (size_t)N * v, Nm[v]);
    for (int i = 0; i < N; ++i) {                                             /* linearTFT.m:45-61 */
        double x1 = xn[2 * i], y1 = xn[2 * i + 1], x2 = xn[2 * N + 2 * i], y2 = xn[2 * N + 2 * i + 1], x3 = xn[4 * N + 2 * i], y3 = xn[4 * N + 2 * i + 1];
        double r1[27] = {x1, 0, -x1 * x2, 0, 0, 0, -x1 * x3, 0, x1 * x2 * x3, y1, 0, -x2 * y1, 0, 0, 0, -x3 * y1, 0, x2 * x3 * y1, 1, 0, -x2, 0, 0, 0, -x3, 0, x2 * x3};
        double r2[27] = {0, x1, -x1 * y2, 0, 0, 0, 0, -x1 * x3, x1 * x3 * y2, 0, y1, -y1 * y2, 0, 0, 0, 0, -x3 * y1, x3 * y1 * y2, 0, 1, -y2, 0, 0, 0, 0, -x3, x3 * y2};
        double r3[27] = {0, 0, 0, x1, 0, -x1 * x2, -x1 * y3, 0, x1 * x2 * y3, 0, 0, 0, y1, 0, -x2 * y1, -y1 * y3, 0, x2 * y1 * y3, 0, 0, 0, 1, 0, -x2, -y3, 0, x2 * y3};
        double r4[27] = {0, 0, 0, 0, x1, -x1 * y2, 0, -x1 * y3, x1 * y2 * y3, 0, 0, 0, 0, y1, -y1 * y2, 0, -y1 * y3, y1 * y2 * y3, 0, 0, 0, 0, 1, -y2, 0, -y3, y2 * y3};
        for (int c = 0; c < 27; ++c) { A[IDX(4 * i, c, m)] = r1[c]; A[IDX(4 * i + 1, c, m)] = r2[c]; A[IDX(4 * i + 2, c, m)] = r3[c]; A[IDX(4 * i + 3, c, m)] = r4[c]; }
    }
    double t[27], e21[3], e31[3];
    svd_last_v(A, m, 27, t, work);                                            /* :64-67 */
    epipoles(t, e21, e31, 0, work);                                           /* :71-79 */
    double E[27 * 18], US[27 * 18], VE[18 * 18], sE[18];
    memset(E, 0, sizeof E);                                                   /* :82 */
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) {
        E[IDX(j + 3 * k + 9 * i, j + 3 * i, 27)] = e31[k];                    /* kron(eye(3),kron(epi31,eye(3))) */
        E[IDX(j + 3 * k + 9 * i, 9 + k + 3 * i, 27)] = -e21[j];               /* -kron(eye(9),epi21) */
    }
    memcpy(US, E, sizeof E);
    svd_jacobi(US, 27, 18, VE, sE);                                           /* :83 */
    double tol = 27 * (nextafter(sE[0], INFINITY) - sE[0]);
    int rk = 0; while (rk < 18 && sE[rk] > tol) ++rk;
    if (rk > 15) rk = 15;
    double* Up = US;                                                          /* first rk columns, normalised */
    for (int c = 0; c < rk; ++c) for (int r = 0; r < 27; ++r) Up[IDX(r, c, 27)] /= sE[c];
    for (int c = 0; c < rk; ++c) for (int r = 0; r < m; ++r) { double x = 0; for (int k = 0; k < 27; ++k) x += A[IDX(r, k, m)] * Up[IDX(k, c, 27)]; AU[IDX(r, c, m)] = x; }
    double tp[18];
    svd_last_v(AU, m, rk, tp, work);                                          /* :84 */
    for (int r = 0; r < 27; ++r) { double x = 0; for (int c = 0; c < rk; ++c) x += Up[IDX(r, c, 27)] * tp[c]; t[r] = x; }   /* :85 */
    double T1[27], T2[27];
    transform_tft_inv(t, Nm[0], Nm[1], Nm[2], T1);                            /* LinearTFTPoseEstimation.m:53 */
    memcpy(Tout, T1, sizeof T1);
    double Kc[3][9];                                                          /* CalM is 9x3 column-major */
    for (int v = 0; v < 3; ++v) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Kc[v][IDX(r, c, 3)] = calm[(3 * v + r) + 9 * c];
    const double *K1 = Kc[0], *K2 = Kc[1], *K3 = Kc[2];
    transform_tft_inv(T1, K1, K2, K3, T2);                                    /* R_t_from_TFT.m:44 */
    epipoles(T2, e21, e31, 1, work);
    double M[9], E21[9], E31[9], cm[9];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 3; ++r) { double x = 0; for (int k = 0; k < 3; ++k) x += T2[r + 3 * k + 9 * i] * e31[k]; M[IDX(r, i, 3)] = x; }
    cm[0] = 0; cm[3] = -e21[2]; cm[6] = e21[1]; cm[1] = e21[2]; cm[4] = 0; cm[7] = -e21[0]; cm[2] = -e21[1]; cm[5] = e21[0]; cm[8] = 0;
    mat3_mul(cm, M, E21);                                                     /* :57 */
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 3; ++r) { double x = 0; for (int j = 0; j < 3; ++j) x += T2[j + 3 * r + 9 * i] * e21[j]; M[IDX(r, i, 3)] = x; }
    cm[0] = 0; cm[3] = -e31[2]; cm[6] = e31[1]; cm[1] = e31[2]; cm[4] = 0; cm[7] = -e31[0]; cm[2] = -e31[1]; cm[5] = e31[0]; cm[8] = 0;
    mat3_mul(cm, M, E31); for (int e = 0; e < 9; ++e) E31[e] = -E31[e];       /* :58 */
    double R2[9], t2[3], R3[9], t3[3];
    int ok = recover_R_t(E21, K1, K2, corresp, 1, N, R2, t2, work) && recover_R_t(E31, K1, K3, corresp, 2, N, R3, t3, work);
    int status = 0;
    if (!ok) status = 3;
    else {
        t3_scale_and_reconst(K1, K2, K3, R2, t2, R3, t3, corresp, N, reconst, work);
        write_pose(R2, t2, Rt2); write_pose(R3, t3, Rt3);
    }
    free(A); free(AU); free(work); free(xn);
    return status;
}

/* linearF.m:32-62 on already (outer-)normalised points xa, xb (2 x N each, column-major): normalises again (:45-46), N x 9
 * DLT (:48-55), inner de-normalisation (:58), rank-2 projection by zeroing the third singular value (:61-62).  F column-major. */
static void normalize2d_pts(const double* x, int N, double* xn, double* Nm) {   /* Normalize2Ddata.m:33-39 on a 2 x N array */
    double cx = 0, cy = 0, d = 0;
    for (int i = 0; i < N; ++i) { cx += x[2 * i]; cy += x[2 * i + 1]; }
    cx /= N; cy /= N;
    for (int i = 0; i < N; ++i) { double dx = x[2 * i] - cx, dy = x[2 * i + 1] - cy; d += sqrt(dx * dx + dy * dy); }
    d /= N;
    const double s = sqrt(2.0) / d;
    memset(Nm, 0, 9 * sizeof(double));
    Nm[IDX(0, 0, 3)] = s; Nm[IDX(1, 1, 3)] = s; Nm[IDX(0, 2, 3)] = -sqrt(2.0) * cx / d; Nm[IDX(1, 2, 3)] = -sqrt(2.0) * cy / d; Nm[IDX(2, 2, 3)] = 1.0;
    for (int i = 0; i < N; ++i) { xn[2 * i] = Nm[IDX(0, 0, 3)] * x[2 * i] + Nm[IDX(0, 2, 3)]; xn[2 * i + 1] = Nm[IDX(1, 1, 3)] * x[2 * i + 1] + Nm[IDX(1, 2, 3)]; }
}
static void linear_f(const double* xa, const double* xb, int N, double* F, double* work) {
    double* p1 = (double*)malloc(sizeof(double) * 4 * (size_t)N); double* p2 = p1 + 2 * (size_t)N;
    double N1[9], N2[9];
    normalize2d_pts(xa, N, p1, N1);
    normalize2d_pts(xb, N, p2, N2);
    const int m = N, n = 9;
    double* A = (double*)malloc(sizeof(double) * (size_t)((m < n ? n : m)) * n);
    /* with fewer rows than columns (N = 8) MATLAB's full svd still returns a 9 x 9 V whose last column spans the null space:
     * pad with zero rows so that the one-sided Jacobi sees a 9 x 9 matrix */
    const int mm = m < n ? n : m;
    memset(A, 0, sizeof(double) * (size_t)mm * n);
    for (int i = 0; i < N; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const double r[9] = {x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, 1.0};      /* :51-52 */
        for (int c = 0; c < 9; ++c) A[IDX(i, c, mm)] = r[c];
    }
    double v[9], Fn[9], tmp[9], N2t[9];
    svd_last_v(A, mm, n, v, work);                                              /* :54-55: F = reshape(V(:,9),3,3) */
    memcpy(Fn, v, sizeof Fn);
    mat3_T(N2, N2t); mat3_mul(N2t, Fn, tmp); mat3_mul(tmp, N1, Fn);               /* :58 */
    double US[9], V[9], sv[3], Vt[9];
    memcpy(US, Fn, sizeof US);
    svd_jacobi(US, 3, 3, V, sv);                                                /* :61-62: U diag(s1,s2,0) V' */
    for (int r = 0; r < 3; ++r) US[IDX(r, 2, 3)] = 0.0;
    mat3_T(V, Vt); mat3_mul(US, Vt, F);
    free(p1); free(A);
}

/* TFT_from_P.m:25-33: T(j,k,i) = (-1)^(i+1) det[P1 without row i; P2(j,:); P3(k,:)], unit Frobenius norm.  P column-major 3x4. */
static double det4(const double m[4][4]) {
    double d = 0;
    for (int c = 0; c < 4; ++c) {
        double s[3][3];
        for (int r = 1; r < 4; ++r) { int cc = 0; for (int k = 0; k < 4; ++k) if (k != c) s[r - 1][cc++] = m[r][k]; }
        const double d3 = s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0]) + s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]);
        d += ((c & 1) ? -1.0 : 1.0) * m[0][c] * d3;
    }
    return d;
}
static void tft_from_P(const double* P1, const double* P2, const double* P3, double* T) {
    double nn = 0;
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) {
        double m[4][4]; int rr = 0;
        for (int r = 0; r < 3; ++r) if (r != i) { for (int c = 0; c < 4; ++c) m[rr][c] = P1[IDX(r, c, 3)]; ++rr; }
        for (int c = 0; c < 4; ++c) { m[2][c] = P2[IDX(j, c, 3)]; m[3][c] = P3[IDX(k, c, 3)]; }
        const double v = ((i & 1) ? -1.0 : 1.0) * det4(m);
        T[j + 3 * k + 9 * i] = v; nn += v * v;
    }
    nn = sqrt(nn);
    for (int e = 0; e < 27; ++e) T[e] /= nn;
}

/* LinearFPoseEstimation.m:42-78 for one triplet.  Returns status (0 ok, 1 too few, 3 no pose). */
static int linear_f_pose_one(const double* corresp, const double* calm, int N, double* Rt2, double* Rt3, double* Tout, double* reconst) {
    if (N < 8) return 1;
    double* work = (double*)malloc(sizeof(double) * ((size_t)(N < 9 ? 9 : N) * 9 + 81 + 9 + 64));
    double* xn = (double*)malloc(sizeof(double) * 6 * (size_t)N);
    double Nm[3][9];
    for (int v = 0; v < 3; ++v) normalize2d(corresp, v, N, xn + 2 * (size_t)N * v, Nm[v]);   /* :46-48 */
    double F21[9], F31[9], tmp[9], Nt[9];
    linear_f(xn, xn + 2 * (size_t)N, N, F21, work);                              /* :51-52 */
    linear_f(xn, xn + 4 * (size_t)N, N, F31, work);
    mat3_T(Nm[1], Nt); mat3_mul(Nt, F21, tmp); mat3_mul(tmp, Nm[0], F21);       /* :55-56 */
    mat3_T(Nm[2], Nt); mat3_mul(Nt, F31, tmp); mat3_mul(tmp, Nm[0], F31);
    double Kc[3][9];
    for (int v = 0; v < 3; ++v) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Kc[v][IDX(r, c, 3)] = calm[(3 * v + r) + 9 * c];
    const double *K1 = Kc[0], *K2 = Kc[1], *K3 = Kc[2];
    double E21[9], E31[9], Kt[9];
    mat3_T(K2, Kt); mat3_mul(Kt, F21, tmp); mat3_mul(tmp, K1, E21);              /* recover_R_t: E = K2' F K1   (:86) */
    mat3_T(K3, Kt); mat3_mul(Kt, F31, tmp); mat3_mul(tmp, K1, E31);
    double R2[9], t2[3], R3[9], t3[3];
    const int ok = recover_R_t(E21, K1, K2, corresp, 1, N, R2, t2, work) && recover_R_t(E31, K1, K3, corresp, 2, N, R3, t3, work);
    int status = 0;
    if (!ok) status = 3;
    else {
        t3_scale_and_reconst(K1, K2, K3, R2, t2, R3, t3, corresp, N, reconst, work);   /* :64-76 */
        write_pose(R2, t2, Rt2); write_pose(R3, t3, Rt3);
        double P1[12] = {0}, P2[12], P3[12];
        memcpy(P1, K1, 9 * sizeof(double));
        compose_cam(K2, R2, t2, P2); compose_cam(K3, R3, t3, P3);
        tft_from_P(P1, P2, P3, Tout);                                           /* :78 */
    }
    free(work); free(xn);
    return status;
}

/* Batched entry point with the layout of include/tftfund.h.  threads <= 0: all cores. Returns threads used. */
int oracle_c_linear_tft_pose_batch(const double* corresp, const double* calm, long calm_stride, long B, int N,
                                   double* Rt2, double* Rt3, double* T, double* reconst, int* status, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long b = 0; b < B; ++b) {
        int st = linear_tft_pose_one(corresp + b * 6 * (long)N, calm + b * calm_stride, N, Rt2 + b * 12, Rt3 + b * 12, T + b * 27,
                                     reconst ? reconst + b * 3 * (long)N : 0);
        if (status) status[b] = st;
    }
    return used;
}

int oracle_c_linear_f_pose_batch(const double* corresp, const double* calm, long calm_stride, long B, int N,
                                 double* Rt2, double* Rt3, double* T, double* reconst, int* status, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long b = 0; b < B; ++b) {
        int st = linear_f_pose_one(corresp + b * 6 * (long)N, calm + b * calm_stride, N, Rt2 + b * 12, Rt3 + b * 12, T + b * 27,
                                   reconst ? reconst + b * 3 * (long)N : 0);
        if (status) status[b] = st;
    }
    return used;
}
