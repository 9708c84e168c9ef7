/* Plain-C user of the C ABI (include/tftfund.h): the binding a C/C++ host program -- or the MEX shim -- makes.
 *   gcc tests/c_driver.c -Iinclude -Ltft_vs_fund_amd -ltftfund -lm -o c_driver && LD_LIBRARY_PATH=tft_vs_fund_amd ./c_driver
 * Builds a noise-free synthetic triplet batch (three cameras looking at the origin, points in a cube), runs
 * LinearTFTPoseEstimation and ResslTFTPoseEstimation through the _host entry points, and checks the recovered
 * poses against the ground truth, the too-few-points status and the error path.  Exit code 0 = all good. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tftfund.h"

static void matmul3(const double* A, const double* B, double* C) {        /* row-major 3x3 */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}
static void rot(double ax, double ay, double az, double* R) {
    const double Rx[9] = {1, 0, 0, 0, cos(ax), -sin(ax), 0, sin(ax), cos(ax)}, Ry[9] = {cos(ay), 0, sin(ay), 0, 1, 0, -sin(ay), 0, cos(ay)},
                 Rz[9] = {cos(az), -sin(az), 0, sin(az), cos(az), 0, 0, 0, 1};
    double t[9];
    matmul3(Rx, Ry, t); matmul3(t, Rz, R);
}
static unsigned long long rng = 88172645463325252ULL;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }

int main(void) {
    enum { B = 3, N = 40 };
    const double K[9] = {2500, 0, 900, 0, 2500, 600, 0, 0, 1};
    double calm[27];                                                       /* 9x3 column-major: [K;K;K] */
    for (int v = 0; v < 3; ++v) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) calm[(3 * v + r) + 9 * c] = K[3 * r + c];
    double R2[9], R3[9];
    rot(0.05, -0.25, 0.02, R2); rot(-0.04, 0.30, -0.03, R3);
    const double t2[3] = {400, 30, 60}, t3[3] = {-500, -20, 90};
    double* corresp = malloc(sizeof(double) * B * 6 * N);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const double X[3] = {400 * urand() - 200, 400 * urand() - 200, 1500 + 400 * urand()};
            const double* Rs[3] = {NULL, R2, R3}; const double* ts[3] = {NULL, t2, t3};
            for (int v = 0; v < 3; ++v) {
                double Y[3];
                for (int i = 0; i < 3; ++i) Y[i] = v ? Rs[v][3 * i] * X[0] + Rs[v][3 * i + 1] * X[1] + Rs[v][3 * i + 2] * X[2] + ts[v][i] : X[i];
                corresp[(b * N + n) * 6 + 2 * v] = K[0] * Y[0] / Y[2] + K[2];
                corresp[(b * N + n) * 6 + 2 * v + 1] = K[4] * Y[1] / Y[2] + K[5];
            }
        }
    tff_ctx* ctx = NULL;
    if (tff_ctx_create(&ctx, 0) != 0) { fprintf(stderr, "tff_ctx_create: %s\n", tff_last_error()); return 2; }
    double Rt2[B * 12], Rt3[B * 12], T[B * 27], rec[B * 3 * N];
    int32_t iter[B], status[B];
    int fails = 0;
    int (*methods[2])(tff_ctx*, const double*, const double*, int64_t, int64_t, int32_t, double*, double*, double*, double*, int32_t*, int32_t*) =
        {tff_linear_tft_pose_batch_host, tff_ressl_tft_pose_batch_host};
    const char* names[2] = {"LinearTFTPoseEstimation", "ResslTFTPoseEstimation"};
    const double nt2 = sqrt(t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]);
    for (int m = 0; m < 2; ++m) {
        if (methods[m](ctx, corresp, calm, 0, B, N, Rt2, Rt3, T, rec, iter, status) != 0) { fprintf(stderr, "%s: %s\n", names[m], tff_last_error()); return 3; }
        double worst = 0;
        for (int b = 0; b < B; ++b) {
            if (status[b] != TFF_ST_OK) ++fails;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {                              /* outputs are 3x4 column-major, |t2| = 1 */
                    const double g2 = (c < 3) ? R2[3 * r + c] : t2[r] / nt2, g3 = (c < 3) ? R3[3 * r + c] : t3[r] / nt2;
                    worst = fmax(worst, fabs(Rt2[b * 12 + r + 3 * c] - g2));
                    worst = fmax(worst, fabs(Rt3[b * 12 + r + 3 * c] - g3));
                }
        }
        printf("%-26s max |pose - ground truth| = %.2e, iter[0] = %d\n", names[m], worst, iter[0]);
        if (!(worst < 1e-7)) ++fails;
    }
    /* too few correspondences -> per-triplet status, NaN outputs, call itself succeeds */
    if (tff_linear_tft_pose_batch_host(ctx, corresp, calm, 0, 1, 6, Rt2, Rt3, T, NULL, NULL, status) != 0 || status[0] != TFF_ST_TOO_FEW || Rt2[0] == Rt2[0]) ++fails;
    /* invalid arguments -> error code + message */
    if (tff_linear_tft_pose_batch_host(ctx, NULL, calm, 0, 1, 8, Rt2, Rt3, T, NULL, NULL, status) == 0 || strlen(tff_last_error()) == 0) ++fails;
    tff_ctx_destroy(ctx);
    /* multi-GPU entry point over all visible devices (one host thread + stream per device; one device: a clique of one) */
    {
        tff_multi* mg = NULL;
        if (tff_multi_create(&mg, NULL, 0) != 0) { fprintf(stderr, "tff_multi_create: %s\n", tff_last_error()); return 4; }
        double mRt2[B * 12], mRt3[B * 12], mT[B * 27];
        int32_t mst[B];
        if (tff_pose_batch_host_multi(mg, TFF_METHOD_LINEAR_TFT, corresp, calm, 0, B, N, mRt2, mRt3, mT, NULL, NULL, mst) != 0) {
            fprintf(stderr, "tff_pose_batch_host_multi: %s\n", tff_last_error()); return 5;
        }
        double worst = 0;
        for (int b = 0; b < B; ++b) {
            if (mst[b] != TFF_ST_OK) ++fails;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) worst = fmax(worst, fabs(mRt2[b * 12 + r + 3 * c] - R2[3 * r + c]));
        }
        int64_t b0, b1;
        tff_multi_shard(mg, B, tff_multi_size(mg) - 1, &b0, &b1);
        printf("multi (%d device%s)            max |R2 - ground truth| = %.2e, last shard [%lld, %lld)\n", (int)tff_multi_size(mg),
               tff_multi_size(mg) == 1 ? "" : "s", worst, (long long)b0, (long long)b1);
        if (!(worst < 1e-7) || b1 != B) ++fails;
        tff_multi_destroy(mg);
    }
    free(corresp);
    printf(fails ? "FAILED (%d)\n" : "c_driver ok\n", fails);
    return fails ? 1 : 0;
}
