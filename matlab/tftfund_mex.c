/*
 * tftfund_mex.c -- MEX gateway from MATLAB to libtftfund.so (C ABI in include/tftfund.h).
 *
 *   [R_t_2, R_t_3, Reconst, T, iter] = tftfund_mex(method, Corresp, CalM)
 *
 * method : 'linear_tft' | 'linear_f' | 'ressl_tft' | 'nordberg_tft' | 'faugpapa_tft' | 'optim_f' | 'pi' | 'picol'
 *          (one entry per tff_<method>_pose_batch_host symbol)
 * Corresp: 6 x N double, or 6 x N x B for a batch of B triplets
 * CalM   : 9 x 3 double (shared) or 9 x 3 x B
 * Outputs follow the reference's calling convention (experiments.m:108):
 *   R_t_2, R_t_3  3 x 4 (x B),  Reconst 3 x N (x B),  T 3 x 3 x 3 (x B),  iter 1 x B double.
 * Fewer outputs may be requested (example.m:42 takes three; experiments_real.m:126 skips
 * Reconst and T) -- Reconst is only computed when nlhs >= 3.
 *
 * Per-triplet status codes become MATLAB errors for B = 1 (mirroring error() in linearF.m:36 and
 * the unassigned-output error of R_t_from_TFT.m:101); for B > 1 failed triplets come back as NaN.
 *
 * Build (where MATLAB's mex and ROCm are installed; cannot be built in the GPU-less CI image):
 *   mex -I../include tftfund_mex.c -L../tft_vs_fund_amd -ltftfund -lamdhip64
 * The three wrappers LinearTFTPoseEstimation.m / LinearFPoseEstimation.m in this directory keep the
 * reference's names and signatures; put this directory ahead of the reference on the MATLAB path.
 */
#include <string.h>
#include "mex.h"
#include "tftfund.h"

static tff_ctx* g_ctx = NULL;

static void cleanup(void) {
    if (g_ctx) { tff_ctx_destroy(g_ctx); g_ctx = NULL; }
}

typedef int (*pose_host_fn)(tff_ctx*, const double*, const double*, int64_t, int64_t, int32_t, double*, double*, double*,
                            double*, int32_t*, int32_t*);

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char method[32];
    pose_host_fn fn = NULL;
    const mwSize* dc;
    mwSize ndc, N, B, b;
    int64_t calm_stride;
    int rc;
    int32_t *iter, *status;
    double *Rt2, *Rt3, *T, *Rec = NULL;
    mxArray *aRt2, *aRt3, *aT, *aRec = NULL;

    if (nrhs != 3) mexErrMsgIdAndTxt("tftfund:nargin", "usage: tftfund_mex(method, Corresp, CalM)");
    if (nlhs > 5) mexErrMsgIdAndTxt("tftfund:nargout", "at most five outputs");
    if (mxGetString(prhs[0], method, sizeof method)) mexErrMsgIdAndTxt("tftfund:method", "method must be a string");
    if (!strcmp(method, "linear_tft")) fn = tff_linear_tft_pose_batch_host;
    else if (!strcmp(method, "linear_f")) fn = tff_linear_f_pose_batch_host;
    else if (!strcmp(method, "ressl_tft")) fn = tff_ressl_tft_pose_batch_host;
    else if (!strcmp(method, "nordberg_tft")) fn = tff_nordberg_tft_pose_batch_host;
    else if (!strcmp(method, "faugpapa_tft")) fn = tff_faugpapa_tft_pose_batch_host;
    else if (!strcmp(method, "optim_f")) fn = tff_optim_f_pose_batch_host;
    else if (!strcmp(method, "pi")) fn = tff_pi_pose_batch_host;
    else if (!strcmp(method, "picol")) fn = tff_picol_pose_batch_host;
    else mexErrMsgIdAndTxt("tftfund:method", "unknown method '%s'", method);
    if (!mxIsDouble(prhs[1]) || mxIsComplex(prhs[1]) || !mxIsDouble(prhs[2]) || mxIsComplex(prhs[2]))
        mexErrMsgIdAndTxt("tftfund:type", "Corresp and CalM must be real double");
    ndc = mxGetNumberOfDimensions(prhs[1]);
    dc = mxGetDimensions(prhs[1]);
    if (dc[0] != 6 || ndc > 3) mexErrMsgIdAndTxt("tftfund:shape", "Corresp must be 6 x N (x B)");
    N = dc[1];
    B = (ndc == 3) ? dc[2] : 1;
    {
        const mwSize* dk = mxGetDimensions(prhs[2]);
        mwSize ndk = mxGetNumberOfDimensions(prhs[2]);
        if (dk[0] != 9 || dk[1] != 3) mexErrMsgIdAndTxt("tftfund:shape", "CalM must be 9 x 3 (x B)");
        calm_stride = (ndk == 3 && dk[2] == B && B > 1) ? 27 : 0;
    }
    if (!g_ctx) {
        if ((rc = tff_ctx_create(&g_ctx, 0)) != 0) mexErrMsgIdAndTxt("tftfund:hip", "tff_ctx_create: %s", tff_last_error());
        mexAtExit(cleanup);
    }
    {
        mwSize d34[3] = {3, 4, 0}, d333[4] = {3, 3, 3, 0}, d3n[3] = {3, 0, 0};
        d34[2] = B; d333[3] = B; d3n[1] = N; d3n[2] = B;
        aRt2 = mxCreateNumericArray(B > 1 ? 3 : 2, d34, mxDOUBLE_CLASS, mxREAL);
        aRt3 = mxCreateNumericArray(B > 1 ? 3 : 2, d34, mxDOUBLE_CLASS, mxREAL);
        aT = mxCreateNumericArray(B > 1 ? 4 : 3, d333, mxDOUBLE_CLASS, mxREAL);
        if (nlhs >= 3) aRec = mxCreateNumericArray(B > 1 ? 3 : 2, d3n, mxDOUBLE_CLASS, mxREAL);
    }
    Rt2 = mxGetPr(aRt2); Rt3 = mxGetPr(aRt3); T = mxGetPr(aT);
    if (aRec) Rec = mxGetPr(aRec);
    iter = (int32_t*)mxCalloc(B ? B : 1, sizeof(int32_t));
    status = (int32_t*)mxCalloc(B ? B : 1, sizeof(int32_t));
    /* MATLAB's column-major 6 x N x B and 9 x 3 (x B) arrays are exactly the C ABI's layout */
    rc = fn(g_ctx, mxGetPr(prhs[1]), mxGetPr(prhs[2]), calm_stride, (int64_t)B, (int32_t)N, Rt2, Rt3, T, Rec, iter, status);
    if (rc != 0) mexErrMsgIdAndTxt("tftfund:hip", "%s: %s", method, tff_last_error());
    if (B == 1 && status[0] == TFF_ST_TOO_FEW)
        mexErrMsgIdAndTxt("tftfund:tooFew", "not enough correspondences for %s (N = %d)", method, (int)N);
    if (B == 1 && status[0] == TFF_ST_NO_PARAM)
        mexErrMsgIdAndTxt("tftfund:noParam", "The minimal param could not be found");
    if (B == 1 && status[0] == TFF_ST_NO_POSE)
        mexErrMsgIdAndTxt("tftfund:noPose", "no pose candidate with non-negative cheirality score");
    plhs[0] = aRt2;
    if (nlhs >= 2) plhs[1] = aRt3; else mxDestroyArray(aRt3);
    if (nlhs >= 3) plhs[2] = aRec;
    if (nlhs >= 4) plhs[3] = aT; else mxDestroyArray(aT);
    if (nlhs >= 5) {
        plhs[4] = mxCreateDoubleMatrix(1, B, mxREAL);
        for (b = 0; b < B; ++b) mxGetPr(plhs[4])[b] = (double)iter[b];
    }
    mxFree(iter); mxFree(status);
}
