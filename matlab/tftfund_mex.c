/*
 * tftfund_mex.c -- MEX gateway from MATLAB to libtftfund.so (C ABI in include/tftfund.h).
 *
 *   [R_t_2, R_t_3, Reconst, T, iter] = tftfund_mex(method, Corresp, CalM [, devices])
 *
 * method : 'linear_tft' | 'linear_f' | 'ressl_tft' | 'nordberg_tft' | 'faugpapa_tft' | 'optim_f' | 'pi' | 'picol'
 *          (one entry per tff_<method>_pose_batch_host symbol)
 * Corresp: 6 x N double, or 6 x N x B for a batch of B triplets
 * CalM   : 9 x 3 double (shared) or 9 x 3 x B
 * devices: optional row vector of HIP device ordinals (0-based).  Given, the batch is cut into contiguous shards, one per
 *          device, each on its own host thread and stream (tff_pose_batch_host_multi: independent triplets, no collective);
 *          omitted: device 0.
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
 * The nine one-call wrappers in this directory (the eight *PoseEstimation.m methods of experiments.m:51-59 and
 * BundleAdjustment.m) keep the
 * reference's names and signatures; put this directory ahead of the reference on the MATLAB path.
 */
#include <string.h>
#include "mex.h"
#include "tftfund.h"

static tff_ctx* g_ctx = NULL;
static tff_multi* g_multi = NULL;      /* multi-GPU handle for the device list last used */
static int32_t g_multi_dev[64];
static int32_t g_multi_n = 0;

static void cleanup(void) {
    if (g_ctx) { tff_ctx_destroy(g_ctx); g_ctx = NULL; }
    if (g_multi) { tff_multi_destroy(g_multi); g_multi = NULL; g_multi_n = 0; }
}

typedef int (*pose_host_fn)(tff_ctx*, const double*, const double*, int64_t, int64_t, int32_t, double*, double*, double*,
                            double*, int32_t*, int32_t*);

/* [R_t, Reconst, iter, repr_err] = tftfund_mex('bundle_adjustment', Corresp (2M x N), CalM (3M x 3), R_t_0 (3M x 4) [, Reconst0 (3 x N)]),
 * M = 2 .. 6: the arrays go to tff_bundle_adjust_views_batch_host as they are (it takes MATLAB's own layouts). */
static void bundle_adjustment(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    double err = 0;
    int32_t it = 0, st = 0;
    mwSize N, M;
    mxArray* rec;
    if (nrhs < 4 || nrhs > 5 || nlhs > 4) mexErrMsgIdAndTxt("tftfund:nargin", "usage: tftfund_mex('bundle_adjustment', Corresp, CalM, R_t_0 [, Reconst0])");
    if (!mxIsDouble(prhs[1]) || mxGetM(prhs[1]) % 2 || mxGetM(prhs[1]) < 4 || mxGetM(prhs[1]) > 12) mexErrMsgIdAndTxt("tftfund:shape", "Corresp must be 2M x N, M = 2 .. 6 views");
    M = mxGetM(prhs[1]) / 2;
    N = mxGetN(prhs[1]);
    if (!mxIsDouble(prhs[2]) || mxGetM(prhs[2]) != 3 * M || mxGetN(prhs[2]) != 3) mexErrMsgIdAndTxt("tftfund:shape", "CalM must be 3M x 3");
    if (!mxIsDouble(prhs[3]) || mxGetM(prhs[3]) != 3 * M || mxGetN(prhs[3]) != 4) mexErrMsgIdAndTxt("tftfund:shape", "R_t_0 must be 3M x 4");
    if (nrhs == 5 && (!mxIsDouble(prhs[4]) || mxGetM(prhs[4]) != 3 || mxGetN(prhs[4]) != N)) mexErrMsgIdAndTxt("tftfund:shape", "Reconst0 must be 3 x N");
    if (!g_ctx) {
        if (tff_ctx_create(&g_ctx, 0) != 0) mexErrMsgIdAndTxt("tftfund:hip", "tff_ctx_create: %s", tff_last_error());
        mexAtExit(cleanup);
    }
    plhs[0] = mxCreateDoubleMatrix(3 * M, 4, mxREAL);
    rec = mxCreateDoubleMatrix(3, N, mxREAL);
    if (tff_bundle_adjust_views_batch_host(g_ctx, (int32_t)M, mxGetPr(prhs[2]), 0, mxGetPr(prhs[3]), mxGetPr(prhs[1]), 1, (int32_t)N,
                                           nrhs == 5 ? mxGetPr(prhs[4]) : NULL, mxGetPr(plhs[0]), mxGetPr(rec), &it, &err, &st) != 0)
        mexErrMsgIdAndTxt("tftfund:hip", "bundle_adjustment: %s", tff_last_error());
    if (st == TFF_ST_TOO_FEW)                                    /* the reference stops here too: triangulation3D.m:36-38 returns nothing, BundleAdjustment.m:73-74 */
        mexErrMsgIdAndTxt("tftfund:views", "fewer than two complete views to triangulate from");
    if (nlhs >= 2) plhs[1] = rec; else mxDestroyArray(rec);
    if (nlhs >= 3) plhs[2] = mxCreateDoubleScalar((double)it);
    if (nlhs >= 4) plhs[3] = mxCreateDoubleScalar(err);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char method[32];
    pose_host_fn fn = NULL;
    int32_t method_id = -1;
    const mwSize* dc;
    mwSize ndc, N, B, b;
    int64_t calm_stride;
    int rc;
    int32_t *iter, *status;
    double *Rt2, *Rt3, *T, *Rec = NULL;
    mxArray *aRt2, *aRt3, *aT, *aRec = NULL;

    if (nrhs >= 1 && mxIsChar(prhs[0]) && !mxGetString(prhs[0], method, sizeof method) && !strcmp(method, "bundle_adjustment")) {
        bundle_adjustment(nlhs, plhs, nrhs, prhs);
        return;
    }
    if (nrhs != 3 && nrhs != 4) mexErrMsgIdAndTxt("tftfund:nargin", "usage: tftfund_mex(method, Corresp, CalM [, devices])");
    if (nlhs > 5) mexErrMsgIdAndTxt("tftfund:nargout", "at most five outputs");
    if (mxGetString(prhs[0], method, sizeof method)) mexErrMsgIdAndTxt("tftfund:method", "method must be a string");
    if (!strcmp(method, "linear_tft")) { fn = tff_linear_tft_pose_batch_host; method_id = TFF_METHOD_LINEAR_TFT; }
    else if (!strcmp(method, "linear_f")) { fn = tff_linear_f_pose_batch_host; method_id = TFF_METHOD_LINEAR_F; }
    else if (!strcmp(method, "ressl_tft")) { fn = tff_ressl_tft_pose_batch_host; method_id = TFF_METHOD_RESSL_TFT; }
    else if (!strcmp(method, "nordberg_tft")) { fn = tff_nordberg_tft_pose_batch_host; method_id = TFF_METHOD_NORDBERG_TFT; }
    else if (!strcmp(method, "faugpapa_tft")) { fn = tff_faugpapa_tft_pose_batch_host; method_id = TFF_METHOD_FAUGPAPA_TFT; }
    else if (!strcmp(method, "optim_f")) { fn = tff_optim_f_pose_batch_host; method_id = TFF_METHOD_OPTIM_F; }
    else if (!strcmp(method, "pi")) { fn = tff_pi_pose_batch_host; method_id = TFF_METHOD_PI; }
    else if (!strcmp(method, "picol")) { fn = tff_picol_pose_batch_host; method_id = TFF_METHOD_PICOL; }
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
        if (ndk == 3 && dk[2] != 1 && dk[2] != B) mexErrMsgIdAndTxt("tftfund:shape", "CalM is 9 x 3 x K: K must be 1 or the batch size");
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
    if (nrhs == 4) {                                                   /* device list: shard the batch over several GPUs */
        const mwSize nd = mxGetNumberOfElements(prhs[3]);
        int32_t dev[64];
        mwSize k;
        int same;
        if (!mxIsDouble(prhs[3]) || nd < 1 || nd > 64) mexErrMsgIdAndTxt("tftfund:devices", "devices must be a vector of 1..64 device ordinals");
        for (k = 0; k < nd; ++k) dev[k] = (int32_t)mxGetPr(prhs[3])[k];
        same = g_multi && g_multi_n == (int32_t)nd;
        for (k = 0; same && k < nd; ++k) same = dev[k] == g_multi_dev[k];
        if (!same) {
            if (g_multi) { tff_multi_destroy(g_multi); g_multi = NULL; }
            if (tff_multi_create(&g_multi, dev, (int32_t)nd) != 0) mexErrMsgIdAndTxt("tftfund:hip", "tff_multi_create: %s", tff_last_error());
            for (k = 0; k < nd; ++k) g_multi_dev[k] = dev[k];
            g_multi_n = (int32_t)nd;
            mexAtExit(cleanup);
        }
        rc = tff_pose_batch_host_multi(g_multi, method_id, mxGetPr(prhs[1]), mxGetPr(prhs[2]), calm_stride, (int64_t)B, (int32_t)N,
                                       Rt2, Rt3, T, Rec, iter, status);
    } else {
        rc = fn(g_ctx, mxGetPr(prhs[1]), mxGetPr(prhs[2]), calm_stride, (int64_t)B, (int32_t)N, Rt2, Rt3, T, Rec, iter, status);
    }
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
