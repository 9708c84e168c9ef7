/* TEST INFRASTRUCTURE ONLY -- a minimal implementation of the MX / MEX calls declared in tests/mex_stub/mex.h (column-major double and
 * char arrays, errors as longjmp back into the driver) plus a driver that calls matlab/tftfund_mex.c's mexFunction the way MATLAB would:
 *     mex_driver <method> <B> <N>      reads B x (6 N) + 27 doubles (Corresp, CalM) from stdin, writes R_t_2 | R_t_3 | T | iter to stdout.
 * tests/test_mex_shim.py builds it with gcc against libtftfund.so and compares the outputs with the C ABI's. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mex.h"

struct mxArray_tag { mxClassID cls; mwSize ndim; mwSize dims[4]; double* pr; char* str; };
static jmp_buf g_jmp;
static char g_err[512];
static void (*g_exit)(void) = NULL;

static mxArray* make(mxClassID cls, mwSize ndim, const mwSize* dims) {
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    size_t n = 1;
    mwSize k;
    a->cls = cls; a->ndim = ndim < 2 ? 2 : ndim;
    for (k = 0; k < 4; ++k) a->dims[k] = 1;
    for (k = 0; k < ndim; ++k) { a->dims[k] = dims[k]; n *= dims[k]; }
    a->pr = (double*)calloc(n ? n : 1, sizeof(double));
    return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { mwSize d[2]; (void)flag; d[0] = m; d[1] = n; return make(mxDOUBLE_CLASS, 2, d); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = v; return a; }
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag) { (void)flag; return make(cls, ndim, dims); }
void mxDestroyArray(mxArray* a) { if (a) { free(a->pr); free(a->str); free(a); } }
void* mxCalloc(mwSize n, mwSize size) { return calloc(n ? n : 1, size); }
void mxFree(void* p) { free(p); }
double* mxGetPr(const mxArray* a) { return a->pr; }
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) { size_t n = 1; mwSize k; for (k = 1; k < a->ndim; ++k) n *= a->dims[k]; return n; }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
size_t mxGetNumberOfElements(const mxArray* a) { size_t n = 1; mwSize k; for (k = 0; k < a->ndim; ++k) n *= a->dims[k]; return n; }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (a->cls != mxCHAR_CLASS || !a->str || strlen(a->str) + 1 > buflen) return 1;
    strcpy(buf, a->str);
    return 0;
}
_Bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
_Bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
_Bool mxIsComplex(const mxArray* a) { (void)a; return 0; }
int mexAtExit(void (*f)(void)) { g_exit = f; return 0; }
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    va_list ap;
    int n = snprintf(g_err, sizeof g_err, "%s: ", id);
    va_start(ap, fmt);
    vsnprintf(g_err + n, sizeof g_err - (size_t)n, fmt, ap);
    va_end(ap);
    longjmp(g_jmp, 1);
}

int main(int argc, char** argv) {
    mxArray *prhs[3], *plhs[5] = {0, 0, 0, 0, 0};
    mwSize dc[3], dk[2];
    long B, N, b;
    size_t got;
    if (argc != 4) { fprintf(stderr, "usage: mex_driver method B N   |   mex_driver bundle_adjustment M N\n"); return 2; }
    B = atol(argv[2]); N = atol(argv[3]);
    if (!strcmp(argv[1], "bundle_adjustment")) {                 /* stdin: Corresp (2M x N), CalM (3M x 3), R_t_0 (3M x 4); stdout: R_t (3M x 4), Reconst (3 x N), iter, repr_err */
        const long M = B;
        mxArray *in[4], *out[4] = {0, 0, 0, 0};
        mwSize d2[2];
        in[0] = (mxArray*)calloc(1, sizeof(mxArray));
        in[0]->cls = mxCHAR_CLASS; in[0]->ndim = 2; in[0]->dims[0] = 1; in[0]->dims[1] = strlen(argv[1]);
        in[0]->str = (char*)malloc(strlen(argv[1]) + 1); strcpy(in[0]->str, argv[1]);
        d2[0] = (mwSize)(2 * M); d2[1] = (mwSize)N; in[1] = mxCreateNumericArray(2, d2, mxDOUBLE_CLASS, mxREAL);
        d2[0] = (mwSize)(3 * M); d2[1] = 3; in[2] = mxCreateNumericArray(2, d2, mxDOUBLE_CLASS, mxREAL);
        d2[0] = (mwSize)(3 * M); d2[1] = 4; in[3] = mxCreateNumericArray(2, d2, mxDOUBLE_CLASS, mxREAL);
        got = fread(in[1]->pr, sizeof(double), (size_t)(2 * M * N), stdin);
        got += fread(in[2]->pr, sizeof(double), (size_t)(9 * M), stdin);
        got += fread(in[3]->pr, sizeof(double), (size_t)(12 * M), stdin);
        if (got != (size_t)(2 * M * N + 21 * M)) { fprintf(stderr, "short input\n"); return 2; }
        if (setjmp(g_jmp)) { fprintf(stderr, "MEXERROR %s\n", g_err); if (g_exit) g_exit(); return 3; }
        mexFunction(4, out, 4, (const mxArray**)in);
        fwrite(mxGetPr(out[0]), sizeof(double), (size_t)(12 * M), stdout);
        fwrite(mxGetPr(out[1]), sizeof(double), (size_t)(3 * N), stdout);
        fwrite(mxGetPr(out[2]), sizeof(double), 1, stdout);
        fwrite(mxGetPr(out[3]), sizeof(double), 1, stdout);
        if (g_exit) g_exit();
        return 0;
    }
    prhs[0] = (mxArray*)calloc(1, sizeof(mxArray));
    prhs[0]->cls = mxCHAR_CLASS; prhs[0]->ndim = 2; prhs[0]->dims[0] = 1; prhs[0]->dims[1] = strlen(argv[1]);
    prhs[0]->str = (char*)malloc(strlen(argv[1]) + 1); strcpy(prhs[0]->str, argv[1]);
    dc[0] = 6; dc[1] = (mwSize)N; dc[2] = (mwSize)B;
    prhs[1] = mxCreateNumericArray(B > 1 ? 3 : 2, dc, mxDOUBLE_CLASS, mxREAL);
    dk[0] = 9; dk[1] = 3;
    prhs[2] = mxCreateNumericArray(2, dk, mxDOUBLE_CLASS, mxREAL);
    got = fread(prhs[1]->pr, sizeof(double), (size_t)(6 * N * B), stdin);
    got += fread(prhs[2]->pr, sizeof(double), 27, stdin);
    if (got != (size_t)(6 * N * B + 27)) { fprintf(stderr, "short input\n"); return 2; }
    if (setjmp(g_jmp)) { fprintf(stderr, "MEXERROR %s\n", g_err); if (g_exit) g_exit(); return 3; }
    mexFunction(5, plhs, 3, (const mxArray**)prhs);
    fwrite(mxGetPr(plhs[0]), sizeof(double), (size_t)(12 * B), stdout);
    fwrite(mxGetPr(plhs[1]), sizeof(double), (size_t)(12 * B), stdout);
    fwrite(mxGetPr(plhs[3]), sizeof(double), (size_t)(27 * B), stdout);
    for (b = 0; b < B; ++b) fwrite(&mxGetPr(plhs[4])[b], sizeof(double), 1, stdout);
    if (mxGetNumberOfElements(plhs[2]) != (size_t)(3 * N * B)) { fprintf(stderr, "Reconst has the wrong size\n"); return 4; }
    if (g_exit) g_exit();
    return 0;
}
