/* TEST INFRASTRUCTURE ONLY -- a stand-in for MATLAB's <mex.h> so that matlab/tftfund_mex.c can go through a C compiler in the build
 * container (no MATLAB there): the types and the prototypes of the MEX / MX API calls the gateway uses, with the signatures MATLAB
 * documents (C Matrix API, "mxCreateNumericArray", "mexErrMsgIdAndTxt", ...).  Declarations only; tests/mex_stub/mex_stub.c holds a
 * minimal implementation for the GPU test that drives mexFunction end to end.  The real build links against MATLAB's own header. */
#ifndef TFF_TEST_MEX_H
#define TFF_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef size_t mwSize;
typedef ptrdiff_t mwSignedIndex;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum { mxUNKNOWN_CLASS = 0, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef int mxLogicalInt;

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
void mxDestroyArray(mxArray* pa);
void* mxCalloc(mwSize n, mwSize size);
void mxFree(void* ptr);
double* mxGetPr(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
mwSize mxGetNumberOfDimensions(const mxArray* pa);
const mwSize* mxGetDimensions(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
_Bool mxIsChar(const mxArray* pa);
_Bool mxIsDouble(const mxArray* pa);
_Bool mxIsComplex(const mxArray* pa);
int mexAtExit(void (*exit_fcn)(void));
void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...)
#if defined(__GNUC__)
    __attribute__((noreturn, format(printf, 2, 3)))
#endif
    ;
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#ifdef __cplusplus
}
#endif
#endif
