/* mex.h -- COMPILE-CHECK STUB, not MathWorks' header.  Declares only the handful of MEX API entry points that
 * matlab/epiekf_mex.cpp and matlab/epiekf_rt_mex.cpp use, with the signatures documented in the MATLAB C Matrix API
 * reference, so that tests/test_abi_and_host.py can run the gateways through `g++ -fsyntax-only` in an image that has no
 * MATLAB.  Nothing here is linked or executed; on a MATLAB host the real <mex.h> is used. */
#ifndef EPIEKF_TEST_MEX_STUB_H
#define EPIEKF_TEST_MEX_STUB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6 } mxClassID;
double *mxGetPr(const mxArray *pa);
int mxGetString(const mxArray *pa, char *str, mwSize strlen);
bool mxIsEmpty(const mxArray *pa);
void mxDestroyArray(mxArray *pa);
double mxGetScalar(const mxArray *pa);
size_t mxGetM(const mxArray *pa);
size_t mxGetN(const mxArray *pa);
size_t mxGetNumberOfElements(const mxArray *pa);
mwSize mxGetNumberOfDimensions(const mxArray *pa);
const mwSize *mxGetDimensions(const mxArray *pa);
void *mxGetData(const mxArray *pa);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
void mxSetFieldByNumber(mxArray *pa, mwSize index, int fieldnumber, mxArray *value);
void mexErrMsgTxt(const char *msg);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif
