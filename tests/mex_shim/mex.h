/* mex.h -- a small IMPLEMENTED stand-in for MathWorks' MEX / C Matrix API (tests/mex_shim/mex_shim.cpp), so that the
 * gateways in matlab/ can be compiled, linked against libepiekf.so and EXECUTED on a box without MATLAB
 * (tests/mex_shim/driver.cpp calls their mexFunction with MATLAB-shaped column-major arrays).  It implements the ~20 entry
 * points the gateways use, with the documented semantics: mxGetM = first dimension, mxGetN = product of the others,
 * freshly created arrays are zero-filled, mexErrMsg* never return (they throw MexError, which the driver reports).
 * Test infrastructure only; on a MATLAB host the real <mex.h> is used. */
#ifndef EPIEKF_TEST_MEX_SHIM_H
#define EPIEKF_TEST_MEX_SHIM_H
#include <stddef.h>
#include <stdint.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxSTRUCT_CLASS = 2 } mxClassID;
#ifdef __cplusplus
extern "C" {
#endif
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
mxArray *mxCreateString(const char *str);
void mxSetFieldByNumber(mxArray *pa, mwSize index, int fieldnumber, mxArray *value);
/* shim-only accessors for the driver */
int mxShimNumberOfFields(const mxArray *pa);
mxArray *mxShimGetFieldByNumber(const mxArray *pa, int fieldnumber);
const char *mxShimGetFieldName(const mxArray *pa, int fieldnumber);
int mxShimClass(const mxArray *pa);
int mexAtExit(void (*exit_fcn)(void));   /* registered functions run when the MEX file is cleared: the driver runs them at exit */
void mxShimRunAtExit(void);
void mexErrMsgTxt(const char *msg);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#include <string>
struct MexError { std::string id, msg; };
#endif
#endif
