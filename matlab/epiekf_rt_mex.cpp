// epiekf_rt_mex.cpp -- MEX gateway for Tools/Rt_ExpFitEKF.m:  out = epiekf_rt_mex(x, rp, inv_monitor_len, order)
// x is 1 x T, rp the EPI_RT_PRM_COUNT x 1 column built by matlab/Tools/Rt_ExpFitEKF.m.  B = 1, so MATLAB's
// column-major arrays are the ABI's [T][rows][1] arrays and are passed straight through.
// Build on a MATLAB host:  mex -I../include epiekf_rt_mex.cpp -L../epidemicmodeling_amd -lepiekf
#include <string.h>
#include "mex.h"
#include "epiekf.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 4) mexErrMsgTxt("epiekf_rt_mex: 4 inputs expected");
    const mwSize T = mxGetN(prhs[0]);
    if (mxGetM(prhs[0]) != 1) mexErrMsgTxt("scalar-observation filter: size(x,1) must be 1");
    if (mxGetNumberOfElements(prhs[1]) != EPI_RT_PRM_COUNT) mexErrMsgTxt("epiekf_rt_mex: rp must have 19 elements");
    epi_rt_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.B = 1; d.T = (int32_t)T; d.Sx = 1;
    d.L = (int32_t)mxGetScalar(prhs[2]); d.order = (int32_t)mxGetScalar(prhs[3]);
    const mwSize dP[3] = {2, 2, T}, dK[3] = {2, 1, T};
    mxArray *v[9] = {mxCreateDoubleMatrix(2, T, mxREAL), mxCreateDoubleMatrix(2, T, mxREAL),
                     mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL), mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL),
                     mxCreateNumericArray(3, dK, mxDOUBLE_CLASS, mxREAL), mxCreateDoubleMatrix(2, T, mxREAL),
                     mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL), mxCreateDoubleMatrix(1, T, mxREAL),
                     mxCreateDoubleMatrix(T, 1, mxREAL)};   // squeeze(rho): T x 1
    epi_rt_outputs out = {mxGetPr(v[0]), mxGetPr(v[1]), mxGetPr(v[2]), mxGetPr(v[3]), mxGetPr(v[4]),
                          mxGetPr(v[5]), mxGetPr(v[6]), mxGetPr(v[7]), mxGetPr(v[8])};
    char err[256] = {0};
    const int rc = epi_rt_expfit_run_host(&d, NULL, mxGetPr(prhs[0]), mxGetPr(prhs[1]), &out, /*device=*/0, err);
    if (rc == EPI_ERR_UNDEFINED_ORDER) mexErrMsgTxt(err);       // the reference's own error text
    if (rc != EPI_OK) mexErrMsgIdAndTxt("epiekf:error", "%s (%s)", err, epi_status_string(rc));
    const char *names[] = {"S_MINUS", "S_PLUS", "P_MINUS", "P_PLUS", "K_GAIN", "S_SMOOTH", "P_SMOOTH", "innovations", "rho"};
    plhs[0] = mxCreateStructMatrix(1, 1, 9, names);
    for (int i = 0; i < 9; i++) mxSetFieldByNumber(plhs[0], 0, i, v[i]);
    (void)nlhs;
}
