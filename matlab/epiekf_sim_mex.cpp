// epiekf_sim_mex.cpp -- MEX gateway for the forward simulators and the NPI cost of the path
// (Tools/SIalpha_Controlled.m, SEIRP.m, SEIRPSaturatedResource.m, NPICost.m), one chain per call:
//   [s, i, alpha]     = epiekf_sim_mex('sialpha', u, sp, z)            u n_npi x K, sp 48 x 1 (EPI_SIM_* rows), z 3 x K or []
//   out               = epiekf_sim_mex('seirp', par, init, dt, sat)    par 7 x K, init 5 x 1, sat 6 x 1 or []; out 5 x K
//   [J0, J1]          = epiekf_sim_mex('npicost', newcases, inputs, weights)   inputs n x T, weights n x T or n x 1
//   [s, i]            = epiekf_sim_mex('si', alpha, [beta; s0; i0], K, dt)      alpha with at least K-1 elements
// With B = 1 MATLAB's column-major arrays ARE the ABI's [K][rows][1] arrays and are passed straight through.
// Build on a MATLAB host:  mex -I../include epiekf_sim_mex.cpp -L../epidemicmodeling_amd -lepiekf
#include <string.h>
#include "mex.h"
#include "epiekf.h"

static void fail_if(int rc, const char *err)
{
    if (rc != EPI_OK) mexErrMsgIdAndTxt("epiekf:error", "%s (%s)", err, epi_status_string(rc));
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[16] = {0}, err[256] = {0};
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd) != 0) mexErrMsgTxt("epiekf_sim_mex: first argument is the command string");
    if (strcmp(cmd, "sialpha") == 0) {
        if (nrhs != 4) mexErrMsgTxt("epiekf_sim_mex('sialpha', u, sp, z): 4 inputs expected");
        const mwSize n = mxGetM(prhs[1]), K = mxGetN(prhs[1]);
        if (mxGetNumberOfElements(prhs[2]) != EPI_SIM_PRM_COUNT) mexErrMsgTxt("sp must have 48 elements");
        const bool noise = !mxIsEmpty(prhs[3]);
        if (noise && (mxGetM(prhs[3]) != 3 || mxGetN(prhs[3]) != K)) mexErrMsgTxt("z must be 3 x K");
        epi_sim_desc d;
        memset(&d, 0, sizeof d);
        d.abi_version = EPIEKF_ABI_VERSION; d.B = 1; d.K = (int32_t)K; d.Su = 1; d.n_npi = (int32_t)n; d.noise = noise;
        mxArray *o[3] = {mxCreateDoubleMatrix(1, K, mxREAL), mxCreateDoubleMatrix(1, K, mxREAL), mxCreateDoubleMatrix(1, K, mxREAL)};
        fail_if(epi_sialpha_sim_host(&d, NULL, mxGetPr(prhs[1]), mxGetPr(prhs[2]), noise ? mxGetPr(prhs[3]) : NULL,
                                     mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]), NULL, NULL, /*device=*/0, err), err);
        for (int k = 0; k < 3; k++)
            if (k < nlhs || k == 0) plhs[k] = o[k]; else mxDestroyArray(o[k]);
    } else if (strcmp(cmd, "seirp") == 0) {
        if (nrhs != 5) mexErrMsgTxt("epiekf_sim_mex('seirp', par, init, dt, sat): 5 inputs expected");
        const mwSize K = mxGetN(prhs[1]);
        if (mxGetM(prhs[1]) != 7 || mxGetNumberOfElements(prhs[2]) != 5) mexErrMsgTxt("par must be 7 x K and init 5 x 1");
        const bool sat = !mxIsEmpty(prhs[4]);
        if (sat && mxGetNumberOfElements(prhs[4]) != 6) mexErrMsgTxt("sat must have 6 elements");
        plhs[0] = mxCreateDoubleMatrix(5, K, mxREAL);
        fail_if(epi_seirp_sim_host(1, (int32_t)K, (int32_t)K, mxGetScalar(prhs[3]), sat, /*integrator=*/0, mxGetPr(prhs[1]),
                                   mxGetPr(prhs[2]), sat ? mxGetPr(prhs[4]) : NULL, mxGetPr(plhs[0]), /*device=*/0, err), err);
    } else if (strcmp(cmd, "npicost") == 0) {
        if (nrhs != 4) mexErrMsgTxt("epiekf_sim_mex('npicost', newcases, inputs, weights): 4 inputs expected");
        const mwSize n = mxGetM(prhs[2]), T = mxGetN(prhs[2]);
        if (mxGetNumberOfElements(prhs[1]) != T) mexErrMsgTxt("newcases must have T elements");
        const mwSize wn = mxGetN(prhs[3]);
        if (mxGetM(prhs[3]) != n || (wn != T && wn != 1)) mexErrMsgTxt("Arrays have incompatible sizes for this operation.");
        plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
        mxArray *j1 = mxCreateDoubleMatrix(1, 1, mxREAL);
        fail_if(epi_npi_cost_host(1, (int32_t)T, (int32_t)n, 1, (wn == T && T > 1) ? 1 : 0, NULL, mxGetPr(prhs[1]), mxGetPr(prhs[2]),
                                  mxGetPr(prhs[3]), mxGetPr(plhs[0]), mxGetPr(j1), /*device=*/0, err), err);
        if (nlhs > 1) plhs[1] = j1; else mxDestroyArray(j1);
    } else if (strcmp(cmd, "si") == 0) {
        if (nrhs != 5) mexErrMsgTxt("epiekf_sim_mex('si', alpha, prm, K, dt): 5 inputs expected");
        const mwSize K = (mwSize)mxGetScalar(prhs[3]);
        if (K < 1 || (K > 1 && mxGetNumberOfElements(prhs[1]) < K - 1)) mexErrMsgTxt("Index exceeds the number of array elements.");
        if (mxGetNumberOfElements(prhs[2]) != 3) mexErrMsgTxt("prm must be [beta; s0; i0]");
        plhs[0] = mxCreateDoubleMatrix(1, K, mxREAL);
        mxArray *iv = mxCreateDoubleMatrix(1, K, mxREAL);
        fail_if(epi_si_controlled_host(1, (int32_t)K, 1, mxGetScalar(prhs[4]), NULL, mxGetPr(prhs[1]), mxGetPr(prhs[2]),
                                       mxGetPr(plhs[0]), mxGetPr(iv), /*device=*/0, err), err);
        if (nlhs > 1) plhs[1] = iv; else mxDestroyArray(iv);
    } else {
        mexErrMsgTxt("epiekf_sim_mex: unknown command");
    }
}
