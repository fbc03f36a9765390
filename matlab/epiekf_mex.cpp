// epiekf_mex.cpp -- MEX gateway: MATLAB  <->  libepiekf.so (include/epiekf.h).
//
//   out = epiekf_mex(model_id, u, x, prm, s_init, Ps_init, s_final, Ps_final, Q, R_v, L, order, obs_type)
//
// One call == one call of the reference L2 wrapper (Tools/SIAlphaModelEKF.m:1 ...).  With B = 1 the C ABI's
// arrays have exactly MATLAB's column-major layout, so mxGetPr() pointers are passed straight through and the
// outputs are written in place -- no transposition.  `prm` is the EPI_PRM_COUNT x 1 column the .m wrapper
// builds from the `params` struct (w already resolved through the implicit-expansion rule).
// Build on a MATLAB host (mex.h is not available in the build image of this repo):
//   mex -I../include epiekf_mex.cpp -L../epidemicmodeling_amd -lepiekf
#include <string.h>
#include "mex.h"
#include "epiekf.h"

static void fail(int rc, const char *err)
{
    // the four reference errors keep their MATLAB text; everything else is prefixed
    if (rc >= EPI_ERR_OBS_TYPE && rc <= EPI_ERR_UNDEFINED_ORDER) mexErrMsgTxt(err);
    mexErrMsgIdAndTxt("epiekf:error", "%s (%s)", err, epi_status_string(rc));
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    mexAtExit(epi_host_pool_release);       // `clear mex` hands the library's pooled host contexts and helper streams back
    if (nrhs != 13) mexErrMsgTxt("epiekf_mex: 13 inputs expected");
    const int model = (int)mxGetScalar(prhs[0]);
    const int m = epi_model_dim(model);
    if (m < 0) mexErrMsgTxt("epiekf_mex: unknown model id");
    const mxArray *u = prhs[1], *x = prhs[2], *Rv = prhs[9];
    const mwSize T = mxGetN(x), n_npi = mxGetM(u);
    if (mxGetM(x) != 1) mexErrMsgTxt("scalar-observation filter: size(x,1) must be 1");

    epi_batch_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.model = model; d.B = 1; d.T = (int32_t)T; d.Sx = 1; d.Su = 1;
    d.n_npi = (int32_t)n_npi; d.L = (int32_t)mxGetScalar(prhs[10]); d.order = (int32_t)mxGetScalar(prhs[11]);
    d.obs_type = (int32_t)mxGetScalar(prhs[12]);
    const mwSize nR = mxGetNumberOfElements(Rv);
    if (nR == 1) d.r_mode = 0;                                   // scalar R_v: fixed_R = true  (GenericEKF.m:79-81)
    else if (nR == T && (mxGetM(Rv) == 1 || mxGetN(Rv) == 1)) d.r_mode = 1;   // 1 x T vector (:82-85)
    else mexErrMsgTxt("Observation noise covariance noise mismatch");
    // Q_w: m x m, or m x m x T (time-varying; column-major pages == the ABI's [T][m*m][1])
    const mwSize nQ = mxGetNumberOfElements(prhs[8]);
    if (nQ == (mwSize)(m * m)) d.q_mode = 0;
    else if (nQ == (mwSize)(m * m) * T) d.q_mode = 1;
    else mexErrMsgTxt("Process noise covariance noise mismatch");
    const bool has_uos = (model <= EPI_MODEL_SIA6_BWD);
    d.out_mask = EPI_OUT_ALL & ~(has_uos ? 0u : (unsigned)EPI_OUT_U_OPT_SMOOTH);

    epi_inputs in;
    memset(&in, 0, sizeof in);
    in.x = mxGetPr(x); in.u = mxGetPr(u); in.prm = mxGetPr(prhs[3]);
    in.s_init = mxGetPr(prhs[4]); in.Ps_init = mxGetPr(prhs[5]);
    in.s_final = mxGetPr(prhs[6]); in.Ps_final = mxGetPr(prhs[7]); in.Q = mxGetPr(prhs[8]);
    if (d.r_mode == 0) in.R_scalar = mxGetPr(Rv); else in.R_series = mxGetPr(Rv);

    // outputs, allocated with MATLAB's shapes (column-major == [T][rows][1])
    const mwSize dP[3] = {(mwSize)m, (mwSize)m, T}, dK[3] = {(mwSize)m, 1, T};
    mxArray *o_u = mxCreateDoubleMatrix(n_npi, T, mxREAL), *o_us = mxCreateDoubleMatrix(n_npi, T, mxREAL);
    mxArray *o_sm = mxCreateDoubleMatrix(m, T, mxREAL), *o_sp = mxCreateDoubleMatrix(m, T, mxREAL);
    mxArray *o_ss = mxCreateDoubleMatrix(m, T, mxREAL);
    mxArray *o_pm = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_pp = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_ps = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_k = mxCreateNumericArray(3, dK, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_in = mxCreateDoubleMatrix(1, T, mxREAL), *o_rho = mxCreateDoubleMatrix(T, 1, mxREAL);
    epi_outputs out;
    memset(&out, 0, sizeof out);
    out.u_opt = mxGetPr(o_u); out.u_opt_smooth = has_uos ? mxGetPr(o_us) : NULL;
    out.S_MINUS = mxGetPr(o_sm); out.S_PLUS = mxGetPr(o_sp); out.S_SMOOTH = mxGetPr(o_ss);
    out.P_MINUS = mxGetPr(o_pm); out.P_PLUS = mxGetPr(o_pp); out.P_SMOOTH = mxGetPr(o_ps);
    out.K_GAIN = mxGetPr(o_k); out.innovations = mxGetPr(o_in); out.rho = mxGetPr(o_rho);

    char err[256] = {0};
    const int rc = epi_ekf_run_host(&d, &in, &out, /*device=*/0, err);
    if (rc != EPI_OK) fail(rc, err);

    // one struct out; the .m wrappers unpack it in the reference's output order
    const char *names[] = {"u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS",
                           "P_SMOOTH", "K_GAIN", "innovations", "rho"};
    mxArray *vals[] = {o_u, o_us, o_sm, o_sp, o_ss, o_pm, o_pp, o_ps, o_k, o_in, o_rho};
    plhs[0] = mxCreateStructMatrix(1, 1, 11, names);
    for (int i = 0; i < 11; i++) mxSetFieldByNumber(plhs[0], 0, i, vals[i]);
    (void)nlhs;
}
