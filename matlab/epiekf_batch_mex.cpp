// epiekf_batch_mex.cpp -- batched MEX gateway: MANY chains of one model in ONE call (include/epiekf.h, B > 1).
//
//   out = epiekf_batch_mex(model_id, u, x, prm, s_init, Ps_init, s_final, Ps_final, Q, R_v, L, order, obs_type,
//                          x_series, u_series)
//
// MATLAB's column-major arrays map onto the ABI's time-major / chain-minor layout without any transposition when the
// CHAIN index is the FIRST dimension:
//   u        Su x n_npi x T     (ABI [T][n_npi][Su])        x        Sx x T         (ABI [T][Sx])
//   prm      B x 61             (ABI [61][B])               s_init   B x m,  Ps_init  B x m*m (columns = vec(Ps_init))
//   s_final  B x m              Ps_final B x m*m            Q        B x m*m
//   R_v      B x 1 (fixed R, adapted when beta ~= 1) or Sx x T (per-day variances of each observation series)
//   x_series, u_series          B x 1 int32, zero-based: which observation / control series a chain reads
//                               ([] = identity, then Sx == B / Su == B)
// and the outputs come back the same way:  S_* B x m x T,  P_* B x m*m x T,  K_GAIN B x m x T,  u_opt* B x n_npi x T,
// innovations, rho B x T.  squeeze(out.S_SMOOTH(c, :, :)) is what the reference call of chain c returns.
// This is how the serial loops of the reference (for ll = 1 : num_pareto_front_points, Tools/TrainPredictPrescribeNPI.m:
// 421-460; for k = 1 : NumGeoLocations, :93) become one launch -- see matlab/Tools/SIAlphaModelEKFOptControlledSweep.m.
// Build on a MATLAB host:  mex -I../include epiekf_batch_mex.cpp -L../epidemicmodeling_amd -lepiekf
#include <string.h>
#include "mex.h"
#include "epiekf.h"

static const int32_t *series_or_null(const mxArray *a, mwSize B, const char *name)
{
    if (mxGetNumberOfElements(a) == 0) return NULL;
    if (mxGetNumberOfElements(a) != B) mexErrMsgIdAndTxt("epiekf:arg", "%s must have one entry per chain", name);
    return (const int32_t *)mxGetData(a);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    mexAtExit(epi_host_pool_release);       // `clear mex` hands the library's pooled host contexts and helper streams back
    if (nrhs != 15) mexErrMsgTxt("epiekf_batch_mex: 15 inputs expected");
    const int model = (int)mxGetScalar(prhs[0]);
    const int m = epi_model_dim(model);
    if (m < 0) mexErrMsgTxt("epiekf_batch_mex: unknown model id");
    const mxArray *u = prhs[1], *x = prhs[2], *prm = prhs[3], *Rv = prhs[9];
    const mwSize *du = mxGetDimensions(u);
    if (mxGetNumberOfDimensions(u) != 3) mexErrMsgTxt("u must be Su x n_npi x T");
    const mwSize Su = du[0], n_npi = du[1], T = du[2];
    const mwSize Sx = mxGetM(x), B = mxGetM(prm);
    if (mxGetN(x) != T) mexErrMsgTxt("x must be Sx x T with the T of u");
    if (mxGetN(prm) != EPI_PRM_COUNT) mexErrMsgTxt("prm must be B x 61 (epiekf_pack_params, one row per chain)");

    epi_batch_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.model = model; d.B = (int32_t)B; d.T = (int32_t)T;
    d.Sx = (int32_t)Sx; d.Su = (int32_t)Su; d.n_npi = (int32_t)n_npi;
    d.L = (int32_t)mxGetScalar(prhs[10]); d.order = (int32_t)mxGetScalar(prhs[11]); d.obs_type = (int32_t)mxGetScalar(prhs[12]);
    const mwSize nR = mxGetNumberOfElements(Rv);
    if (nR == B && mxGetN(Rv) == 1) d.r_mode = 0;                       // one fixed R_v per chain
    else if (mxGetM(Rv) == Sx && mxGetN(Rv) == T) d.r_mode = 1;         // per-day variances per observation series
    else mexErrMsgTxt("Observation noise covariance noise mismatch");
    if (mxGetNumberOfElements(prhs[8]) != B * (mwSize)(m * m)) mexErrMsgTxt("Process noise covariance noise mismatch");
    const bool has_uos = (model <= EPI_MODEL_SIA6_BWD);
    d.out_mask = EPI_OUT_ALL & ~(has_uos ? 0u : (unsigned)EPI_OUT_U_OPT_SMOOTH);

    epi_inputs in;
    memset(&in, 0, sizeof in);
    in.x_series = series_or_null(prhs[13], B, "x_series");
    in.u_series = series_or_null(prhs[14], B, "u_series");
    in.x = mxGetPr(x); in.u = mxGetPr(u); in.prm = mxGetPr(prm);
    in.s_init = mxGetPr(prhs[4]); in.Ps_init = mxGetPr(prhs[5]);
    in.s_final = mxGetPr(prhs[6]); in.Ps_final = mxGetPr(prhs[7]); in.Q = mxGetPr(prhs[8]);
    if (d.r_mode == 0) in.R_scalar = mxGetPr(Rv); else in.R_series = mxGetPr(Rv);

    const mwSize dS[3] = {B, (mwSize)m, T}, dP[3] = {B, (mwSize)(m * m), T}, dU[3] = {B, n_npi, T};
    mxArray *o_u = mxCreateNumericArray(3, dU, mxDOUBLE_CLASS, mxREAL), *o_us = mxCreateNumericArray(3, dU, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_sm = mxCreateNumericArray(3, dS, mxDOUBLE_CLASS, mxREAL), *o_sp = mxCreateNumericArray(3, dS, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_ss = mxCreateNumericArray(3, dS, mxDOUBLE_CLASS, mxREAL), *o_k = mxCreateNumericArray(3, dS, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_pm = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL), *o_pp = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_ps = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
    mxArray *o_in = mxCreateDoubleMatrix(B, T, mxREAL), *o_rho = mxCreateDoubleMatrix(B, T, mxREAL);
    epi_outputs out;
    memset(&out, 0, sizeof out);
    out.u_opt = mxGetPr(o_u); out.u_opt_smooth = has_uos ? mxGetPr(o_us) : NULL;
    out.S_MINUS = mxGetPr(o_sm); out.S_PLUS = mxGetPr(o_sp); out.S_SMOOTH = mxGetPr(o_ss);
    out.P_MINUS = mxGetPr(o_pm); out.P_PLUS = mxGetPr(o_pp); out.P_SMOOTH = mxGetPr(o_ps);
    out.K_GAIN = mxGetPr(o_k); out.innovations = mxGetPr(o_in); out.rho = mxGetPr(o_rho);

    char err[256] = {0};
    const int rc = epi_ekf_run_host(&d, &in, &out, /*device=*/0, err);
    if (rc >= EPI_ERR_OBS_TYPE && rc <= EPI_ERR_UNDEFINED_ORDER) mexErrMsgTxt(err);    // the reference's own error text
    if (rc != EPI_OK) mexErrMsgIdAndTxt("epiekf:error", "%s (%s)", err, epi_status_string(rc));

    const char *names[] = {"u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS",
                           "P_SMOOTH", "K_GAIN", "innovations", "rho"};
    mxArray *vals[] = {o_u, o_us, o_sm, o_sp, o_ss, o_pm, o_pp, o_ps, o_k, o_in, o_rho};
    plhs[0] = mxCreateStructMatrix(1, 1, 11, names);
    for (int i = 0; i < 11; i++) mxSetFieldByNumber(plhs[0], 0, i, vals[i]);
    (void)nlhs;
}
