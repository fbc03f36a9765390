// epiekf_pipeline_mex.cpp -- MEX gateway for the stages of Tools/TrainPredictPrescribeNPI.m around the filter, each for ALL
// regions in one call (include/epiekf.h: the *_host entry points).  The REGION index is the FIRST dimension of every array,
// so MATLAB's column-major arrays are the ABI's region-minor arrays without any transposition.
//
//   out = epiekf_pipeline_mex('prescribe', x, u, R_v, prm, s_init, Ps_init, s_final, Ps_final, Q_w, epsilons, sp, J0_prefix,
//                             J1_prefix, t_hist, L, order, obs_type, devices)
//       the cost-weight sweep of every region -- SIAlphaModelEKFOptControlled for each epsilon, the scenario scoring and the
//       Pareto front (Tools/TrainPredictPrescribeNPI.m:421-493, 624-633) -- on the GPUs listed in `devices` (zero-based ids,
//       [] = device 0).  x, R_v  R x T (NaN over the horizon);  u  R x n_npi x T (NaN = choose optimally);  prm  R x 61
//       (epiekf_pack_params per region; its epsilon entry is ignored);  s_init, s_final  R x 6;  Ps_init, Ps_final, Q_w
//       R x 36 (vec'd matrices);  epsilons  P x 1 = human_npi_cost_factor;  sp  R x 48 (EPI_SIM_* columns: s/i/alpha_historic
//       (end), model constants, NPI_MAXES, npi_weights);  J0_prefix, J1_prefix  R x 1 (sequential historic sums of NPICost).
//       out.J0, out.J1  P x R (column r = J0_opt_control of region r);  out.on_front  P x R (logical as double);
//       out.I_opt  R x 1 (ONE-based);  out.u_opt  R x n_npi x T = opt_control_input_smooth of the optimum;  out.S_opt
//       R x 6 x T = its S_SMOOTH.
//   out = epiekf_pipeline_mex('preprocess', cases, deaths, population, ip, W, first_num_days, min_cases)
//       :142-198, 201-202, 240.  cases, deaths ([] = none)  S x T cumulative counts;  population  S x 1;  ip ([] = none)
//       S x n_npi x T.  Fields new_refined, new_smoothed, zero_lag, x_new, x_total, R_v, fatality (S x T), I0 (S x 1),
//       ip_filled (S x n_npi x T).
//   [a, b, min_err, iters] = epiekf_pipeline_mex('nnls', X, y, max_iters)
//       :251-276 ('NONNEGATIVELS').  X  S x n x D = NPI_MAXES - InterventionPlans over the window;  y  S x D;  a  S x n;
//       b, min_err, iters  S x 1.
//   [J0, J1, u] = epiekf_pipeline_mex('mc', sp, u_min, n_scen, K, seed, z, J0_prefix, J1_prefix, prefix_days)
//       :496-521.  sp  R x 48;  u_min  R x n_npi;  z ([] = noise-free)  (n_scen*R) x 3 x K;  J0, J1  R x n_scen;
//       u  (n_scen*R) x n_npi x K (only when requested).
// Build on a MATLAB host:  mex -I../include epiekf_pipeline_mex.cpp -L../epidemicmodeling_amd -lepiekf
#include <string.h>
#include <vector>
#include "mex.h"
#include "epiekf.h"

static void fail_if(int rc, const char *err)
{
    if (rc != EPI_OK) mexErrMsgIdAndTxt("epiekf:error", "%s (%s)", err, epi_status_string(rc));
}
static const double *opt(const mxArray *a) { return mxIsEmpty(a) ? NULL : mxGetPr(a); }
static void want(const mxArray *a, mwSize rows, mwSize cols, const char *what)
{
    if (mxGetM(a) != rows || mxGetN(a) != cols) mexErrMsgIdAndTxt("epiekf:arg", "%s must be %d x %d", what, (int)rows, (int)cols);
}
static mxArray *dbl3(mwSize a, mwSize b, mwSize c)
{
    const mwSize d[3] = {a, b, c};
    return mxCreateNumericArray(3, d, mxDOUBLE_CLASS, mxREAL);
}

static void prescribe(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    // (an optional 20th input: epi_prescribe_desc.placement_tries -- the first sweep of a size keeps the fastest of that many device arenas)
    if (nrhs != 19 && nrhs != 20) mexErrMsgTxt("epiekf_pipeline_mex('prescribe', ...): 19 inputs expected (+ optional placement_tries)");
    const mxArray *x = prhs[1], *u = prhs[2];
    if (mxGetNumberOfDimensions(u) != 3) mexErrMsgTxt("u must be R x n_npi x T");
    const mwSize *du = mxGetDimensions(u);
    const mwSize R = du[0], n = du[1], T = du[2], P = mxGetNumberOfElements(prhs[10]);
    want(x, R, T, "x"); want(prhs[3], R, T, "R_v"); want(prhs[4], R, EPI_PRM_COUNT, "prm");
    want(prhs[5], R, 6, "s_init"); want(prhs[6], R, 36, "Ps_init"); want(prhs[7], R, 6, "s_final"); want(prhs[8], R, 36, "Ps_final");
    want(prhs[9], R, 36, "Q_w"); want(prhs[11], R, EPI_SIM_PRM_COUNT, "sp");
    if (mxGetNumberOfElements(prhs[12]) != R || mxGetNumberOfElements(prhs[13]) != R) mexErrMsgTxt("J0_prefix, J1_prefix must have one entry per region");
    if (P < 1) mexErrMsgTxt("epsilons is empty");
    epi_prescribe_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.R = (int32_t)R; d.P = (int32_t)P; d.T = (int32_t)T; d.t_hist = (int32_t)mxGetScalar(prhs[14]);
    d.n_npi = (int32_t)n; d.L = (int32_t)mxGetScalar(prhs[15]); d.order = (int32_t)mxGetScalar(prhs[16]); d.obs_type = (int32_t)mxGetScalar(prhs[17]);
    if (nrhs == 20) d.placement_tries = (int32_t)mxGetScalar(prhs[19]);
    epi_prescribe_inputs in;
    memset(&in, 0, sizeof in);
    in.x = mxGetPr(x); in.u = mxGetPr(u); in.R_series = mxGetPr(prhs[3]); in.prm = mxGetPr(prhs[4]);
    in.s_init = mxGetPr(prhs[5]); in.Ps_init = mxGetPr(prhs[6]); in.s_final = mxGetPr(prhs[7]); in.Ps_final = mxGetPr(prhs[8]);
    in.Q = mxGetPr(prhs[9]); in.eps = mxGetPr(prhs[10]); in.sp = mxGetPr(prhs[11]); in.J0_prefix = mxGetPr(prhs[12]); in.J1_prefix = mxGetPr(prhs[13]);
    std::vector<int> devs;
    for (mwSize k = 0; k < mxGetNumberOfElements(prhs[18]); k++) devs.push_back((int)mxGetPr(prhs[18])[k]);
    if (devs.empty()) devs.push_back(0);
    const char *names[] = {"J0", "J1", "on_front", "I_opt", "u_opt", "S_opt"};
    mxArray *f[6] = {mxCreateDoubleMatrix(P, R, mxREAL), mxCreateDoubleMatrix(P, R, mxREAL), mxCreateDoubleMatrix(P, R, mxREAL),
                     mxCreateDoubleMatrix(R, 1, mxREAL), dbl3(R, n, T), dbl3(R, 6, T)};
    std::vector<int32_t> on((size_t)(R * P)), iopt((size_t)R);
    epi_prescribe_outputs out;
    memset(&out, 0, sizeof out);
    out.J0 = mxGetPr(f[0]); out.J1 = mxGetPr(f[1]); out.on_front = on.data(); out.i_opt = iopt.data();
    out.u_opt = mxGetPr(f[4]); out.S_opt = mxGetPr(f[5]);
    char err[256] = {0};
    const int rc = epi_sweep_prescribe_host(&d, &in, &out, (int)devs.size(), devs.data(), err);
    if (rc != EPI_OK) { for (mxArray *a : f) mxDestroyArray(a); fail_if(rc, err); }
    for (size_t k = 0; k < on.size(); k++) mxGetPr(f[2])[k] = (double)on[k];
    for (size_t k = 0; k < iopt.size(); k++) mxGetPr(f[3])[k] = (double)iopt[k] + 1.0;      // MATLAB indices start at one
    plhs[0] = mxCreateStructMatrix(1, 1, 6, names);
    for (int k = 0; k < 6; k++) mxSetFieldByNumber(plhs[0], 0, k, f[k]);
}

static void preprocess(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (nrhs != 8) mexErrMsgTxt("epiekf_pipeline_mex('preprocess', cases, deaths, population, ip, W, first_num_days, min_cases): 8 inputs expected");
    const mwSize S = mxGetM(prhs[1]), T = mxGetN(prhs[1]);
    const bool has_d = !mxIsEmpty(prhs[2]), has_ip = !mxIsEmpty(prhs[4]);
    if (has_d) want(prhs[2], S, T, "deaths");
    if (mxGetNumberOfElements(prhs[3]) != S) mexErrMsgTxt("population must have one entry per region");
    mwSize n = 0;
    if (has_ip) {
        if (mxGetNumberOfDimensions(prhs[4]) != 3 || mxGetDimensions(prhs[4])[0] != S || mxGetDimensions(prhs[4])[2] != T) mexErrMsgTxt("ip must be S x n_npi x T");
        n = mxGetDimensions(prhs[4])[1];
    }
    epi_pre_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.S = (int32_t)S; d.T = (int32_t)T; d.n_npi = (int32_t)n; d.W = (int32_t)mxGetScalar(prhs[5]);
    d.first_num_days = (int32_t)mxGetScalar(prhs[6]); d.min_cases = mxGetScalar(prhs[7]);
    const char *names[] = {"new_refined", "new_smoothed", "zero_lag", "x_new", "x_total", "R_v", "fatality", "I0", "ip_filled"};
    mxArray *f[9];
    for (int k = 0; k < 7; k++) f[k] = (k == 6 && !has_d) ? mxCreateDoubleMatrix(0, 0, mxREAL) : mxCreateDoubleMatrix(S, T, mxREAL);
    f[7] = mxCreateDoubleMatrix(S, 1, mxREAL);
    f[8] = has_ip ? dbl3(S, n, T) : mxCreateDoubleMatrix(0, 0, mxREAL);
    epi_pre_outputs out;
    memset(&out, 0, sizeof out);
    out.new_refined = mxGetPr(f[0]); out.new_smoothed = mxGetPr(f[1]); out.zero_lag = mxGetPr(f[2]); out.x_new = mxGetPr(f[3]);
    out.x_total = mxGetPr(f[4]); out.R_v = mxGetPr(f[5]); out.fatality = has_d ? mxGetPr(f[6]) : NULL; out.I0 = mxGetPr(f[7]);
    out.ip_filled = has_ip ? mxGetPr(f[8]) : NULL;
    char err[256] = {0};
    const int rc = epi_preprocess_host(&d, mxGetPr(prhs[1]), opt(prhs[2]), mxGetPr(prhs[3]), opt(prhs[4]), &out, /*device=*/0, err);
    if (rc != EPI_OK) { for (mxArray *a : f) mxDestroyArray(a); fail_if(rc, err); }
    plhs[0] = mxCreateStructMatrix(1, 1, 9, names);
    for (int k = 0; k < 9; k++) mxSetFieldByNumber(plhs[0], 0, k, f[k]);
}

static void nnls(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 4) mexErrMsgTxt("epiekf_pipeline_mex('nnls', X, y, max_iters): 4 inputs expected");
    if (mxGetNumberOfDimensions(prhs[1]) != 3) mexErrMsgTxt("X must be S x n x D");
    const mwSize *dx = mxGetDimensions(prhs[1]);
    const mwSize S = dx[0], n = dx[1], D = dx[2];
    want(prhs[2], S, D, "y");
    epi_nnls_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.S = (int32_t)S; d.D = (int32_t)D; d.n = (int32_t)n; d.max_iters = (int32_t)mxGetScalar(prhs[3]);
    mxArray *a = mxCreateDoubleMatrix(S, n, mxREAL), *b = mxCreateDoubleMatrix(S, 1, mxREAL), *me = mxCreateDoubleMatrix(S, 1, mxREAL);
    mxArray *it = mxCreateDoubleMatrix(S, 1, mxREAL);
    std::vector<int32_t> iters((size_t)S);
    char err[256] = {0};
    const int rc = epi_nnls_affine_fit_host(&d, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(a), mxGetPr(b), mxGetPr(me), iters.data(), NULL, /*device=*/0, err);
    if (rc != EPI_OK) { mxDestroyArray(a); mxDestroyArray(b); mxDestroyArray(me); mxDestroyArray(it); fail_if(rc, err); }
    for (size_t k = 0; k < iters.size(); k++) mxGetPr(it)[k] = (double)iters[k];
    mxArray *o[4] = {a, b, me, it};
    for (int k = 0; k < 4; k++)
        if (k < nlhs || k == 0) plhs[k] = o[k]; else mxDestroyArray(o[k]);
}

static void mc(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 10) mexErrMsgTxt("epiekf_pipeline_mex('mc', sp, u_min, n_scen, K, seed, z, J0_prefix, J1_prefix, prefix_days): 10 inputs expected");
    const mwSize R = mxGetM(prhs[1]), n = mxGetN(prhs[2]);
    want(prhs[1], R, EPI_SIM_PRM_COUNT, "sp");
    if (mxGetM(prhs[2]) != R) mexErrMsgTxt("u_min must be R x n_npi");
    epi_mc_desc d;
    memset(&d, 0, sizeof d);
    d.abi_version = EPIEKF_ABI_VERSION; d.R = (int32_t)R; d.n_scen = (int32_t)mxGetScalar(prhs[3]); d.K = (int32_t)mxGetScalar(prhs[4]); d.n_npi = (int32_t)n;
    const double seed = mxGetScalar(prhs[5]);
    d.seed_lo = (uint32_t)((uint64_t)seed & 0xFFFFFFFFu); d.seed_hi = (uint32_t)((uint64_t)seed >> 32);
    d.noise = mxIsEmpty(prhs[6]) ? 0 : 1; d.prefix_days = (int32_t)mxGetScalar(prhs[9]);
    if (d.n_scen < 1 || d.K < 1) mexErrMsgTxt("n_scen and K must be positive");
    const mwSize B = R * (mwSize)d.n_scen;
    if (d.noise && mxGetNumberOfElements(prhs[6]) != B * 3 * (mwSize)d.K) mexErrMsgTxt("z must be (n_scen*R) x 3 x K");
    if (d.prefix_days > 0 && (mxGetNumberOfElements(prhs[7]) != R || mxGetNumberOfElements(prhs[8]) != R)) mexErrMsgTxt("J0_prefix, J1_prefix must have one entry per region");
    mxArray *j0 = mxCreateDoubleMatrix(R, (mwSize)d.n_scen, mxREAL), *j1 = mxCreateDoubleMatrix(R, (mwSize)d.n_scen, mxREAL);
    mxArray *uo = nlhs > 2 ? dbl3(B, n, (mwSize)d.K) : NULL;
    char err[256] = {0};
    const int rc = epi_random_npi_mc_host(&d, mxGetPr(prhs[1]), mxGetPr(prhs[2]), opt(prhs[6]), d.prefix_days > 0 ? mxGetPr(prhs[7]) : NULL,
                                          d.prefix_days > 0 ? mxGetPr(prhs[8]) : NULL, uo ? mxGetPr(uo) : NULL, mxGetPr(j0), mxGetPr(j1), /*device=*/0, err);
    if (rc != EPI_OK) { mxDestroyArray(j0); mxDestroyArray(j1); mxDestroyArray(uo); fail_if(rc, err); }
    plhs[0] = j0;
    if (nlhs > 1) plhs[1] = j1; else mxDestroyArray(j1);
    if (nlhs > 2) plhs[2] = uo;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[16] = {0};
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd) != 0) mexErrMsgTxt("epiekf_pipeline_mex: first argument is the command string");
    mexAtExit(epi_host_pool_release);       // `clear mex` hands the library's pooled contexts, helper streams and worker threads back
    if (strcmp(cmd, "prescribe") == 0) prescribe(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "preprocess") == 0) preprocess(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "nnls") == 0) nnls(nlhs, plhs, nrhs, prhs);
    else if (strcmp(cmd, "mc") == 0) mc(nlhs, plhs, nrhs, prhs);
    else mexErrMsgTxt("epiekf_pipeline_mex: unknown command");
}
