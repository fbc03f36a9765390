"""The device stages of Tools/TrainPredictPrescribeNPI.m chained for ALL regions at once:

    data-set columns --preprocess--> x, R_v, u, I0                                   (:142-198, 201-202, 240)
      --SIAlphaModelEKF, zero input--> alpha estimate                                 (:203-248)
      --lsqnonneg + intercept loop--> a, b                                            (:251-276)
      --SIAlphaModelEKF, real inputs--> alpha estimate                                (:296-305)
      --lsqnonneg + intercept loop--> a2, b2                                          (:306-330)
      --SIAlphaModelEKF, last plan held over the horizon--> s/i/alpha_historic        (:351-362)
      --SIAlphaModelEKFOptControlled sweep over the cost weights--> u_opt_smooth      (:421-460)
      --SIalpha_Controlled + NPICost--> (J0, J1) per cost weight                      (:481-493)
      --Pareto front + optimum--> prescription per region                             (:624-633)

The reference walks the regions (and, inside, the 250 cost weights) in serial MATLAB loops; here every stage is one
batched call into libepiekf.so.  Only per-region vectors (I0, a, b, end-of-history state, cost prefixes) cross the
host between stages; the filter outputs stay in HBM.  `prescribe()` returns every intermediate so that the tests can
re-derive each stage from the previous one with the CPU oracle."""
from __future__ import annotations

import numpy as np
import torch

from . import batch, layout as L, synth


def filter_setup(I0, N, dt=1.0):
    """s_init, Q_w, Ps_init of the 3-state filter per region (TrainPredictPrescribeNPI.m:229-237) -> [3,S],[9,S],[9,S]."""
    S = N.shape[0]
    stds = np.stack([10.0 * I0 / N, 30.0 * I0 / N, np.full(S, 1e-2)])
    Q = np.zeros((9, S)); P0 = np.zeros((9, S))
    for d in range(3):
        Q[d * 3 + d] = dt ** 2 * stds[d] ** 2
        P0[d * 3 + d] = dt ** 2 * (10 * stds[d]) ** 2
    s_init = np.stack([(N - I0) / N, I0 / N, np.full(S, synth.ALPHA0)])
    return s_init, Q, P0


def _prm3(N, a, b, n_npi):
    S = N.shape[0]
    prm = synth._base_prm(S)
    prm[L.PRM_S_MIN] = synth.MIN_CASES / N
    prm[L.PRM_I_MIN] = synth.MIN_CASES / N
    prm[L.PRM_B] = b
    prm[L.PRM_A:L.PRM_A + n_npi] = a
    return prm


def workload3(x, R, u, N, I0, a, b):
    """SIAlphaModelEKF over all regions: x, R [T,S], u [T,n,S]; a [n,S], b [S]."""
    T, S = x.shape
    s_init, Q, P0 = filter_setup(I0, N)
    return synth.Workload(model="SIAlphaModelEKF", T=T, n_npi=u.shape[1], x=np.ascontiguousarray(x), u=np.ascontiguousarray(u),
                          R_series=np.ascontiguousarray(R), R_scalar=None, x_series=None, u_series=None,
                          prm=_prm3(N, a, b, u.shape[1]), s_init=s_init, Ps_init=P0, s_final=np.full((3, S), np.nan),
                          Ps_final=np.full((9, S), np.nan), Q=Q)


def sweep_region_inputs(N, I0, a, b, n, w_eff=1.0):
    """Per-REGION arguments of the 6-state sweep (TrainPredictPrescribeNPI.m:423-453): prm [61,S] (EPSILON row left 0),
    s_init [6,S], Ps_init, s_final, Ps_final, Q [36,S].  The cost weight is the only thing that differs between the P
    chains of a region."""
    S = N.shape[0]
    s3, Q3, P3 = filter_setup(I0, N)
    prm = _prm3(N, a, b, n)
    prm[L.PRM_W_EFF:L.PRM_W_EFF + n] = w_eff
    s_init = np.zeros((6, S)); s_init[:3] = s3
    Q = np.zeros((36, S)); P0 = np.zeros((36, S))
    for d in range(3):
        Q[d * 6 + d] = Q3[d * 3 + d]; P0[d * 6 + d] = P3[d * 3 + d]
    for d in range(3, 6):
        Q[d * 6 + d] = synth.Q_LAMBDA ** 2; P0[d * 6 + d] = 10.0 * synth.Q_LAMBDA ** 2
    s_final = np.full((6, S), np.nan); s_final[3:] = 0.0
    Ps_final = np.zeros((36, S))
    for i in range(3):
        for j in range(3):
            Ps_final[i + 6 * j] = np.nan
    for d in range(3, 6):
        Ps_final[d * 6 + d] = 1e-8
    return dict(prm=prm, s_init=s_init, Ps_init=P0, s_final=s_final, Ps_final=Ps_final, Q=Q)


def workload6(x, R, u, N, I0, a, b, eps_grid, w_eff=1.0):
    """SIAlphaModelEKFOptControlled sweep: chain c = region * n_eps + e (:421-460).  x, R [T,S] with NaN over the
    horizon, u [T,n,S] with NaN over the horizon."""
    T, S = x.shape
    n, P = u.shape[1], eps_grid.shape[0]
    rr = np.repeat(np.arange(S), P)
    reg = sweep_region_inputs(N, I0, a, b, n, w_eff)
    prm = reg["prm"][:, rr]
    prm[L.PRM_EPSILON] = np.tile(eps_grid, S)
    return synth.Workload(model="SIAlphaModelEKFOptControlled", T=T, n_npi=n, x=np.ascontiguousarray(x),
                          u=np.ascontiguousarray(u), R_series=np.ascontiguousarray(R), R_scalar=None,
                          x_series=rr.astype(np.int32), u_series=rr.astype(np.int32), prm=prm, s_init=reg["s_init"][:, rr],
                          Ps_init=reg["Ps_init"][:, rr], s_final=reg["s_final"][:, rr], Ps_final=reg["Ps_final"][:, rr],
                          Q=reg["Q"][:, rr])


def scoring_region_inputs(hist_end, a, b, u_max, wts):
    """Per-region scoring block sp [48,S] (EPI_SIM_* rows) of the sweep's tail (:481): end-of-history state, model
    constants, NPI_MAXES, NPICost weights."""
    n, S = a.shape
    sp = np.zeros((batch.SIM_PRM_COUNT, S))
    sp[0:3] = hist_end
    sp[3], sp[4], sp[5] = synth.ALPHA_MIN, synth.ALPHA_MAX, synth.MODEL_GAMMA
    sp[6], sp[7], sp[11] = b, synth.MODEL_BETA, 1.0
    sp[batch.SIM_A:batch.SIM_A + n] = a
    sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = u_max[:, None]
    sp[batch.SIM_W:batch.SIM_W + n] = wts
    return sp


def _alpha_smooth(w, device):
    dw = batch.DeviceWorkload(w, device)
    r = batch.EkfRunner(dw, outputs=["S_SMOOTH"])
    r.run()
    torch.cuda.synchronize(dw.device)
    return r.out["S_SMOOTH"].cpu().numpy()                 # [T, 3, S]


def prescribe(cases, deaths, population, ip, horizon=30, n_eps=50, num_regression_days=60, npi_weights=None,
              W=7, device="cuda:0"):
    """Run the chain above.  cases/deaths [T,S] cumulative counts (NaN = missing), population [S], ip [T,n,S] (NaN = N/A).
    Returns a dict with every intermediate and `prescription` [horizon, n, S]: the smoothed optimal plan of each region's
    Pareto optimum (`I_opt`), plus `front` [S, n_eps] and (J0, J1) [S, n_eps]."""
    T, S = cases.shape
    n = ip.shape[1]
    N = np.asarray(population, dtype=np.float64)
    u_max = synth.IP_MAXES[:n]
    out = {}
    pre = {k: v.cpu().numpy() for k, v in batch.preprocess(cases, N, deaths, ip, W=W, min_cases=synth.MIN_CASES,
                                                            first_num_days=7, device=device).items()}
    out["pre"] = pre
    x, R, u, I0 = pre["x_new"], pre["R_v"], pre["ip_filled"], pre["I0"]
    # round 1: zero input, a = 0, b = 0 -> alpha estimate -> regression
    S1 = _alpha_smooth(workload3(x, R, np.zeros_like(u), N, I0, np.zeros((n, S)), np.zeros(S)), device)
    D = min(num_regression_days, T)
    X = np.ascontiguousarray(u_max[None, :, None] - u[T - D:])
    fit1 = {k: v.cpu().numpy() for k, v in batch.nnls_affine_fit(X, np.ascontiguousarray(S1[T - D:, 2]), device=device).items()}
    # round 2: real inputs -> refined alpha -> second regression
    S2 = _alpha_smooth(workload3(x, R, u, N, I0, fit1["a"], fit1["b"]), device)
    fit2 = {k: v.cpu().numpy() for k, v in batch.nnls_affine_fit(X, np.ascontiguousarray(S2[T - D:, 2]), device=device).items()}
    out.update(alpha_round1=S1[:, 2], fit1=fit1, alpha_round2=S2[:, 2], fit2=fit2)
    # forecast set-up (:333-341): R_v padded with its mean, observations and (for the sweep) controls NaN over the horizon
    R_mean = R.sum(axis=0) / T
    xh = np.concatenate([x, np.full((horizon, S), np.nan)]); Rh = np.concatenate([R, np.repeat(R_mean[None], horizon, 0)])
    u_fixed = np.concatenate([u, np.repeat(u[-1:], horizon, 0)])                       # last plan held (:351-353)
    Sf = _alpha_smooth(workload3(xh, Rh, u_fixed, N, I0, fit2["a"], fit2["b"]), device)
    hist = Sf[:T]                                                                      # s/i/alpha_historic (:355-357)
    out.update(R_mean=R_mean, historic=hist)
    # Pareto sweep over the cost weights
    eps_grid = synth.epsilon_grid(n_eps)
    u_nan = np.concatenate([u, np.full((horizon, n, S), np.nan)])
    w6 = workload6(xh, Rh, u_nan, N, I0, fit2["a"], fit2["b"], eps_grid)
    dw = batch.DeviceWorkload(w6, device)
    runner = batch.EkfRunner(dw, outputs=["u_opt_smooth", "S_SMOOTH"], lane_block="auto")   # chain-blocked outputs
    runner.run()
    uos = runner.out["u_opt_smooth"]                                                   # stays in HBM (blocked layout)
    # scoring (:481-493): simulate the horizon from the end-of-history state, NPICost over [historic, horizon]
    wts = np.ones((n, S)) if npi_weights is None else np.asarray(npi_weights, dtype=np.float64)
    rr = np.repeat(np.arange(S), n_eps)
    sp_region = scoring_region_inputs(hist[T - 1], fit2["a"], fit2["b"], u_max, wts)
    sp = sp_region[:, rr]
    J0p_region = np.cumsum(hist[:, 0] * hist[:, 1] * hist[:, 2], axis=0)[-1]           # sequential historic sums
    J1p_region = np.cumsum((wts[None] * u).reshape(T * n, S), axis=0)[-1]
    J0p, J1p = J0p_region[rr], J1p_region[rr]
    out.update(sp_region=sp_region, J0_prefix_region=J0p_region, J1_prefix_region=J1p_region, x_sweep=xh, R_sweep=Rh, u_sweep=u_nan,
               sweep_region=sweep_region_inputs(N, I0, fit2["a"], fit2["b"], n), I0=I0, u_fixed=u_fixed, X_reg=X)
    sc = batch.score_sweep(uos, T, sp, J0p, J1p, B=S * n_eps)
    front, i_opt = batch.pareto_front(sc["J0"], sc["J1"], S)
    torch.cuda.synchronize(dw.device)
    i_opt_h = i_opt.cpu().numpy()
    chains = np.arange(S) * n_eps + i_opt_h
    if uos.dim() == 4:            # blocked [T+H, nblk, n, blk]: pick (block, lane) of every optimum chain
        cb = torch.as_tensor(chains // runner.blk, device=uos.device); cr = torch.as_tensor(chains % runner.blk, device=uos.device)
        best = uos[T:][:, cb, :, cr].permute(1, 2, 0)          # index dims come first: [S, H, n] -> [H, n, S]
    else:
        best = uos[T:].index_select(2, torch.as_tensor(chains, device=uos.device))
    out.update(eps_grid=eps_grid, sp=sp, J0_prefix=J0p, J1_prefix=J1p,
               J0=sc["J0"].cpu().numpy().reshape(S, n_eps), J1=sc["J1"].cpu().numpy().reshape(S, n_eps),
               front=front.cpu().numpy(), i_opt=i_opt_h, sweep=w6,
               prescription=best.cpu().numpy())
    return out
