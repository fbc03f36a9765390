"""Host-side mirror of the reference's Tools/ functions for the EKF/EKS hot path.

Same names, argument order, shapes and error behaviour as the MATLAB functions; NumPy arrays stand in
for MATLAB arrays (u: n_npi x T, x: 1 x T, S: m x T, P: m x m x T, K_GAIN: m x 1 x T).  Every call
goes through the C ABI of libepiekf.so (epi_ekf_run_host) -- the same entry point the MEX gateway
in matlab/ binds.  There is no CPU implementation behind these functions.

  SIAlphaModelEKF                      Tools/SIAlphaModelEKF.m:1
  SIAlphaModelEKFOptControlled         Tools/SIAlphaModelEKFOptControlled.m:1
  SIAlphaModelBackwardEKF              Tools/SIAlphaModelBackwardEKF.m:1
  SIAlphaModelBackwardEKFOptControlled Tools/SIAlphaModelBackwardEKFOptControlled.m:1
  NewCaseEKFEstimatorWithOptimalNPI    Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1 (10 outputs)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import layout as L
from ._lib import EpiError

__all__ = ["SIAlphaModelEKF", "SIAlphaModelEKFOptControlled", "SIAlphaModelBackwardEKF",
           "SIAlphaModelBackwardEKFOptControlled", "NewCaseEKFEstimatorWithOptimalNPI",
           "NewCaseEKFEstimatorWithOptimalNPI_codegen", "EpiError", "resolve_w"]

# params fields each reference model actually reads (a missing one is a MATLAB run-time error there)
_FIELDS3 = ("dt", "a", "b", "u_max", "alpha_min", "alpha_max", "gamma", "beta", "obs_type")
_FIELDS = {
    "SIAlphaModelEKF": _FIELDS3 + ("s_min", "i_min"),
    "SIAlphaModelBackwardEKF": _FIELDS3,
    "SIAlphaModelEKFOptControlled": _FIELDS3 + ("u_min", "sigma", "epsilon", "w"),
    "SIAlphaModelBackwardEKFOptControlled": _FIELDS3 + ("u_min", "sigma", "epsilon", "w"),
    "NewCaseEKFEstimatorWithOptimalNPI": _FIELDS3 + ("u_min", "sigma", "epsilon", "w"),
    "NewCaseEKFEstimatorWithOptimalNPI_codegen": _FIELDS3[:-1] + ("u_min", "sigma", "epsilon", "w"),
}


def _get(params, k):
    try:
        return params[k] if isinstance(params, dict) else getattr(params, k)
    except (KeyError, AttributeError):
        raise KeyError(f'Reference to non-existent field "{k}".') from None


def resolve_w(w, n: int) -> np.ndarray:
    """params.w as the 6-state callbacks see it: `phi(kk)` is a LINEAR index into
    `epsilon*w - gamma*s6*a` with `a` n x 1 (Tools/SIAlphaModelEKFOptControlled.m:49,107).
    w n x 1 -> w(kk); w 1 x n (row, as testPrescribeXPRIZE02.m:56 passes) -> n x n matrix whose
    linear indices 1..n are column 1 -> w(1) for every kk; w n x D -> w(kk,1); scalar -> scalar."""
    w = np.asarray(w, dtype=np.float64)
    if w.size == 1:
        return np.full(n, float(w.reshape(-1)[0]))
    if w.ndim == 1:
        w = w.reshape(-1, 1)
    if w.ndim != 2:
        raise ValueError("params.w must be a scalar, vector or matrix")
    if w.shape[0] == 1:
        return np.full(n, float(w[0, 0]))
    if w.shape[0] != n:
        raise ValueError("Arrays have incompatible sizes for this operation.")
    return np.ascontiguousarray(w[:, 0], dtype=np.float64)


def _run(model, u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma,
         inv_monitor_len, order, device=0):
    m = L.MODEL_DIM[model]
    u = np.asarray(u, dtype=np.float64)
    if u.ndim == 1:
        u = u.reshape(-1, 1)
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 2 and x.shape[0] != 1:
        raise ValueError("this engine implements the scalar-observation filter: size(x,1) must be 1")
    x = x.reshape(-1)
    T = x.shape[0]
    nn = u.shape[0]
    if u.shape[1] != T:
        raise ValueError("size(u,2) must equal size(x,2)")
    s_init = np.asarray(s_init, dtype=np.float64).reshape(-1)
    if s_init.shape[0] != m:
        raise ValueError(f"{model}: length(s_init) must be {m}")
    for k in _FIELDS[model]:
        _get(params, k)
    prm = np.zeros((L.PRM_COUNT, 1))
    six = m == 6
    prm[L.PRM_DT] = _get(params, "dt"); prm[L.PRM_BETA] = _get(params, "beta"); prm[L.PRM_GAMMA] = _get(params, "gamma")
    prm[L.PRM_B] = _get(params, "b")
    prm[L.PRM_ALPHA_MIN] = _get(params, "alpha_min"); prm[L.PRM_ALPHA_MAX] = _get(params, "alpha_max")
    if model == "SIAlphaModelEKF":
        prm[L.PRM_S_MIN] = _get(params, "s_min"); prm[L.PRM_I_MIN] = _get(params, "i_min")
    vec = lambda k: np.asarray(_get(params, k), dtype=np.float64).reshape(-1)
    a, u_max = vec("a"), vec("u_max")
    if a.shape[0] != nn or u_max.shape[0] != nn:
        raise ValueError("Incorrect dimensions for matrix multiplication: params.a / params.u_max vs u")
    prm[L.PRM_A:L.PRM_A + nn, 0] = a
    prm[L.PRM_U_MAX:L.PRM_U_MAX + nn, 0] = u_max
    if six:
        prm[L.PRM_SIGMA] = _get(params, "sigma"); prm[L.PRM_EPSILON] = _get(params, "epsilon")
        prm[L.PRM_U_MIN:L.PRM_U_MIN + nn, 0] = vec("u_min")
        prm[L.PRM_W_EFF:L.PRM_W_EFF + nn, 0] = resolve_w(_get(params, "w"), nn)
    prm[L.PRM_V_BAR] = float(np.asarray(v_bar).reshape(-1)[0])
    prm[L.PRM_BETA_EKF] = float(beta); prm[L.PRM_GAMMA_EKF] = float(gamma)

    # Q_w: GenericExtendedKalmanFilter.m:63-76.  The first test (size(Q_w,1) == size(Q_w,2)) also catches
    # m x m x D arrays: Q = repmat(Q_w,1,1,T) makes Q(:,:,k) = Q_w(:,:,mod(k-1,D)+1), i.e. the pages are used
    # cyclically (time-varying when D == T); a length-T vector is q(k) in B*Q*B' with B = eye(m).
    Q = np.asarray(Q_w, dtype=np.float64)
    generic = not model.startswith("NewCase")
    q_mode, Qt = 0, None
    if Q.ndim <= 2 and Q.size == 1:
        Qm = float(Q.reshape(-1)[0]) * np.eye(m)          # B*Q*B' with B = eye(m)
    elif Q.ndim == 2 and Q.shape == (m, m):
        Qm = Q
    elif generic and Q.ndim == 3 and Q.shape[:2] == (m, m):
        q_mode, Qt = 1, Q[:, :, np.arange(T) % Q.shape[2]]
    elif generic and Q.ndim == 3 and Q.shape[:2] == (1, 1):
        q_mode, Qt = 1, Q[0, 0, np.arange(T) % Q.shape[2]][None, None, :] * np.eye(m)[:, :, None]
    elif generic and (Q.ndim == 1 or (Q.ndim == 2 and min(Q.shape) == 1)) and Q.size == T:
        q_mode, Qt = 1, Q.reshape(-1)[None, None, :] * np.eye(m)[:, :, None]
    else:
        raise EpiError(-2, "Process noise covariance noise mismatch")
    # R_v: GenericExtendedKalmanFilter.m:79-91
    R = np.asarray(R_v, dtype=np.float64)
    if R.size == 1:
        r_mode, R_scalar, R_series = 0, np.array([float(R.reshape(-1)[0])]), None
    elif generic and (R.ndim == 1 or (R.ndim == 2 and min(R.shape) == 1)) and R.size == T:   # isvector && length == T
        r_mode, R_scalar, R_series = 1, None, np.ascontiguousarray(R.reshape(T, 1))
    else:
        raise EpiError(-3, "Observation noise covariance noise mismatch")

    col = lambda v, n: np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1)[:, None]) if np.asarray(v).size == n else None
    fcol = lambda Mx: np.ascontiguousarray(np.asarray(Mx, dtype=np.float64).reshape(m, m).reshape(-1, order="F")[:, None])
    si, sf = col(s_init, m), col(s_final, m)
    if sf is None:
        raise ValueError(f"length(s_final) must be {m}")
    Pi, Pf = fcol(Ps_init), fcol(Ps_final)
    if q_mode:   # [T][m*m][1], pages column-major like every matrix of the ABI
        Qc = np.ascontiguousarray(Qt.transpose(2, 1, 0).reshape(T, m * m, 1))
    else:
        Qc = fcol(Qm)
    xs = np.ascontiguousarray(x.reshape(T, 1))
    us = np.ascontiguousarray(u.T.reshape(T, nn, 1))      # [T][n_npi][1] == MATLAB column-major n_npi x T

    has_uos = generic
    names = ["u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH",
             "K_GAIN", "innovations", "rho"]
    if not has_uos:
        names.remove("u_opt_smooth")
    mask = 0
    for n_ in names:
        mask |= L.OUT_BITS[n_]
    desc = _lib.make_desc(model, 1, T, 1, 1, nn, int(inv_monitor_len), int(order), _get(params, "obs_type") if "obs_type" in _FIELDS[model] else "NEWCASES", r_mode, mask, q_mode)
    out = {n_: np.zeros((T, max(L.out_rows(n_, m, nn), 1), 1)) for n_ in names}
    ins, outs = _lib.Inputs(), _lib.Outputs()
    keep = [xs, us, prm, si, sf, Pi, Pf, Qc, R_scalar, R_series]
    ins.x, ins.u, ins.prm = xs.ctypes.data, us.ctypes.data, prm.ctypes.data
    ins.s_init, ins.s_final, ins.Ps_init, ins.Ps_final, ins.Q = si.ctypes.data, sf.ctypes.data, Pi.ctypes.data, Pf.ctypes.data, Qc.ctypes.data
    ins.R_scalar = R_scalar.ctypes.data if R_scalar is not None else None
    ins.R_series = R_series.ctypes.data if R_series is not None else None
    for n_ in names:
        setattr(outs, n_, out[n_].ctypes.data)
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_ekf_run_host(C.byref(desc), C.byref(ins), C.byref(outs), _dev_index(device), err)
    _lib.check(rc, err)
    del keep
    S = lambda n_: np.asfortranarray(out[n_][:, :, 0].T)
    P = lambda n_: np.asfortranarray(out[n_][:, :, 0].reshape(T, m, m).transpose(2, 1, 0))
    res = {
        "u_opt": S("u_opt"), "S_MINUS": S("S_MINUS"), "S_PLUS": S("S_PLUS"), "S_SMOOTH": S("S_SMOOTH"),
        "P_MINUS": P("P_MINUS"), "P_PLUS": P("P_PLUS"), "P_SMOOTH": P("P_SMOOTH"),
        "K_GAIN": np.asfortranarray(out["K_GAIN"][:, :, 0].T.reshape(m, 1, T)),
        "innovations": out["innovations"][:, 0, 0].reshape(1, T).copy(),
        "rho": out["rho"][:, 0, 0].reshape(T, 1).copy(),     # squeeze(rho): T x 1
    }
    if has_uos:
        res["u_opt_smooth"] = S("u_opt_smooth")
    return res


def _eleven(r):
    return (r["u_opt"], r["u_opt_smooth"], r["S_MINUS"], r["S_PLUS"], r["S_SMOOTH"], r["P_MINUS"], r["P_PLUS"],
            r["P_SMOOTH"], r["K_GAIN"], r["innovations"], r["rho"])


def SIAlphaModelEKF(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma,
                    inv_monitor_len, order):
    """[u_opt, u_opt_smooth, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH, K_GAIN, innovations, rho]
    = SIAlphaModelEKF(...)  -- Tools/SIAlphaModelEKF.m:1"""
    return _eleven(_run("SIAlphaModelEKF", u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w,
                        R_v, beta, gamma, inv_monitor_len, order))


def SIAlphaModelEKFOptControlled(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta,
                                 gamma, inv_monitor_len, order):
    """Tools/SIAlphaModelEKFOptControlled.m:1 (NaN entries of u are replaced by the bang-bang control)."""
    return _eleven(_run("SIAlphaModelEKFOptControlled", u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar,
                        v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order))


def SIAlphaModelBackwardEKF(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma,
                            inv_monitor_len, order):
    """Tools/SIAlphaModelBackwardEKF.m:1 (time-flipped 3-state filter)."""
    return _eleven(_run("SIAlphaModelBackwardEKF", u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar,
                        Q_w, R_v, beta, gamma, inv_monitor_len, order))


def SIAlphaModelBackwardEKFOptControlled(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v,
                                         beta, gamma, inv_monitor_len, order):
    """Tools/SIAlphaModelBackwardEKFOptControlled.m:1 (time-flipped 6-state filter)."""
    return _eleven(_run("SIAlphaModelBackwardEKFOptControlled", u, x, params, s_init, Ps_init, s_final, Ps_final,
                        w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order))


def NewCaseEKFEstimatorWithOptimalNPI(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v,
                                      beta, gamma, inv_monitor_len, order):
    """[u_opt, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH, K_GAIN, innovations, rho]
    -- Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1 (Tools/ output order, 10 outputs)."""
    r = _run("NewCaseEKFEstimatorWithOptimalNPI", u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar,
             Q_w, R_v, beta, gamma, inv_monitor_len, order)
    return (r["u_opt"], r["S_MINUS"], r["S_PLUS"], r["S_SMOOTH"], r["P_MINUS"], r["P_PLUS"], r["P_SMOOTH"],
            r["K_GAIN"], r["innovations"], r["rho"])


def NewCaseEKFEstimatorWithOptimalNPI_codegen(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w,
                                              R_v, beta, gamma, inv_monitor_len, order):
    """[u_opt, S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho]
    -- MatlabCodeGenerator/NewCaseEKFEstimatorWithOptimalNPI.m:1 (codegen output order; observation
    clamp is the identity and the observation is always NEWCASES there)."""
    r = _run("NewCaseEKFEstimatorWithOptimalNPI_codegen", u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar,
             v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order)
    return (r["u_opt"], r["S_MINUS"], r["S_PLUS"], r["P_MINUS"], r["P_PLUS"], r["K_GAIN"], r["S_SMOOTH"],
            r["P_SMOOTH"], r["innovations"], r["rho"])


def Rt_ExpFitEKF(x, s_init, params, w_bar, v_bar, Ps_init, Q_w, R_v, beta, gamma, inv_monitor_len, order, device=0):
    """[S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho] = Rt_ExpFitEKF(...)
    -- Tools/Rt_ExpFitEKF.m:1 (exponential-fit EKF/EKS over the new-case counts; order 2 adds the Hessian terms).
    x is 1 x T; params = [time_scale, alpha, sigma]."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 2 and x.shape[0] != 1:
        raise ValueError("this engine implements the scalar-observation filter: size(x,1) must be 1")
    x = np.ascontiguousarray(x.reshape(-1, 1))
    T = x.shape[0]
    s_init = np.asarray(s_init, dtype=np.float64).reshape(-1)
    params = np.asarray(params, dtype=np.float64).reshape(-1)
    w_bar = np.asarray(w_bar, dtype=np.float64).reshape(-1)
    if s_init.shape[0] != 2 or w_bar.shape[0] < 2:
        raise ValueError("Rt_ExpFitEKF: s_init and w_bar must have 2 elements")
    if params.shape[0] < 3:
        raise IndexError("Index exceeds the number of array elements.")      # params(3), Rt_ExpFitEKF.m:136
    rp = np.zeros((19, 1))
    rp[0:3, 0] = params[:3]; rp[3:5, 0] = w_bar[:2]
    rp[5] = float(np.asarray(v_bar).reshape(-1)[0]); rp[6] = float(np.asarray(R_v).reshape(-1)[0])
    rp[7], rp[8] = float(beta), float(gamma)
    rp[9:11, 0] = s_init
    rp[11:15, 0] = np.asarray(Ps_init, dtype=np.float64).reshape(2, 2).reshape(-1, order="F")
    rp[15:19, 0] = np.asarray(Q_w, dtype=np.float64).reshape(2, 2).reshape(-1, order="F")
    d = _lib.RtDesc()
    d.abi_version, d.B, d.T, d.Sx, d.L, d.order = _lib.ABI_VERSION, 1, T, 1, int(inv_monitor_len), int(order)
    rows = {"S_MINUS": 2, "S_PLUS": 2, "P_MINUS": 4, "P_PLUS": 4, "K_GAIN": 2, "S_SMOOTH": 2, "P_SMOOTH": 4,
            "innovations": 1, "rho": 1}
    out = {n: np.zeros((T, r, 1)) for n, r in rows.items()}
    outs = _lib.RtOutputs()
    for n in rows:
        setattr(outs, n, out[n].ctypes.data)
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_rt_expfit_run_host(C.byref(d), None, x.ctypes.data, rp.ctypes.data, C.byref(outs), _dev_index(device), err)
    _lib.check(rc, err)
    S = lambda n: np.asfortranarray(out[n][:, :, 0].T)
    P = lambda n: np.asfortranarray(out[n][:, :, 0].reshape(T, 2, 2).transpose(2, 1, 0))
    return (S("S_MINUS"), S("S_PLUS"), P("P_MINUS"), P("P_PLUS"), np.asfortranarray(out["K_GAIN"][:, :, 0].T.reshape(2, 1, T)),
            S("S_SMOOTH"), P("P_SMOOTH"), out["innovations"][:, 0, 0].reshape(1, T).copy(),
            out["rho"][:, 0, 0].reshape(T, 1).copy())          # squeeze(rho): T x 1


# ---------------------------------------------------------------------------------------------------------------------
# forward simulators and the NPI cost: one call = one chain through the host-pointer entry
# points epi_sialpha_sim_host, epi_seirp_sim_host, epi_npi_cost_host (what matlab/epiekf_sim_mex.cpp binds too)
# ---------------------------------------------------------------------------------------------------------------------
_SIM_FIELDS = {"s0": 0, "i0": 1, "alpha0": 2, "alpha_min": 3, "alpha_max": 4, "gamma": 5, "b": 6, "beta": 7,
               "s_noise_std": 8, "i_noise_std": 9, "alpha_noise_std": 10, "dt": 11}
_SIM_A, _SIM_U_MAX, _SIM_PRM_COUNT = 12, 24, 48


def _vp(a):
    return None if a is None else a.ctypes.data


def _dev_index(device):
    """0, "cuda:0" or a torch.device -> the HIP device ordinal."""
    if isinstance(device, int):
        return device
    if not isinstance(device, str):
        idx = getattr(device, "index", None)          # torch.device
        if isinstance(idx, int):
            return idx
    t = str(device)
    return int(t.rsplit(":", 1)[1]) if ":" in t else 0


def _matlab_round(v):
    """MATLAB round(): halves away from zero."""
    return int(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))


def _per_step(name, v, need):
    """A per-step parameter array indexed (1 : need) by the .m loop: shorter => MATLAB's index error."""
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.shape[0] < need:
        raise IndexError(f"Index exceeds the number of array elements ({name}).")
    return a


def SIalpha_Controlled(u, s0, i0, alpha0, u_max, alpha_min, alpha_max, gamma, a, b, beta, s_noise_std, i_noise_std,
                       alpha_noise_std, K, dt, noise=None, rng=None, device=0):
    """[s, i, alpha] = SIalpha_Controlled(u, s0, i0, alpha0, u_max, alpha_min, alpha_max, gamma, a, b, beta,
    s_noise_std, i_noise_std, alpha_noise_std, K, dt) -- Tools/SIalpha_Controlled.m:1 (forward Euler, clamps, the
    initial sample dropped :30-32).  u is n_npi x K.  The .m draws three randn per step from MATLAB's global stream,
    which cannot be reproduced outside MATLAB: pass `noise` (3 x K standard-normal draws, rows = s, i, alpha) for a
    reproducible run; otherwise they come from `rng` (numpy Generator, default: a fresh one) when any noise std is
    non-zero.  Returns three 1 x K arrays."""
    K = int(K)
    u = np.asarray(u, dtype=np.float64)
    if u.ndim != 2 or u.shape[1] < K:
        raise IndexError("Index in position 2 exceeds array bounds (u).")                  # u(:, t), :27
    n = u.shape[0]
    a = np.asarray(a, dtype=np.float64).reshape(-1); u_max = np.asarray(u_max, dtype=np.float64).reshape(-1)
    if a.shape[0] != n or u_max.shape[0] != n:
        raise ValueError("Incorrect dimensions for matrix multiplication (a'*(u_max - u(:, t))).")
    if n > 12:
        raise ValueError("this engine supports at most 12 NPIs")
    sp = np.zeros((_SIM_PRM_COUNT, 1))
    for name, v in (("s0", s0), ("i0", i0), ("alpha0", alpha0), ("alpha_min", alpha_min), ("alpha_max", alpha_max),
                    ("gamma", gamma), ("b", b), ("beta", beta), ("s_noise_std", s_noise_std), ("i_noise_std", i_noise_std),
                    ("alpha_noise_std", alpha_noise_std), ("dt", dt)):
        sp[_SIM_FIELDS[name], 0] = float(v)
    sp[_SIM_A:_SIM_A + n, 0] = a
    sp[_SIM_U_MAX:_SIM_U_MAX + n, 0] = u_max
    z = None
    if noise is not None:
        z = np.asarray(noise, dtype=np.float64)
        if z.shape != (3, K):
            raise ValueError("noise must be 3 x K")
    elif float(s_noise_std) != 0.0 or float(i_noise_std) != 0.0 or float(alpha_noise_std) != 0.0:
        z = (rng or np.random.default_rng()).standard_normal((3, K))
    zz = None if z is None else np.ascontiguousarray(z.T)                 # [K][3][1]
    uu = np.ascontiguousarray(u[:, :K].T)                                 # [K][n_npi][1]
    d = _lib.SimDesc()
    d.abi_version, d.B, d.K, d.Su, d.n_npi, d.noise = _lib.ABI_VERSION, 1, K, 1, n, int(zz is not None)
    out = [np.zeros((1, K)) for _ in range(3)]
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_sialpha_sim_host(C.byref(d), None, _vp(uu), _vp(sp), _vp(zz), _vp(out[0]), _vp(out[1]), _vp(out[2]),
                                         None, None, _dev_index(device), err)
    _lib.check(rc, err)
    return tuple(out)


def SI_Controlled(alpha, beta, s0, i0, K, dt, device=0):
    """[s, i] = SI_Controlled(alpha, beta, s0, i0, K, dt) -- Tools/SI_Controlled.m:1 (2-state forward Euler with a
    time-dependent infection rate alpha(1 : K-1); K samples, the first one is the initial condition)."""
    K = int(K)
    if K < 1:
        raise IndexError("Index exceeds the number of array elements (s(1) = s0 with K = 0).")
    al = np.ascontiguousarray(_per_step("alpha", alpha, K - 1)[:max(K - 1, 1)]) if K > 1 else np.zeros(1)
    prm = np.array([[float(beta)], [float(s0)], [float(i0)]])
    s, i = np.zeros((1, K)), np.zeros((1, K))
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_si_controlled_host(1, K, 1, float(dt), None, _vp(al), _vp(prm), _vp(s), _vp(i), _dev_index(device), err)
    _lib.check(rc, err)
    return s, i


def _seirp(per_step, init, T, dt, sat, device):
    K = _matlab_round(float(T) / float(dt))
    if K < 1:
        raise IndexError("Index exceeds the number of array elements (s(1) = s0 with K = 0).")
    par = np.zeros((K, 7, 1))
    for j, (name, v) in enumerate(per_step):
        if v is None:
            continue
        a = _per_step(name, v, K - 1)
        par[:min(K, a.shape[0]), j, 0] = a[:K]
    ini = np.ascontiguousarray(np.asarray(init, dtype=np.float64).reshape(5, 1))
    st = None if sat is None else np.ascontiguousarray(np.asarray(sat, dtype=np.float64).reshape(6, 1))
    o = np.zeros((K, 5, 1))
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_seirp_sim_host(1, K, K, float(dt), int(st is not None), 0, _vp(par), _vp(ini), _vp(st), _vp(o),
                                       _dev_index(device), err)
    _lib.check(rc, err)
    return tuple(o[:, q, 0].reshape(1, K).copy() for q in range(5))


def SEIRP(alpha_e, alpha_i, kappa, rho, beta, mu, gamma, s0, e0, i0, r0, p0, T, dt, device=0):
    """[s, e, i, r, p] = SEIRP(alpha_e, alpha_i, kappa, rho, beta, mu, gamma, s0, e0, i0, r0, p0, T, dt)
    -- Tools/SEIRP.m:1 (forward Euler; K = round(T/dt) samples, the first one is the initial condition; the seven
    parameters are per-step arrays of at least K-1 elements)."""
    return _seirp((("alpha_e", alpha_e), ("alpha_i", alpha_i), ("kappa", kappa), ("rho", rho), ("beta", beta), ("mu", mu),
                   ("gamma", gamma)), (s0, e0, i0, r0, p0), T, dt, None, device)


def SEIRPSaturatedResource(alpha_e, alpha_i, kappa, rho, gamma, s0, e0, i0, r0, p0, T, dt, beta_0, beta_s, mu_0, mu_s,
                           sigma, i_0, device=0):
    """[s, e, i, r, p] = SEIRPSaturatedResource(alpha_e, alpha_i, kappa, rho, gamma, s0, e0, i0, r0, p0, T, dt, beta_0,
    beta_s, mu_0, mu_s, sigma, i_0) -- Tools/SEIRPSaturatedResource.m:1 (recovery and death rates gated by
    tanh((i - i_0)/sigma), :27-29)."""
    return _seirp((("alpha_e", alpha_e), ("alpha_i", alpha_i), ("kappa", kappa), ("rho", rho), ("beta", None), ("mu", None),
                   ("gamma", gamma)), (s0, e0, i0, r0, p0), T, dt, (beta_0, beta_s, mu_0, mu_s, sigma, i_0), device)


def _npi_cost_host(nc, u_tn, w, device):
    """nc [T], u_tn [T][n], w [T][n] or [n] -> (J0, J1) through epi_npi_cost_host (one chain)."""
    T, n = u_tn.shape
    J0, J1 = np.zeros(1), np.zeros(1)
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_npi_cost_host(1, T, n, 1, int(w.ndim == 2), None, _vp(nc), _vp(u_tn), _vp(w), _vp(J0), _vp(J1),
                                      _dev_index(device), err)
    _lib.check(rc, err)
    return float(J0[0]), float(J1[0])


def NPICost(newcases, inputs, weights, device=0):
    """[J0, J1] = NPICost(newcases, inputs, weights) -- Tools/NPICost.m:1: J0 = mean(newcases),
    J1 = mean(weights(:).*inputs(:)).  inputs and weights are n_npi x T (weights may also be n_npi x 1: implicit
    expansion of `weights .* inputs`)."""
    nc = np.ascontiguousarray(np.asarray(newcases, dtype=np.float64).reshape(-1))
    u = np.asarray(inputs, dtype=np.float64); w = np.asarray(weights, dtype=np.float64)
    if u.ndim != 2:
        raise ValueError("inputs must be n_npi x T")
    n, T = u.shape
    if w.ndim == 1:
        w = w.reshape(-1, 1)
    if w.shape not in ((n, T), (n, 1)):
        raise ValueError("Arrays have incompatible sizes for this operation (weights .* inputs).")
    u_tn = np.ascontiguousarray(u.T)
    wd = np.ascontiguousarray(w.T) if (w.shape[1] == T and T > 1) else np.ascontiguousarray(w[:, 0])
    if nc.shape[0] == T:
        return _npi_cost_host(nc, u_tn, wd, device)
    # newcases and inputs of different lengths are legal in the .m (two independent means): two calls
    J0, _ = _npi_cost_host(nc, np.zeros((nc.shape[0], 1)), np.zeros(1), device)
    _, J1 = _npi_cost_host(np.zeros(T), u_tn, wd, device)
    return J0, J1
