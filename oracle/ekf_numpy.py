"""ekf_numpy.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE ("parity unpinned").

Second, independent restatement of the reference's EKF/EKS hot path, written
straight from the MATLAB sources with NumPy matrix algebra and LAPACK for the
two MATLAB built-ins that matter (``pinv`` -> SVD, ``mrdivide`` -> LU).  It is
slow (pure Python time loop) and is used only by tests/ and by
tests/golden/make_golden.py to cross-check oracle/ekf_oracle.c: two readings of
the .m files must agree before the GPU path is compared with either.

Follows, function by function:
  Tools/GenericExtendedKalmanFilter.m:1-234
  Tools/SIAlphaModelEKF.m:27-109, Tools/SIAlphaModelEKFOptControlled.m:27-168
  Tools/SIAlphaModelBackwardEKF.m:19-130, Tools/SIAlphaModelBackwardEKFOptControlled.m:19-189
  Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1-290 (+ MatlabCodeGenerator/ twin)
  Tools/SIalpha_Controlled.m, SI_Controlled.m, SEIRP.m, SEIRPSaturatedResource.m, NPICost.m

Arrays use MATLAB shapes: u (n_npi, T), x (T,), S (m, T), P (m, m, T).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

EPS = float(np.finfo(np.float64).eps)


class MatlabError(Exception):
    """Stands in for MATLAB's error('...'); .args[0] is the reference's message."""


@dataclass
class Params:
    """The reference's ``params`` struct (TrainPredictPrescribeNPI.m:202-224)."""
    dt: float = 1.0
    beta: float = 0.0
    gamma: float = 0.0
    sigma: float = 1.0
    b: float = 0.0
    epsilon: float = float("nan")
    s_min: float = 0.0
    i_min: float = 0.0
    alpha_min: float = 0.0
    alpha_max: float = float("inf")
    a: np.ndarray = field(default_factory=lambda: np.zeros(12))
    u_min: np.ndarray = field(default_factory=lambda: np.zeros(12))
    u_max: np.ndarray = field(default_factory=lambda: np.zeros(12))
    w: object = float("nan")   # as the caller set it: scalar NaN, (12,), (1,12), (12,1) or (12,D)
    obs_type: str = "NEWCASES"


def resolve_w(w, a_len: int) -> np.ndarray:
    """``phi(kk)`` is a LINEAR index into ``epsilon*w - gamma*s6*a`` (a is n x 1).

    MATLAB implicit expansion (SIAlphaModelEKFOptControlled.m:49,107): w n x 1 ->
    w(kk); w 1 x n -> the n x n matrix's first column -> w(1) for every kk;
    w n x D -> w(kk, 1); scalar -> that scalar."""
    w = np.asarray(w, dtype=np.float64)
    if w.ndim == 0 or w.size == 1:
        return np.full(a_len, float(w.reshape(-1)[0]))
    if w.ndim == 1:               # a plain vector is taken as a column (n x 1)
        w = w.reshape(-1, 1)
    rows, _ = w.shape
    if rows == 1:                 # 1 x n row: result is n x n, column 1 uses w(1)
        return np.full(a_len, float(w[0, 0]))
    if rows != a_len:
        raise MatlabError("Arrays have incompatible sizes for this operation.")
    return w[:, 0].astype(np.float64).copy()


def mmax(a, b):
    """MATLAB max(a, b): NaN-ignoring."""
    return np.fmax(a, b)


def mmin(a, b):
    return np.fmin(a, b)


def matlab_eps(x: float) -> float:
    x = abs(float(x))
    if x == 0.0 or not math.isfinite(x):
        return 2.0 ** -1074 if x == 0.0 else float("nan")
    e = math.frexp(x)[1] - 1
    return 2.0 ** max(e - 52, -1074)


def matlab_pinv(A: np.ndarray):
    """pinv.m: [U,S,V]=svd(A,'econ'); tol=max(size(A))*eps(norm(s,inf)); r=sum(s>tol);
    X=(V(:,1:r).*(1./s(1:r)).')*U(:,1:r)'.  Returns (X, r)."""
    U, s, Vh = np.linalg.svd(A, full_matrices=False)
    tol = max(A.shape) * matlab_eps(s.max() if s.size else 0.0)
    r = int(np.sum(s > tol))
    if r == 0:
        return np.zeros((A.shape[1], A.shape[0])), 0
    V = Vh.conj().T
    X = (V[:, :r] * (1.0 / s[:r])) @ U[:, :r].conj().T
    return X, r


def mrdivide(B: np.ndarray, A: np.ndarray) -> np.ndarray:
    """B/A for square A: (A'\\B')' through LAPACK dgesv (LU, partial pivoting)."""
    import scipy.linalg as sla
    lu, piv = sla.lu_factor(A.T, check_finite=False)
    return sla.lu_solve((lu, piv), B.T, check_finite=False).T


# --------------------------------------------------------------------------
# model callbacks ("handles")
# --------------------------------------------------------------------------
class _Model:
    def __init__(self, m, flipped, lo_zero, phi_ge=False, obs_clamp=True, obs_fixed=False):
        self.m, self.flipped, self.lo_zero = m, flipped, lo_zero
        self.phi_ge, self.obs_clamp, self.obs_fixed = phi_ge, obs_clamp, obs_fixed

    # StateHardMargins
    def state_hard_margins(self, s, p):
        s = s.copy()
        lo_s = 0.0 if self.lo_zero else p.s_min
        lo_i = 0.0 if self.lo_zero else p.i_min
        s[0] = mmin(1.0, mmax(lo_s, s[0]))
        s[1] = mmin(1.0, mmax(lo_i, s[1]))
        s[2] = mmin(p.alpha_max, mmax(p.alpha_min, s[2]))
        return s

    def obs_hard_margins(self, x, p):
        return mmax(0.0, x) if self.obs_clamp else x

    def _phi(self, s, p):
        return p.epsilon * p._w_eff - p.gamma * s[5] * p.a

    # NlinStateUpdate
    def nlin_state_update(self, u, s, p):
        u = np.array(u, dtype=np.float64)
        sgn = -1.0 if self.flipped else 1.0
        lo_s = 0.0 if self.lo_zero else p.s_min
        lo_i = 0.0 if self.lo_zero else p.i_min
        if self.m == 6:
            phi = self._phi(s, p)
            for kk in range(len(u)):
                if np.isnan(u[kk]):
                    take_min = (phi[kk] >= 0) if self.phi_ge else (phi[kk] > 0)
                    u[kk] = p.u_min[kk] if take_min else p.u_max[kk]
        sn = np.zeros(self.m)
        drive = float((p.gamma * p.a) @ (p.u_max - u))
        sn[0] = mmax(lo_s, mmin(1.0, s[0] - sgn * (p.dt * s[2] * s[0] * s[1])))
        sn[1] = mmax(lo_i, mmin(1.0, s[1] + sgn * (p.dt * (s[2] * s[0] * s[1] - p.beta * s[1]))))
        sn[2] = mmax(p.alpha_min, mmin(p.alpha_max,
                     s[2] + sgn * (p.dt * (-p.gamma * s[2] + p.gamma * p.b + drive))))
        if self.m == 6:
            rho = s[3] - s[4] - (1 - p.epsilon)
            sn[3] = s[3] + sgn * (p.dt * rho * s[2] * s[1])
            sn[4] = s[4] + sgn * (p.dt * (rho * s[2] * s[0] + p.beta * s[4]))
            sn[5] = s[5] + sgn * (p.dt * (rho * s[0] * s[1] + p.gamma * s[5]))
        return u, sn

    # NlinObsUpdate
    def nlin_obs_update(self, s, v_bar, p):
        ot = "NEWCASES" if self.obs_fixed else p.obs_type
        if ot == "NEWCASES":
            return s[0] * s[1] * s[2] + v_bar
        if ot == "TOTALCASES":
            return 1 - s[0] + v_bar
        raise MatlabError("unknown observation type")

    # ObsJacobian
    def obs_jacobian(self, s, p):
        ot = "NEWCASES" if self.obs_fixed else p.obs_type
        C = np.zeros((1, self.m))
        if ot == "NEWCASES":
            C[0, :3] = [s[1] * s[2], s[0] * s[2], s[0] * s[1]]
        elif ot == "TOTALCASES":
            C[0, 0] = -1.0
        else:
            raise MatlabError("unknown observation type")
        return C

    # StateJacobians
    def state_jacobians(self, u, s, p):
        m = self.m
        sg = -1.0 if self.flipped else 1.0
        dt = p.dt
        A = np.zeros((m, m))
        A[0, 0] = 1 - sg * dt * s[2] * s[1]
        A[0, 1] = -sg * dt * s[2] * s[0]
        A[0, 2] = -sg * dt * s[0] * s[1]
        A[1, 0] = sg * dt * s[1] * s[2]
        A[1, 1] = 1 + sg * dt * (s[0] * s[2] - p.beta)
        A[1, 2] = sg * dt * s[0] * s[1]
        A[2, 2] = 1 - sg * dt * p.gamma
        if m == 6:
            phi = self._phi(s, p)
            for kk in range(len(u)):
                if np.isnan(u[kk]):
                    if -1.0 / p.sigma < phi[kk] < 1.0 / p.sigma:
                        A[2, 5] = A[2, 5] - sg * (p.gamma * dt * (p.sigma / 2) * p.a[kk]
                                                  * (p.u_max[kk] - p.u_min[kk]))
            rho = s[3] - s[4] - (1 - p.epsilon)
            A[3, 1] = sg * dt * s[2] * rho
            A[3, 2] = sg * dt * s[1] * rho
            A[3, 3] = 1 + sg * dt * s[1] * s[2]
            A[3, 4] = -sg * dt * s[1] * s[2]
            A[4, 0] = sg * dt * s[2] * rho
            A[4, 2] = sg * dt * s[0] * rho
            A[4, 3] = sg * dt * s[0] * s[2]
            A[4, 4] = 1 - sg * dt * (s[0] * s[2] - p.beta)
            A[5, 0] = sg * dt * s[1] * rho
            A[5, 1] = sg * dt * s[0] * rho
            A[5, 3] = sg * dt * s[0] * s[1]
            A[5, 4] = -sg * dt * s[0] * s[1]
            A[5, 5] = 1 + sg * dt * p.gamma
        return A


MODELS = {
    "SIAlphaModelEKF": _Model(3, False, False),
    "SIAlphaModelEKFOptControlled": _Model(6, False, True),
    "SIAlphaModelBackwardEKF": _Model(3, True, True),
    "SIAlphaModelBackwardEKFOptControlled": _Model(6, True, True),
    "NewCaseEKFEstimatorWithOptimalNPI": _Model(6, False, True, phi_ge=True),
    "NewCaseEKFEstimatorWithOptimalNPI_codegen": _Model(6, False, True, phi_ge=True,
                                                         obs_clamp=False, obs_fixed=True),
}


def _prep_params(p: Params, n_npi: int) -> Params:
    q = Params(**{k: getattr(p, k) for k in p.__dataclass_fields__})
    q.a = np.asarray(p.a, dtype=np.float64).reshape(-1)
    q.u_min = np.asarray(p.u_min, dtype=np.float64).reshape(-1)
    q.u_max = np.asarray(p.u_max, dtype=np.float64).reshape(-1)
    q._w_eff = resolve_w(p.w, n_npi)
    return q


def generic_ekf(u, x, model: _Model, params: Params, s_init, Ps_init, s_final, Ps_final,
                w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order):
    """GenericExtendedKalmanFilter.m.  Returns the 11 outputs (+ pinv ranks)."""
    u = np.asarray(u, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    T = x.shape[0]
    s_init = np.asarray(s_init, dtype=np.float64).reshape(-1)
    m = s_init.shape[0]
    p = _prep_params(params, u.shape[0])
    L = int(inv_monitor_len)
    S_MINUS = np.zeros((m, T)); S_PLUS = np.zeros((m, T))
    P_MINUS = np.zeros((m, m, T)); P_PLUS = np.zeros((m, m, T))
    K_GAIN = np.zeros((m, 1, T)); innovations = np.zeros(T); rho = np.zeros(T)
    InnMean = np.zeros(L); InnCovN = np.zeros(L); InnCov = np.zeros(L)
    sk_minus = s_init.copy()
    Pk_minus = np.array(Ps_init, dtype=np.float64)

    Q_w = np.atleast_2d(np.asarray(Q_w, dtype=np.float64))
    if Q_w.ndim == 2 and Q_w.shape[0] == Q_w.shape[1]:
        Q = np.repeat(Q_w[:, :, None], T, axis=2)                    # :64-65
    elif Q_w.ndim == 3 and Q_w.shape[0] == Q_w.shape[1] and Q_w.shape[2] == T:
        Q = Q_w                                                      # :64 (repmat, first T used)
    elif Q_w.size == T and min(Q_w.shape) == 1:
        Q = Q_w.reshape(1, 1, T)                                     # :67-69
    else:
        raise MatlabError("Process noise covariance noise mismatch")
    R_v = np.atleast_2d(np.asarray(R_v, dtype=np.float64))
    if R_v.shape[0] == R_v.shape[1]:
        R = np.full(T, float(R_v[0, 0])); fixed_R = True             # :79-81
    elif min(R_v.shape) == 1 and R_v.size == T:
        R = R_v.reshape(-1).copy(); fixed_R = False                  # :82-85
    else:
        raise MatlabError("Observation noise covariance noise mismatch")

    u_opt = np.zeros_like(u); u_opt_smooth = np.zeros_like(u)
    I = np.eye(m)
    for k in range(T):
        S_MINUS[:, k] = sk_minus; P_MINUS[:, :, k] = Pk_minus
        if order not in (1, 2):
            raise MatlabError("Undefined order")
        Ck = model.obs_jacobian(sk_minus, p)
        xk_minus = model.obs_hard_margins(model.nlin_obs_update(sk_minus, v_bar, p), p)
        if not np.isnan(x[k]):
            innovations[k] = x[k] - xk_minus
            Kgain = Pk_minus @ Ck.T / (Ck @ Pk_minus @ Ck.T + gamma * R[k])
            IKC = I - Kgain @ Ck
            Pk_plus = (IKC @ Pk_minus @ IKC.T + Kgain * R[k] @ Kgain.T) / gamma
            sk_plus = sk_minus + (Kgain * innovations[k]).reshape(-1)
        else:
            innovations[k] = 0.0
            Kgain = np.zeros((m, 1)); Pk_plus = Pk_minus.copy(); sk_plus = sk_minus.copy()
        Pk_plus = (Pk_plus + Pk_plus.T) / 2.0
        sk_plus = model.state_hard_margins(sk_plus, p)
        u_opt[:, k], sk_minus = model.nlin_state_update(u[:, k], sk_plus, p)
        Ak = model.state_jacobians(u[:, k], sk_plus, p)
        Qk = Q[:, :, k] if Q.shape[0] == m else Q[0, 0, k] * np.eye(m)
        Pk_minus = Ak @ Pk_plus @ Ak.T + Qk
        Pk_minus = (Pk_minus + Pk_minus.T) / 2.0
        sk_minus = model.state_hard_margins(sk_minus, p)
        S_PLUS[:, k] = sk_plus; P_PLUS[:, :, k] = Pk_plus; K_GAIN[:, :, k] = Kgain
        cnt = min(k + 1, L)
        InnMean = np.concatenate(([innovations[k]], InnMean[:L - 1]))
        mu = _seqsum(InnMean) / cnt
        cc = (innovations[k] - mu) * (innovations[k] - mu)
        InnCov = np.concatenate(([cc], InnCov[:L - 1]))
        InnCovN = np.concatenate(([cc / (R[k] + EPS)], InnCovN[:L - 1]))
        rho[k] = _seqsum(InnCovN) / cnt
        if beta != 1 and not np.isnan(x[k]) and fixed_R and k < T - 1:
            R_estim = _seqsum(InnCov) / cnt
            R[k + 1] = beta * R[k] + (1 - beta) * R_estim

    S_SMOOTH = np.zeros((m, T)); P_SMOOTH = np.zeros((m, m, T))
    S_SMOOTH[:, T - 1] = S_PLUS[:, T - 1]; P_SMOOTH[:, :, T - 1] = P_PLUS[:, :, T - 1]
    s_final = np.asarray(s_final, dtype=np.float64).reshape(-1)
    Ps_final = np.asarray(Ps_final, dtype=np.float64)
    fe = ~np.isnan(s_final)
    S_SMOOTH[fe, T - 1] = s_final[fe]
    fc = ~np.isnan(Ps_final)
    P_SMOOTH[:, :, T - 1][fc] = Ps_final[fc]
    ranks = np.full(T, -1, dtype=np.int32)
    for k in range(T - 2, -1, -1):
        Ak = model.state_jacobians(u[:, k], S_PLUS[:, k], p)
        pm = P_MINUS[:, :, k + 1]
        if np.isnan(pm).any() or np.isinf(pm).any():
            J = np.zeros((m, m))
        else:
            X, ranks[k] = matlab_pinv(pm)
            J = (P_PLUS[:, :, k] @ Ak.T) @ X
        S_SMOOTH[:, k] = model.state_hard_margins(
            S_PLUS[:, k] + J @ (S_SMOOTH[:, k + 1] - S_MINUS[:, k + 1]), p)
        Ps = P_PLUS[:, :, k] - J @ (P_MINUS[:, :, k + 1] - P_SMOOTH[:, :, k + 1]) @ J.T
        P_SMOOTH[:, :, k] = (Ps + Ps.T) / 2.0
        u_opt_smooth[:, k], _ = model.nlin_state_update(u[:, k], S_SMOOTH[:, k], p)
    return (u_opt, u_opt_smooth, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH,
            K_GAIN, innovations, rho, ranks)


def _seqsum(v):
    acc = float(v[0])
    for j in range(1, len(v)):
        acc = acc + float(v[j])
    return acc


def backward_wrapper(u, x, model, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar,
                     Q_w, R_v, beta, gamma, inv_monitor_len, order):
    """SIAlphaModelBackwardEKF.m:19-40 (same for the OptControlled twin)."""
    u = np.asarray(u, dtype=np.float64); x = np.asarray(x, dtype=np.float64).reshape(-1)
    out = generic_ekf(u[:, ::-1], x[::-1], model, params, s_final, Ps_final, s_init, Ps_init,
                      w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order)
    # every output is reversed along its time axis EXCEPT rho: GenericExtendedKalmanFilter.m:233 has squeezed it to
    # T x 1, and `rho_flipped(:, :, end:-1:1)` (SIAlphaModelBackwardEKF.m:40) indexes a third dimension of size 1 --
    # `end` is 1 there, so MATLAB hands the column back un-reversed (filter-step order)
    names = ("u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH", "K_GAIN",
             "innovations", "rho", "ranks")
    assert len(out) == len(names)
    return tuple(o if n == "rho" else np.flip(o, axis=-1) for n, o in zip(names, out))


def newcase_ekf(u, x, model: _Model, params: Params, s_init, Ps_init, s_final, Ps_final,
                w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order):
    """NewCaseEKFEstimatorWithOptimalNPI.m:1-143.  Returns the 10 outputs in Tools/ order."""
    u = np.asarray(u, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    T = x.shape[0]
    s_init = np.asarray(s_init, dtype=np.float64).reshape(-1)
    m = s_init.shape[0]
    p = _prep_params(params, u.shape[0])
    L = int(inv_monitor_len)
    S_MINUS = np.zeros((m, T)); S_PLUS = np.zeros((m, T))
    P_MINUS = np.zeros((m, m, T)); P_PLUS = np.zeros((m, m, T))
    K_GAIN = np.zeros((m, 1, T)); innovations = np.zeros(T); rho = np.zeros(T)
    InnMean = np.zeros(L); InnCovN = np.zeros(L); InnCov = np.zeros(L)
    sk_minus = s_init.copy(); Pk_minus = np.array(Ps_init, dtype=np.float64)
    Q = np.asarray(Q_w, dtype=np.float64); R = float(np.asarray(R_v).reshape(-1)[0])
    u_opt = np.zeros_like(u)
    I = np.eye(m)
    for k in range(T):
        S_MINUS[:, k] = sk_minus; P_MINUS[:, :, k] = Pk_minus
        if order not in (1, 2):
            raise MatlabError("Undefined order")
        Ck = model.obs_jacobian(sk_minus, p)
        xk_minus = model.obs_hard_margins(model.nlin_obs_update(sk_minus, v_bar, p), p)
        if not np.isnan(x[k]):
            innovations[k] = x[k] - xk_minus
            Kgain = Pk_minus @ Ck.T / (Ck @ Pk_minus @ Ck.T + gamma * R)
            Pk_plus = (I - Kgain @ Ck) @ Pk_minus / gamma
            sk_plus = sk_minus + (Kgain * innovations[k]).reshape(-1)
        else:
            innovations[k] = 0.0
            Kgain = np.zeros((m, 1)); Pk_plus = Pk_minus.copy(); sk_plus = sk_minus.copy()
        sk_plus = model.state_hard_margins(sk_plus, p)
        u_opt[:, k], sk_minus = model.nlin_state_update(u[:, k], sk_plus, p)
        Ak = model.state_jacobians(u[:, k], sk_plus, p)
        Pk_minus = Ak @ Pk_plus @ Ak.T + Q
        sk_minus = model.state_hard_margins(sk_minus, p)
        S_PLUS[:, k] = sk_plus; P_PLUS[:, :, k] = Pk_plus; K_GAIN[:, :, k] = Kgain
        cnt = min(k + 1, L)
        InnMean = np.concatenate(([innovations[k]], InnMean[:L - 1]))
        mu = _seqsum(InnMean) / cnt
        cc = (innovations[k] - mu) * (innovations[k] - mu)
        InnCov = np.concatenate(([cc], InnCov[:L - 1]))
        with np.errstate(divide="ignore", invalid="ignore"):
            InnCovN = np.concatenate(([np.float64(cc) / np.float64(R)], InnCovN[:L - 1]))
        rho[k] = _seqsum(InnCovN) / cnt
        if beta != 1 and not np.isnan(x[k]):
            R = beta * R + (1 - beta) * _seqsum(InnCov) / cnt
    S_SMOOTH = np.zeros((m, T)); P_SMOOTH = np.zeros((m, m, T))
    S_SMOOTH[:, T - 1] = S_PLUS[:, T - 1]; P_SMOOTH[:, :, T - 1] = P_PLUS[:, :, T - 1]
    s_final = np.asarray(s_final, dtype=np.float64).reshape(-1)
    Ps_final = np.asarray(Ps_final, dtype=np.float64)
    fe = ~np.isnan(s_final)
    S_SMOOTH[fe, T - 1] = s_final[fe]
    rows, cols = np.nonzero(~np.isnan(Ps_final))
    if rows.size:                                   # P_SMOOTH(row, col, T) = Ps_final(row, col)
        rr, cc_ = np.unique(rows), np.unique(cols)
        P_SMOOTH[np.ix_(rr, cc_, [T - 1])] = Ps_final[np.ix_(rr, cc_)][:, :, None]
    for k in range(T - 2, -1, -1):
        Ak = model.state_jacobians(u[:, k], S_PLUS[:, k], p)
        with np.errstate(all="ignore"):
            J = mrdivide(P_PLUS[:, :, k] @ Ak.T, P_MINUS[:, :, k + 1])
        S_SMOOTH[:, k] = model.state_hard_margins(
            S_PLUS[:, k] + J @ (S_SMOOTH[:, k + 1] - S_MINUS[:, k + 1]), p)
        P_SMOOTH[:, :, k] = P_PLUS[:, :, k] - J @ (P_MINUS[:, :, k + 1] - P_SMOOTH[:, :, k + 1]) @ J.T
    return (u_opt, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH, K_GAIN, innovations, rho)


def run_model(name, *args):
    """Dispatch on the reference's function name."""
    model = MODELS[name]
    if name.startswith("NewCase"):
        return newcase_ekf(args[0], args[1], model, *args[2:])
    if model.flipped:
        return backward_wrapper(args[0], args[1], model, *args[2:])
    return generic_ekf(args[0], args[1], model, *args[2:])


# --------------------------------------------------------------------------
# forward simulators and cost
# --------------------------------------------------------------------------
def sialpha_controlled(u, s0, i0, alpha0, u_max, alpha_min, alpha_max, gamma, a, b, beta,
                       s_noise_std, i_noise_std, alpha_noise_std, K, dt, z=None):
    """SIalpha_Controlled.m:1-32; z (K,3) replaces the three randn calls per step."""
    u = np.asarray(u, dtype=np.float64); a = np.asarray(a, dtype=np.float64).reshape(-1)
    u_max = np.asarray(u_max, dtype=np.float64).reshape(-1)
    s = np.zeros(K + 1); i = np.zeros(K + 1); al = np.zeros(K + 1)
    s[0], i[0], al[0] = s0, i0, alpha0
    z = np.zeros((K, 3)) if z is None else np.asarray(z, dtype=np.float64).reshape(K, 3)
    for t in range(K):
        s[t + 1] = mmax(0.0, mmin(1.0, s[t] - dt * (al[t] * s[t] * i[t] + z[t, 0] * s_noise_std)))
        i[t + 1] = mmax(0.0, mmin(1.0, i[t] + dt * (al[t] * s[t] * i[t] - beta * i[t] + z[t, 1] * i_noise_std)))
        al[t + 1] = mmax(alpha_min, mmin(alpha_max, al[t] + dt * (
            -gamma * al[t] + gamma * b + float((gamma * a) @ (u_max - u[:, t])) + z[t, 2] * alpha_noise_std)))
    return s[1:], i[1:], al[1:]


def si_controlled(alpha, beta, s0, i0, K, dt):
    s = np.zeros(K); i = np.zeros(K)
    s[0], i[0] = s0, i0
    for t in range(K - 1):
        s[t + 1] = mmax(0.0, mmin(1.0, s[t] - dt * alpha[t] * s[t] * i[t]))
        i[t + 1] = mmax(0.0, mmin(1.0, i[t] + dt * (alpha[t] * s[t] * i[t] - beta * i[t])))
    return s, i


def sir(alpha, beta, gamma, s0, i0, r0, K, dt):
    """testScripts/testSIR01.m:28-36 (3-compartment SIR with return flow r -> s, forward Euler, no clamps)."""
    s = np.zeros(K); i = np.zeros(K); r = np.zeros(K)
    s[0], i[0], r[0] = s0, i0, r0
    for t in range(K - 1):
        s[t + 1] = (-alpha * s[t] * i[t] + gamma * r[t]) * dt + s[t]
        i[t + 1] = (alpha * s[t] * i[t] - beta * i[t]) * dt + i[t]
        r[t + 1] = (beta * i[t] - gamma * r[t]) * dt + r[t]
    return s, i, r


def seirp(alpha_e, alpha_i, kappa, rho, beta, mu, gamma, s0, e0, i0, r0, p0, T, dt):
    K = int(round(T / dt))
    s = np.zeros(K); e = np.zeros(K); i = np.zeros(K); r = np.zeros(K); p = np.zeros(K)
    s[0], e[0], i[0], r[0], p[0] = s0, e0, i0, r0, p0
    for t in range(K - 1):
        s[t + 1] = (-alpha_e[t] * s[t] * e[t] - alpha_i[t] * s[t] * i[t] + gamma[t] * r[t]) * dt + s[t]
        e[t + 1] = (alpha_e[t] * s[t] * e[t] + alpha_i[t] * s[t] * i[t] - kappa[t] * e[t] - rho[t] * e[t]) * dt + e[t]
        i[t + 1] = (kappa[t] * e[t] - beta[t] * i[t] - mu[t] * i[t]) * dt + i[t]
        r[t + 1] = (beta[t] * i[t] + rho[t] * e[t] - gamma[t] * r[t]) * dt + r[t]
        p[t + 1] = (mu[t] * i[t]) * dt + p[t]
    return s, e, i, r, p


def seirp_saturated(alpha_e, alpha_i, kappa, rho, gamma, s0, e0, i0, r0, p0, T, dt,
                    beta_0, beta_s, mu_0, mu_s, sigma, i_0):
    K = int(round(T / dt))
    s = np.zeros(K); e = np.zeros(K); i = np.zeros(K); r = np.zeros(K); p = np.zeros(K)
    s[0], e[0], i[0], r[0], p[0] = s0, e0, i0, r0, p0
    for t in range(K - 1):
        h = (math.tanh((i[t] - i_0) / sigma) + 1) / 2
        beta = (beta_s - beta_0) * h + beta_0
        mu = (mu_s - mu_0) * h + mu_0
        s[t + 1] = (-alpha_e[t] * s[t] * e[t] - alpha_i[t] * s[t] * i[t] + gamma[t] * r[t]) * dt + s[t]
        e[t + 1] = (alpha_e[t] * s[t] * e[t] + alpha_i[t] * s[t] * i[t] - kappa[t] * e[t] - rho[t] * e[t]) * dt + e[t]
        i[t + 1] = (kappa[t] * e[t] - beta * i[t] - mu * i[t]) * dt + i[t]
        r[t + 1] = (beta * i[t] + rho[t] * e[t] - gamma[t] * r[t]) * dt + r[t]
        p[t + 1] = (mu * i[t]) * dt + p[t]
    return s, e, i, r, p


def npi_cost(newcases, inputs, weights):
    J0 = float(np.mean(np.asarray(newcases, dtype=np.float64)))
    wi = np.asarray(weights, dtype=np.float64) * np.asarray(inputs, dtype=np.float64)
    return J0, float(np.mean(wi.reshape(-1, order="F")))


# ---------------------------------------------------------------------------------------------------
# Tools/Rt_ExpFitEKF.m -- second, independent reading (NumPy/LAPACK) used to cross-check the C oracle
# ---------------------------------------------------------------------------------------------------
def rt_expfit_ekf(x, s_init, params, w_bar, v_bar, Ps_init, Q_w, R_v, beta, gamma, inv_monitor_len, order):
    """[S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho] = Rt_ExpFitEKF(...)
    for a scalar observation series x (1 x T).  Tools/Rt_ExpFitEKF.m:1-130."""
    if order not in (1, 2):
        raise ValueError("Undefined order")                       # :46, :77
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    T, m, Lw = x.shape[0], 2, int(inv_monitor_len)
    ts, alpha, sigma = (float(v) for v in params)
    w_bar = np.asarray(w_bar, dtype=np.float64).reshape(-1)
    Q = np.asarray(Q_w, dtype=np.float64); R = float(R_v)
    S_MINUS = np.zeros((m, T)); S_PLUS = np.zeros((m, T)); P_MINUS = np.zeros((m, m, T)); P_PLUS = np.zeros((m, m, T))
    K_GAIN = np.zeros((m, 1, T)); innovations = np.zeros((1, T)); rho = np.zeros(T)
    win_mean = np.zeros(Lw); win_cov = np.zeros(Lw); win_covn = np.zeros(Lw)
    sk_minus = np.asarray(s_init, dtype=np.float64).reshape(-1).copy(); Pk_minus = np.asarray(Ps_init, dtype=np.float64).copy()
    C = np.array([[1.0, 0.0]]); D = 1.0

    def jac(s):                                                   # :143-160
        tnh = np.tanh((alpha * s[1] + w_bar[1]) / sigma)
        e = np.exp(ts * s[1])
        A = np.array([[e, ts * s[0] * e], [0.0, alpha * (1 - tnh ** 2)]])
        B = np.array([[1.0, 0.0], [0.0, 1 - tnh ** 2]])
        return A, B, tnh, e

    def hess(s, Pk, Qk):                                          # :163-199
        _, _, tnh, e = jac(s)
        Fs = [np.array([[0.0, ts * e], [ts * e, ts ** 2 * s[0] * e]]),
              np.array([[0.0, 0.0], [0.0, -2 * alpha ** 2 / sigma * tnh * (1 - tnh ** 2)]])]
        Fw = [np.zeros((2, 2)), np.array([[0.0, 0.0], [0.0, -2 / sigma * tnh * (1 - tnh ** 2)]])]
        fs = np.array([np.trace(Pk @ Fs[i]) / 2 for i in range(2)])
        Cs = np.array([[np.trace(Pk @ Fs[i] @ Pk @ Fs[j]) / 2 for j in range(2)] for i in range(2)])
        fw = np.array([np.trace(Qk @ Fw[i]) / 2 for i in range(2)])
        Cw = np.array([[np.trace(Qk @ Fw[i] @ Qk @ Fw[j]) / 2 for j in range(2)] for i in range(2)])
        return fs, Cs, fw, Cw

    for k in range(T):
        S_MINUS[:, k] = sk_minus; P_MINUS[:, :, k] = Pk_minus
        xk_minus = sk_minus[0] + v_bar                            # obs Hessian terms are identically zero (:202-227)
        if not np.isnan(x[k]):
            innov = x[k] - xk_minus
            Kg = Pk_minus @ C.T / (C @ Pk_minus @ C.T + gamma * (D * R * D))
            Pk_plus = (np.eye(m) - Kg @ C) @ Pk_minus / gamma
            sk_plus = sk_minus + (Kg * innov).reshape(-1)
        else:
            innov = 0.0; Kg = np.zeros((m, 1)); Pk_plus = Pk_minus.copy(); sk_plus = sk_minus.copy()
        A, B, tnh, e = jac(sk_plus)
        if order == 2:
            fs, Fsp, fw, Fwp = hess(sk_plus, Pk_plus, Q)
        else:
            fs = fw = np.zeros(m); Fsp = Fwp = np.zeros((m, m))
        sk_minus = np.array([sk_plus[0] * e + w_bar[0], sigma * tnh]) + fs + fw
        Pk_minus = A @ Pk_plus @ A.T + B @ Q @ B.T + Fsp + Fwp
        S_PLUS[:, k] = sk_plus; P_PLUS[:, :, k] = Pk_plus; K_GAIN[:, :, k] = Kg; innovations[0, k] = innov
        cnt = min(k + 1, Lw)
        win_mean = np.concatenate(([innov], win_mean[:-1]))
        mu = win_mean.sum() / cnt
        cc = (innov - mu) ** 2
        win_cov = np.concatenate(([cc], win_cov[:-1])); win_covn = np.concatenate(([cc / R], win_covn[:-1]))
        rho[k] = win_covn.sum() / cnt
        if beta != 1 and not np.isnan(x[k]):
            R = beta * R + (1 - beta) * win_cov.sum() / cnt
    S_SMOOTH = np.zeros_like(S_PLUS); P_SMOOTH = np.zeros_like(P_PLUS)
    S_SMOOTH[:, -1] = S_PLUS[:, -1]; P_SMOOTH[:, :, -1] = P_PLUS[:, :, -1]
    for k in range(T - 2, -1, -1):
        A = jac(S_PLUS[:, k])[0]
        J = np.linalg.solve(P_MINUS[:, :, k + 1].T, (P_PLUS[:, :, k] @ A.T).T).T       # mrdivide
        S_SMOOTH[:, k] = S_PLUS[:, k] + J @ (S_SMOOTH[:, k + 1] - S_MINUS[:, k + 1])
        P_SMOOTH[:, :, k] = P_PLUS[:, :, k] - J @ (P_MINUS[:, :, k + 1] - P_SMOOTH[:, :, k + 1]) @ J.T
    return S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho


# ---------------------------------------------------------------------------------------------------
# Per-region preprocessing (TrainPredictPrescribeNPI.m:142-198,201-202,240) -- independent reading on
# scipy.signal's lfilter / filtfilt, which implement the same published definitions as MATLAB's
# ---------------------------------------------------------------------------------------------------
def preprocess_region(cases, deaths, N_population, W=7, min_cases=1.0, first_num_days=7):
    from scipy.signal import filtfilt, lfilter

    def refine(cum):
        cum = np.asarray(cum, dtype=np.float64)
        d = np.diff(np.concatenate(([cum[0]], cum)))
        d[d < 0] = 0
        r = d.copy()
        if np.isnan(d[-1]):
            ok = np.flatnonzero(~np.isnan(d))
            if ok.size:
                r[-1] = r[ok[-1]]
        r[np.isnan(r)] = 0
        return r

    ref = refine(cases)
    sm = lfilter(np.ones(W), W, ref)
    W2 = int(np.floor(W / 2 + 0.5))
    # a single tap is the identity (SciPy's lfilter_zi rejects it; MATLAB pads one sample and returns x)
    zl = ref.copy() if W2 <= 1 else filtfilt(np.ones(W2), W2, ref, padtype="odd", padlen=3 * (W2 - 1))
    cs = np.cumsum(sm)
    out = {"new_refined": ref, "new_smoothed": sm, "zero_lag": zl, "x_new": sm / N_population,
           "x_total": cs / N_population, "R_v": 0.1 * ((zl - ref) / N_population) ** 2}
    first = np.flatnonzero(sm > 0)[:first_num_days]
    out["I0"] = max(min_cases, float(np.mean(sm[first]))) if first.size else float(min_cases)
    if deaths is not None:
        with np.errstate(all="ignore"):
            fr = np.cumsum(lfilter(np.ones(W), W, refine(deaths))) / cs
        fr[np.isnan(fr)] = 0
        out["fatality"] = fr
    return out


def npi_fill(ip):
    ip = np.array(ip, dtype=np.float64)
    for j in range(ip.shape[1]):
        for i in range(1, ip.shape[0]):
            if np.isnan(ip[i, j]) and not np.isnan(ip[i - 1, j]):
                ip[i, j] = ip[i - 1, j]
    ip[np.isnan(ip)] = 0
    return ip
