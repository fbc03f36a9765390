"""referee_mp.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE ("parity unpinned").

A referee for the outputs no two fp64 implementations can agree on: the reference's filter / smoother evaluated in
multi-precision arithmetic (mpmath, 160 significant digits by default), so that what it returns is -- to ~100 digits
-- what the reference's FORMULAS give on the fp64 inputs, free of the rounding that separates MATLAB's LAPACK, NumPy's
LAPACK, this repo's C oracle and its HIP kernels from one another.  The smoother inverts covariances with condition
numbers 1e30-1e60 (`pinv`, GenericExtendedKalmanFilter.m:215); there the fp64 readings differ from each other by
1e-7 ... O(1), and only an evaluation that does not round can say which of them is nearer the exact result and by how
much ANY fp64 evaluation must be expected to miss it.

Written from the .m text alone (third reading; shares no code with ekf_oracle.c or ekf_numpy.py except the Params
dataclass it accepts):
  Tools/GenericExtendedKalmanFilter.m:60-233        forward loop :98-186, smoother :189-230
  Tools/SIAlphaModelEKF.m:27-89                     3-state callbacks
  Tools/SIAlphaModelEKFOptControlled.m:27-148       6-state callbacks (bang-bang control, slope term A(3,6))
  Tools/SIAlphaModelBackwardEKF.m:19-40,60-97       time-flipped wrappers (every dt term changes sign; rho un-reversed)
  Tools/SIAlphaModelBackwardEKFOptControlled.m:19-40,60-156
  MATLAB built-ins: pinv (svd, tol = max(size(A)) * eps(norm(A)), rank = #(s > tol)), min / max ignoring NaN, eps(x).

What "exact" means here.  Every +, -, *, / and the SVD run at `dps` digits; the DISCRETE decisions are the
reference's: isnan tests, the clamps, the strict `phi > 0` of the bang-bang rule, `|phi| < 1/sigma`, the truncation
`s > tol`.  `eps(norm)` is the spacing of doubles at the (exact) largest singular value -- 2^(floor(log2 s) - 52) --
which is what MATLAB computes from the fp64 value except when s sits within one ulp of a power of two.  Where a
singular value lies within a factor `near` (default 10) of the cut-off the truncation rank is genuinely ambiguous:
any fp64 SVD may land on either side.  Those steps are listed in the result (`near_cutoff`).

Only tests/ and the report scripts under oracle/ and profiles/ import this module.
"""
from __future__ import annotations

import math

import numpy as np
from mpmath import mp, mpf

DEFAULT_DPS = 160

MODELS = {  # name -> (m, flipped, lower clamp of s and i is 0 instead of s_min / i_min)
    "SIAlphaModelEKF": (3, False, False),
    "SIAlphaModelEKFOptControlled": (6, False, True),
    "SIAlphaModelBackwardEKF": (3, True, True),
    "SIAlphaModelBackwardEKFOptControlled": (6, True, True),
    # the older fused filter (Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1-290) and its MATLAB-Coder twin
    # (MatlabCodeGenerator/: observation always NEWCASES, ObsHardMargins the identity)
    "NewCaseEKFEstimatorWithOptimalNPI": (6, False, True),
    "NewCaseEKFEstimatorWithOptimalNPI_codegen": (6, False, True),
}


def _isnan(v):
    return isinstance(v, float) and math.isnan(v) or (hasattr(v, "_mpf_") and mp.isnan(v))


def _F(v):
    """fp64 -> mpf exactly (NaN and Inf kept)."""
    return mpf(float(v))


def _mmax(a, b):
    """MATLAB max(a, b): a NaN operand is ignored."""
    if mp.isnan(a):
        return b
    if mp.isnan(b):
        return a
    return a if a >= b else b


def _mmin(a, b):
    if mp.isnan(a):
        return b
    if mp.isnan(b):
        return a
    return a if a <= b else b


def _matmul(A, B):
    n, k, m = len(A), len(B), len(B[0])
    return [[mp.fsum(A[i][l] * B[l][j] for l in range(k)) for j in range(m)] for i in range(n)]


def _T(A):
    return [list(r) for r in zip(*A)]


def _sym(A):
    n = len(A)
    return [[(A[i][j] + A[j][i]) / 2 for j in range(n)] for i in range(n)]


def _eps_of(x):
    """MATLAB eps(x) for finite x > 0: the spacing of doubles at x."""
    if x == 0:
        return mpf(2) ** -1074
    e = int(mp.floor(mp.log(x, 2)))
    while mpf(2) ** e > x:          # guard the logarithm's last digit at exact powers of two
        e -= 1
    while mpf(2) ** (e + 1) <= x:
        e += 1
    return mpf(2) ** max(e - 52, -1074)


def pinv_exact(A, near=10):
    """MATLAB pinv of a square matrix (lists of mpf): (X, rank, singular values, tol, closest ratio to the cut-off)."""
    n = len(A)
    M = mp.matrix(A)
    U, S, V = mp.svd_r(M)                      # M = U diag(S) V, S descending
    s = [S[i] for i in range(n)]
    tol = n * _eps_of(s[0])
    r = sum(1 for v in s if v > tol)
    X = [[mpf(0)] * n for _ in range(n)]
    for k in range(r):
        inv = 1 / s[k]
        for i in range(n):
            vi = V[k, i] * inv
            for j in range(n):
                X[i][j] += vi * U[j, k]
    ratio = min((max(v / tol, tol / v) if v > 0 else mp.inf) for v in s)
    return X, r, s, tol, ratio


class _Callbacks:
    """The `handles` of one model (SIAlphaModelEKF.m:12-20 and twins), on mpf scalars."""

    def __init__(self, name, p, n_npi):
        self.m, self.flipped, self.lo_zero = MODELS[name]
        self.sg = mpf(-1) if self.flipped else mpf(1)
        f = _F
        self.dt, self.beta, self.gamma, self.sigma, self.b = f(p.dt), f(p.beta), f(p.gamma), f(p.sigma), f(p.b)
        self.eps_cost = f(p.epsilon)
        self.lo_s = mpf(0) if self.lo_zero else f(p.s_min)
        self.lo_i = mpf(0) if self.lo_zero else f(p.i_min)
        self.a_lo, self.a_hi = f(p.alpha_min), f(p.alpha_max)
        self.a = [f(v) for v in np.asarray(p.a, dtype=np.float64).reshape(-1)[:n_npi]]
        self.u_min = [f(v) for v in np.asarray(p.u_min, dtype=np.float64).reshape(-1)[:n_npi]]
        self.u_max = [f(v) for v in np.asarray(p.u_max, dtype=np.float64).reshape(-1)[:n_npi]]
        # phi(kk) is a LINEAR index into epsilon*w - gamma*s6*a: w n x 1 -> w(kk); 1 x n -> w(1); n x D -> w(kk,1)
        w = np.asarray(p.w, dtype=np.float64)
        if w.ndim == 0 or w.size == 1:
            we = [float(w.reshape(-1)[0])] * n_npi
        else:
            w2 = w.reshape(-1, 1) if w.ndim == 1 else w
            we = [float(w2[0, 0])] * n_npi if w2.shape[0] == 1 else [float(v) for v in w2[:n_npi, 0]]
        self.w = [f(v) for v in we]
        self.newcase = name.startswith("NewCase")
        self.codegen = name.endswith("_codegen")
        self.obs_type = "NEWCASES" if self.codegen else p.obs_type
        self.n_npi = n_npi

    def margins(self, s):                                     # StateHardMargins: min(hi, max(lo, .))
        s = list(s)
        s[0] = _mmin(mpf(1), _mmax(self.lo_s, s[0]))
        s[1] = _mmin(mpf(1), _mmax(self.lo_i, s[1]))
        s[2] = _mmin(self.a_hi, _mmax(self.a_lo, s[2]))
        return s

    def phi(self, s, kk):
        return self.eps_cost * self.w[kk] - self.gamma * s[5] * self.a[kk]

    def resolve_u(self, u, s):
        u = list(u)
        if self.m == 6:
            for kk in range(self.n_npi):
                if mp.isnan(u[kk]):
                    ph = self.phi(s, kk)       # strict > in SIAlphaModelEKFOptControlled.m:52, >= in NewCase...m:175
                    u[kk] = self.u_min[kk] if (ph >= 0 if self.newcase else ph > 0) else self.u_max[kk]
        return u

    def state_update(self, u, s):                             # NlinStateUpdate: max(lo, min(hi, .))
        u = self.resolve_u(u, s)
        sg, dt = self.sg, self.dt
        drive = mp.fsum(self.gamma * self.a[k] * (self.u_max[k] - u[k]) for k in range(self.n_npi))
        sn = [mpf(0)] * self.m
        sn[0] = _mmax(self.lo_s, _mmin(mpf(1), s[0] - sg * (dt * s[2] * s[0] * s[1])))
        sn[1] = _mmax(self.lo_i, _mmin(mpf(1), s[1] + sg * (dt * (s[2] * s[0] * s[1] - self.beta * s[1]))))
        sn[2] = _mmax(self.a_lo, _mmin(self.a_hi, s[2] + sg * (dt * (-self.gamma * s[2] + self.gamma * self.b + drive))))
        if self.m == 6:
            rho = s[3] - s[4] - (1 - self.eps_cost)
            sn[3] = s[3] + sg * (dt * rho * s[2] * s[1])
            sn[4] = s[4] + sg * (dt * (rho * s[2] * s[0] + self.beta * s[4]))
            sn[5] = s[5] + sg * (dt * (rho * s[0] * s[1] + self.gamma * s[5]))
        return u, sn

    def h(self, s, v_bar):                                    # NlinObsUpdate
        if self.obs_type == "NEWCASES":
            return s[0] * s[1] * s[2] + v_bar
        if self.obs_type == "TOTALCASES":
            return 1 - s[0] + v_bar
        raise ValueError("unknown observation type")

    def C(self, s):                                           # ObsJacobian
        c = [mpf(0)] * self.m
        if self.obs_type == "NEWCASES":
            c[0], c[1], c[2] = s[1] * s[2], s[0] * s[2], s[0] * s[1]
        elif self.obs_type == "TOTALCASES":
            c[0] = mpf(-1)
        else:
            raise ValueError("unknown observation type")
        return c

    def A(self, u, s):                                        # StateJacobians (u = the ORIGINAL column, NaNs included)
        m, sg, dt = self.m, self.sg, self.dt
        A = [[mpf(0)] * m for _ in range(m)]
        A[0][0] = 1 - sg * dt * s[2] * s[1]
        A[0][1] = -sg * dt * s[2] * s[0]
        A[0][2] = -sg * dt * s[0] * s[1]
        A[1][0] = sg * dt * s[1] * s[2]
        A[1][1] = 1 + sg * dt * (s[0] * s[2] - self.beta)
        A[1][2] = sg * dt * s[0] * s[1]
        A[2][2] = 1 - sg * dt * self.gamma
        if m == 6:
            for kk in range(self.n_npi):
                if mp.isnan(u[kk]):
                    ph = self.phi(s, kk)
                    if -1 / self.sigma < ph < 1 / self.sigma:
                        A[2][5] = A[2][5] - sg * (self.gamma * dt * (self.sigma / 2) * self.a[kk] * (self.u_max[kk] - self.u_min[kk]))
            rho = s[3] - s[4] - (1 - self.eps_cost)
            A[3][1] = sg * dt * s[2] * rho
            A[3][2] = sg * dt * s[1] * rho
            A[3][3] = 1 + sg * dt * s[1] * s[2]
            A[3][4] = -sg * dt * s[1] * s[2]
            A[4][0] = sg * dt * s[2] * rho
            A[4][2] = sg * dt * s[0] * rho
            A[4][3] = sg * dt * s[0] * s[2]
            A[4][4] = 1 - sg * dt * (s[0] * s[2] - self.beta)
            A[5][0] = sg * dt * s[1] * rho
            A[5][1] = sg * dt * s[0] * rho
            A[5][3] = sg * dt * s[0] * s[1]
            A[5][4] = -sg * dt * s[0] * s[1]
            A[5][5] = 1 + sg * dt * self.gamma
        return A


def _generic(name, u, x, params, s_init, Ps_init, s_final, Ps_final, v_bar, Q_w, R_v, beta, gamma, L, near):
    """GenericExtendedKalmanFilter.m with the callbacks of `name`; u [n_npi][T], x [T] as mpf (NaN allowed)."""
    n_npi, T = len(u), len(x)
    cb = _Callbacks(name, params, n_npi)
    m = cb.m
    Q_w = np.atleast_2d(np.asarray(Q_w, dtype=np.float64))
    if Q_w.ndim == 3:
        Qs = [[[_F(Q_w[i, j, k]) for j in range(m)] for i in range(m)] for k in range(T)]
    elif Q_w.shape == (m, m):
        Qk = [[_F(Q_w[i, j]) for j in range(m)] for i in range(m)]
        Qs = [Qk] * T
    else:
        raise ValueError("Process noise covariance noise mismatch")
    R_v = np.atleast_2d(np.asarray(R_v, dtype=np.float64))
    if R_v.shape[0] == R_v.shape[1]:
        R = [_F(R_v[0, 0])] * T; fixed_R = True
    elif min(R_v.shape) == 1 and R_v.size == T:
        R = [_F(v) for v in R_v.reshape(-1)]; fixed_R = False
    else:
        raise ValueError("Observation noise covariance noise mismatch")
    beta_m, gam = _F(beta), _F(gamma)
    v_bar = _F(v_bar)
    EPS = mpf(2) ** -52
    s_minus = [_F(v) for v in np.asarray(s_init, dtype=np.float64).reshape(-1)]
    P_minus = [[_F(Ps_init[i][j]) for j in range(m)] for i in range(m)]
    Sm, Sp, Pm, Pp, Kg, innov, rho, u_opt = [], [], [], [], [], [], [], []
    w_mean, w_cov, w_covn = [mpf(0)] * L, [mpf(0)] * L, [mpf(0)] * L
    I = [[mpf(1) if i == j else mpf(0) for j in range(m)] for i in range(m)]
    for k in range(T):
        Sm.append(list(s_minus)); Pm.append([list(r) for r in P_minus])
        C = cb.C(s_minus)
        xk = cb.h(s_minus, v_bar)
        if not cb.codegen:
            xk = _mmax(mpf(0), xk)                            # ObsHardMargins (SIAlphaModelEKF.m:34-36)
        if not mp.isnan(x[k]):
            inn = x[k] - xk
            PCt = [mp.fsum(P_minus[i][j] * C[j] for j in range(m)) for i in range(m)]
            den = mp.fsum(C[i] * PCt[i] for i in range(m)) + gam * R[k]
            K = [v / den for v in PCt]
            IKC = [[I[i][j] - K[i] * C[j] for j in range(m)] for i in range(m)]
            J1 = _matmul(_matmul(IKC, P_minus), _T(IKC))
            P_plus = [[(J1[i][j] + K[i] * R[k] * K[j]) / gam for j in range(m)] for i in range(m)]
            s_plus = [s_minus[i] + K[i] * inn for i in range(m)]
        else:
            inn = mpf(0); K = [mpf(0)] * m
            P_plus = [list(r) for r in P_minus]; s_plus = list(s_minus)
        P_plus = _sym(P_plus)
        s_plus = cb.margins(s_plus)
        uk = [u[j][k] for j in range(n_npi)]
        uo, s_minus = cb.state_update(uk, s_plus)
        A = cb.A(uk, s_plus)
        APA = _matmul(_matmul(A, P_plus), _T(A))
        P_minus = _sym([[APA[i][j] + Qs[k][i][j] for j in range(m)] for i in range(m)])
        s_minus = cb.margins(s_minus)
        Sp.append(s_plus); Pp.append(P_plus); Kg.append(K); innov.append(inn); u_opt.append(uo)
        cnt = min(k + 1, L)
        w_mean = [inn] + w_mean[:L - 1]
        mu = mp.fsum(w_mean) / cnt
        cc = (inn - mu) * (inn - mu)
        w_cov = [cc] + w_cov[:L - 1]
        w_covn = [cc / (R[k] + EPS)] + w_covn[:L - 1]
        rho.append(mp.fsum(w_covn) / cnt)
        if beta_m != 1 and not mp.isnan(x[k]) and fixed_R and k < T - 1:
            R = list(R)
            R[k + 1] = beta_m * R[k] + (1 - beta_m) * (mp.fsum(w_cov) / cnt)
    # smoother
    Ss = [None] * T; Ps = [None] * T; u_s = [[mpf(0)] * n_npi for _ in range(T)]
    Ss[T - 1] = list(Sp[T - 1]); Ps[T - 1] = [list(r) for r in Pp[T - 1]]
    sf = np.asarray(s_final, dtype=np.float64).reshape(-1)
    Pf = np.asarray(Ps_final, dtype=np.float64)
    for i in range(m):
        if not np.isnan(sf[i]):
            Ss[T - 1][i] = _F(sf[i])
        for j in range(m):
            if not np.isnan(Pf[i, j]):
                Ps[T - 1][i][j] = _F(Pf[i, j])
    ranks = [-1] * T; near_cutoff = []; svals = [None] * T
    for k in range(T - 2, -1, -1):
        uk = [u[j][k] for j in range(n_npi)]
        A = cb.A(uk, Sp[k])
        pm = Pm[k + 1]
        if any(mp.isnan(v) or mp.isinf(v) for r in pm for v in r):
            J = [[mpf(0)] * m for _ in range(m)]
        else:
            X, ranks[k], s, tol, ratio = pinv_exact(pm, near)
            svals[k] = ([float(v) for v in s], float(tol))
            if ratio < near:
                near_cutoff.append((k, float(ratio)))
            J = _matmul(_matmul(Pp[k], _T(A)), X)
        d = [Ss[k + 1][i] - Sm[k + 1][i] for i in range(m)]
        Ss[k] = cb.margins([Sp[k][i] + mp.fsum(J[i][j] * d[j] for j in range(m)) for i in range(m)])
        D = [[Pm[k + 1][i][j] - Ps[k + 1][i][j] for j in range(m)] for i in range(m)]
        JDJ = _matmul(_matmul(J, D), _T(J))
        Ps[k] = _sym([[Pp[k][i][j] - JDJ[i][j] for j in range(m)] for i in range(m)])
        u_s[k], _ = cb.state_update(uk, Ss[k])
    return dict(u_opt=u_opt, u_opt_smooth=u_s, S_MINUS=Sm, S_PLUS=Sp, S_SMOOTH=Ss, P_MINUS=Pm, P_PLUS=Pp, P_SMOOTH=Ps,
                K_GAIN=Kg, innovations=innov, rho=rho, pinv_rank=ranks, near_cutoff=near_cutoff, svals=svals)


def _solve_right(Bm, Am):
    """B / A for square A (mrdivide): X with X A = B, solved exactly (to dps digits) by LU."""
    m = len(Am)
    At = mp.matrix(_T(Am))
    X = [[mpf(0)] * m for _ in range(m)]
    for i in range(m):
        col = mp.lu_solve(At, mp.matrix([[v] for v in Bm[i]]))
        for j in range(m):
            X[i][j] = col[j]
    return X


def _newcase(name, u, x, params, s_init, Ps_init, s_final, Ps_final, v_bar, Q_w, R_v, beta, gamma, L):
    """NewCaseEKFEstimatorWithOptimalNPI.m:28-143: P+ = (I - K C) P- / gamma (:64), no symmetrisation, scalar running R
    (:110-112, carried across missing observations), rho from cc / R without eps (:108), smoother gain by mrdivide with
    no guard (:132), Ps_final overrides the cross product of its non-NaN rows and columns (:125-127)."""
    n_npi, T = len(u), len(x)
    cb = _Callbacks(name, params, n_npi)
    m = cb.m
    Qm = np.asarray(Q_w, dtype=np.float64)
    Q = [[_F(Qm[i, j]) for j in range(m)] for i in range(m)]
    R = _F(np.asarray(R_v, dtype=np.float64).reshape(-1)[0])
    beta_m, gam, v_bar = _F(beta), _F(gamma), _F(v_bar)
    s_minus = [_F(v) for v in np.asarray(s_init, dtype=np.float64).reshape(-1)]
    P_minus = [[_F(Ps_init[i][j]) for j in range(m)] for i in range(m)]
    Sm, Sp, Pm, Pp, Kg, innov, rho, u_opt = [], [], [], [], [], [], [], []
    w_mean, w_cov, w_covn = [mpf(0)] * L, [mpf(0)] * L, [mpf(0)] * L
    I = [[mpf(1) if i == j else mpf(0) for j in range(m)] for i in range(m)]
    for k in range(T):
        Sm.append(list(s_minus)); Pm.append([list(r) for r in P_minus])
        C = cb.C(s_minus)
        xk = cb.h(s_minus, v_bar)
        if not cb.codegen:
            xk = _mmax(mpf(0), xk)
        if not mp.isnan(x[k]):
            inn = x[k] - xk
            PCt = [mp.fsum(P_minus[i][j] * C[j] for j in range(m)) for i in range(m)]
            den = mp.fsum(C[i] * PCt[i] for i in range(m)) + gam * R
            K = [v / den for v in PCt]
            IKC = [[I[i][j] - K[i] * C[j] for j in range(m)] for i in range(m)]
            P_plus = [[v / gam for v in row] for row in _matmul(IKC, P_minus)]
            s_plus = [s_minus[i] + K[i] * inn for i in range(m)]
        else:
            inn = mpf(0); K = [mpf(0)] * m
            P_plus = [list(r) for r in P_minus]; s_plus = list(s_minus)
        s_plus = cb.margins(s_plus)
        uk = [u[j][k] for j in range(n_npi)]
        uo, s_minus = cb.state_update(uk, s_plus)
        A = cb.A(uk, s_plus)
        APA = _matmul(_matmul(A, P_plus), _T(A))
        P_minus = [[APA[i][j] + Q[i][j] for j in range(m)] for i in range(m)]
        s_minus = cb.margins(s_minus)
        Sp.append(s_plus); Pp.append(P_plus); Kg.append(K); innov.append(inn); u_opt.append(uo)
        cnt = min(k + 1, L)
        w_mean = [inn] + w_mean[:L - 1]
        mu = mp.fsum(w_mean) / cnt
        cc = (inn - mu) * (inn - mu)
        w_cov = [cc] + w_cov[:L - 1]
        w_covn = [(cc / R) if R != 0 else (mp.nan if cc == 0 else mp.inf)] + w_covn[:L - 1]
        rho.append(mp.fsum(w_covn) / cnt)
        if beta_m != 1 and not mp.isnan(x[k]):
            R = beta_m * R + (1 - beta_m) * (mp.fsum(w_cov) / cnt)
    Ss = [None] * T; Ps = [None] * T
    Ss[T - 1] = list(Sp[T - 1]); Ps[T - 1] = [list(r) for r in Pp[T - 1]]
    sf = np.asarray(s_final, dtype=np.float64).reshape(-1)
    Pf = np.asarray(Ps_final, dtype=np.float64)
    for i in range(m):
        if not np.isnan(sf[i]):
            Ss[T - 1][i] = _F(sf[i])
    rows, cols = np.nonzero(~np.isnan(Pf))
    for i in np.unique(rows):
        for j in np.unique(cols):
            Ps[T - 1][i][j] = _F(Pf[i, j])
    for k in range(T - 2, -1, -1):
        uk = [u[j][k] for j in range(n_npi)]
        A = cb.A(uk, Sp[k])
        J = _solve_right(_matmul(Pp[k], _T(A)), Pm[k + 1])
        d = [Ss[k + 1][i] - Sm[k + 1][i] for i in range(m)]
        Ss[k] = cb.margins([Sp[k][i] + mp.fsum(J[i][j] * d[j] for j in range(m)) for i in range(m)])
        D = [[Pm[k + 1][i][j] - Ps[k + 1][i][j] for j in range(m)] for i in range(m)]
        JDJ = _matmul(_matmul(J, D), _T(J))
        Ps[k] = [[Pp[k][i][j] - JDJ[i][j] for j in range(m)] for i in range(m)]
    return dict(u_opt=u_opt, u_opt_smooth=[[mpf(0)] * n_npi for _ in range(T)], S_MINUS=Sm, S_PLUS=Sp, S_SMOOTH=Ss, P_MINUS=Pm,
                P_PLUS=Pp, P_SMOOTH=Ps, K_GAIN=Kg, innovations=innov, rho=rho, pinv_rank=[-1] * T, near_cutoff=[], svals=[None] * T)


def _to_float(out, m, n_npi, T, flipped):
    """mpf lists -> MATLAB-shaped fp64 arrays (each value rounded once)."""
    f = lambda v: float(v)
    vec = lambda rows, n: np.array([[f(rows[k][i]) for k in range(T)] for i in range(n)])
    mat = lambda rows: np.array([[[f(rows[k][i][j]) for k in range(T)] for j in range(m)] for i in range(m)])
    res = {"u_opt": vec(out["u_opt"], n_npi), "u_opt_smooth": vec(out["u_opt_smooth"], n_npi),
           "S_MINUS": vec(out["S_MINUS"], m), "S_PLUS": vec(out["S_PLUS"], m), "S_SMOOTH": vec(out["S_SMOOTH"], m),
           "P_MINUS": mat(out["P_MINUS"]), "P_PLUS": mat(out["P_PLUS"]), "P_SMOOTH": mat(out["P_SMOOTH"]),
           "K_GAIN": vec(out["K_GAIN"], m).reshape(m, 1, T), "innovations": np.array([f(v) for v in out["innovations"]]),
           "rho": np.array([f(v) for v in out["rho"]]), "pinv_rank": np.array(out["pinv_rank"], dtype=np.int32)}
    if flipped:
        # SIAlphaModelBackwardEKF.m:30-40: every output reversed in time except rho (squeezed to T x 1 at
        # GenericExtendedKalmanFilter.m:233, so `rho_flipped(:, :, end:-1:1)` reverses a dimension of size 1)
        for k in res:
            if k != "rho":
                res[k] = np.flip(res[k], axis=-1).copy()
    res["near_cutoff"] = np.array(out["near_cutoff"], dtype=np.float64).reshape(-1, 2)
    return res


def run_model(name, u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma,
              inv_monitor_len, order, dps=DEFAULT_DPS, near=10):
    """Same argument list as oracle/ekf_numpy.run_model (the reference's 15 inputs, `params` an ekf_numpy.Params).
    Returns dict of MATLAB-shaped fp64 arrays (the multi-precision results rounded once): the 11 outputs, `pinv_rank`
    (T,), and `near_cutoff` (n, 2): smoother steps k (0-based, in the filter's own time order) at which a singular
    value lies within `near` x of pinv's cut-off, with that ratio.  order 2 == order 1 for these models (their Hessian
    callbacks return zeros, SIAlphaModelEKF.m:92-109)."""
    if name not in MODELS:
        raise ValueError(f"referee_mp: no callbacks for {name}")
    if order not in (1, 2):
        raise ValueError("Undefined order")
    m, flipped, _ = MODELS[name]
    old = mp.dps
    mp.dps = dps
    try:
        u = np.asarray(u, dtype=np.float64); x = np.asarray(x, dtype=np.float64).reshape(-1)
        Pi, Pf = np.asarray(Ps_init, dtype=np.float64), np.asarray(Ps_final, dtype=np.float64)
        if flipped:                                            # SIAlphaModelBackwardEKF.m:19-28
            u, x = u[:, ::-1], x[::-1]
            s_init, s_final, Pi, Pf = s_final, s_init, Pf, Pi
            # Q_w and R_v are handed through un-flipped (:29)
        um = [[_F(v) for v in row] for row in u]
        xm = [_F(v) for v in x]
        if name.startswith("NewCase"):
            out = _newcase(name, um, xm, params, s_init, Pi, s_final, Pf, v_bar, Q_w, R_v, beta, gamma, int(inv_monitor_len))
        else:
            out = _generic(name, um, xm, params, s_init, Pi, s_final, Pf, v_bar, Q_w, R_v, beta, gamma, int(inv_monitor_len), near)
        res = _to_float(out, m, u.shape[0], x.shape[0], flipped)
        if name.startswith("NewCase"):
            del res["u_opt_smooth"]            # the reference has no such output (NewCase...m:1)
        return res
    finally:
        mp.dps = old
