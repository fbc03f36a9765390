"""Shared helpers for the test-suite: Workload <-> MATLAB-shaped arguments, comparisons."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from epidemicmodeling_amd import layout as L  # noqa: E402
from oracle import ekf_numpy as enp  # noqa: E402
from oracle import oracle_lib as olib  # noqa: E402

OUT_NAMES = ["u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH",
             "K_GAIN", "innovations", "rho"]


def chain_args(w, c):
    """MATLAB-shaped arguments of chain `c` of Workload `w` for oracle/ekf_numpy.run_model."""
    m = w.m
    sx = int(w.x_series[c]) if w.x_series is not None else c
    su = int(w.u_series[c]) if w.u_series is not None else c
    p = enp.Params(
        dt=w.prm[L.PRM_DT, c], beta=w.prm[L.PRM_BETA, c], gamma=w.prm[L.PRM_GAMMA, c],
        sigma=w.prm[L.PRM_SIGMA, c], b=w.prm[L.PRM_B, c], epsilon=w.prm[L.PRM_EPSILON, c],
        s_min=w.prm[L.PRM_S_MIN, c], i_min=w.prm[L.PRM_I_MIN, c],
        alpha_min=w.prm[L.PRM_ALPHA_MIN, c], alpha_max=w.prm[L.PRM_ALPHA_MAX, c],
        a=w.prm[L.PRM_A:L.PRM_A + w.n_npi, c].copy(), u_min=w.prm[L.PRM_U_MIN:L.PRM_U_MIN + w.n_npi, c].copy(),
        u_max=w.prm[L.PRM_U_MAX:L.PRM_U_MAX + w.n_npi, c].copy(),
        w=w.prm[L.PRM_W_EFF:L.PRM_W_EFF + w.n_npi, c].copy(), obs_type=w.obs_type)
    u = np.ascontiguousarray(w.u[:, :, su].T)            # n_npi x T
    x = w.x[:, sx].copy()
    R_v = w.R_series[:, sx].copy() if w.R_series is not None else float(w.R_scalar[c])
    Pi = w.Ps_init[:, c].reshape(m, m, order="F")
    Pf = w.Ps_final[:, c].reshape(m, m, order="F")
    if np.ndim(w.Q) == 3:                                # time-varying: [T][m*m][B] -> m x m x T
        Q = np.ascontiguousarray(w.Q[:, :, c].T).reshape(m, m, -1, order="F")
    else:
        Q = w.Q[:, c].reshape(m, m, order="F")
    return (u, x, p, w.s_init[:, c].copy(), Pi, w.s_final[:, c].copy(), Pf, np.zeros(m),
            float(w.prm[L.PRM_V_BAR, c]), Q, R_v, float(w.prm[L.PRM_BETA_EKF, c]),
            float(w.prm[L.PRM_GAMMA_EKF, c]), w.L, w.order)


def numpy_chain(w, c):
    """Run chain c through the NumPy restatement; returns dict name -> MATLAB-shaped array."""
    out = enp.run_model(w.model, *chain_args(w, c))
    if w.model.startswith("NewCase"):
        names = ["u_opt", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH", "K_GAIN",
                 "innovations", "rho"]
        return dict(zip(names, out))
    d = dict(zip(OUT_NAMES, out[:11]))
    d["pinv_rank"] = out[11]
    return d


def oracle_batch(w, n_threads=0, outputs=None):
    """Run Workload `w` through the C oracle's batched driver."""
    return olib.run_batch(w.model, w.T, w.n_npi, w.L, w.order, w.obs_type, w.x, w.u, w.prm, w.s_init,
                          w.Ps_init, w.s_final, w.Ps_final, w.Q, R_series=w.R_series, R_scalar=w.R_scalar,
                          x_series=w.x_series, u_series=w.u_series, n_threads=n_threads, outputs=outputs)


def batch_chain(out, name, c, m):
    """Chain c of batched output `name` reshaped to the MATLAB shape."""
    a = out[name]
    if a.ndim == 2:
        return a[:, c]
    v = a[:, :, c]                      # [T, rows]
    if name.startswith("P_"):
        return v.T.reshape(m, m, -1, order="F")
    if name == "K_GAIN":
        return v.T.reshape(m, 1, -1)
    return v.T


def oracle_guard_fired(ref, model):
    """[B] bool from an oracle result: did the non-finite guard of GenericEKF.m:211 fire at any executed smoother step
    (pinv_rank == -1 there)?  That is bit 0 of the library's per-chain `status`.  The smoother never executes the last
    column (the first one for the time-flipped wrappers), whose rank word is -1 by convention."""
    rk = ref["pinv_rank"]
    ex = rk[1:] if "Backward" in model else rk[:-1]
    return (ex == -1).any(axis=0)


def rel_err(a, b):
    """max |a-b| / max(|b|) over finite entries, with NaN/Inf patterns required to match."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return np.inf
    if not fa.any():
        return 0.0
    scale = np.max(np.abs(b[fb]))
    if scale == 0:
        return float(np.max(np.abs(a[fa])))
    return float(np.max(np.abs(a[fa] - b[fb])) / scale)


def rowwise_rel_err(a, b):
    """Per-row (first axis) relative error, max over rows: each state component is compared
    against its own magnitude (s ~ 1, i ~ 1e-6, lambda ~ 1e20 must not mask each other)."""
    a = np.asarray(a); b = np.asarray(b)
    return max(rel_err(a[i], b[i]) for i in range(a.shape[0]))


def rowwise_abs_rel_err(a, b, floor=1e-12):
    """Per-row error relative to that row's own magnitude, with an absolute floor so that rows that are
    numerically zero (costates at epsilon -> 1) do not turn rounding noise into O(1) 'relative' error."""
    a = np.asarray(a); b = np.asarray(b)
    worst = 0.0
    for i in range(a.shape[0]):
        fa, fb = np.isfinite(a[i]), np.isfinite(b[i])
        if not np.array_equal(fa, fb):
            return np.inf
        if not fa.any():
            continue
        scale = max(np.max(np.abs(b[i][fb])), floor)
        worst = max(worst, float(np.max(np.abs(a[i][fa] - b[i][fb])) / scale))
    return worst


def load_golden(name):
    """tests/golden/<name>.npz -> (Workload, dict of expected outputs)."""
    from epidemicmodeling_amd import synth
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    g = lambda k: d["in_" + k] if ("in_" + k) in d.files else None
    w = synth.Workload(model=str(d["in_model"]), T=int(d["in_T"]), n_npi=int(d["in_n_npi"]), x=d["in_x"], u=d["in_u"],
                       R_series=g("R_series"), R_scalar=g("R_scalar"), x_series=g("x_series"), u_series=g("u_series"),
                       prm=d["in_prm"], s_init=d["in_s_init"], Ps_init=d["in_Ps_init"], s_final=d["in_s_final"],
                       Ps_final=d["in_Ps_final"], Q=d["in_Q"], L=int(d["in_L"]), order=int(d["in_order"]),
                       obs_type=str(d["in_obs_type"]))
    exp = {k[4:]: d[k] for k in d.files if k.startswith("out_")}
    return w, exp


GOLDEN_CASES = ["sia3_cfg3", "sia6_cfg4", "sia6_row3_adaptiveR", "newcase6_row4", "newcase6_codegen_row4",
                "sia3_backward", "sia6_backward"]


def with_time_varying_q(w, seed=0):
    """Same workload with Q_w as an m x m x T array per chain (GenericExtendedKalmanFilter.m:63-73):
    Q [T][m*m][B], each page the fixed Q scaled by a slowly varying positive factor."""
    import copy
    rng = np.random.default_rng(seed)
    w2 = copy.copy(w)
    scale = 1.0 + 0.5 * np.sin(np.arange(w.T) / 7.0)[:, None, None] + 0.1 * rng.random((w.T, 1, w.B))
    w2.Q = np.ascontiguousarray(w.Q[None, :, :] * scale)
    return w2


def make_regression_problem(S=40, D=120, n=12, seed=0):
    """Regression windows like TrainPredictPrescribeNPI.m:251-253: X = NPI_MAXES - InterventionPlans (small integers,
    step-like in time, some NPIs never changed => constant / collinear columns, some always at the maximum => zero
    columns), y = smoothed alpha.  Returns X [D, n, S], y [D, S]."""
    from epidemicmodeling_amd import synth
    rng = np.random.default_rng(seed)
    umax = synth.IP_MAXES[:n]
    lvl = np.floor(rng.random((D, n, S)) * (umax[None, :, None] + 1))
    keep = rng.random((D, n, S)) < 0.04
    keep[0] = True
    idx = np.maximum.accumulate(np.where(keep, np.arange(D)[:, None, None], 0), axis=0)
    ip = np.take_along_axis(lvl, idx, axis=0)                    # piecewise-constant policies
    if n > 2:
        ip[:, 2, ::3] = 1.0                                       # never changed
    if n > 5:
        ip[:, 5, ::3] = 2.0                                       # never changed (collinear with the one above)
    if n > 7:
        ip[:, 7, ::4] = umax[7]                                   # always at the maximum: zero column of X
    X = umax[None, :, None] - ip
    a_true = np.maximum(rng.normal(0.0, 0.02, (n, S)), 0.0)
    y = np.einsum("dns,ns->ds", X, a_true) + 0.08 + 0.004 * rng.standard_normal((D, S))
    y[:, 1::7] -= 0.2                                             # regions whose mean residual is negative
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def lapack_reading_worker(job):
    """Worker of a process pool (spawn context: the children never touch the GPU): chains `cs` of Workload `w` through
    oracle/ekf_numpy.py -- the independent reading of the .m files that uses LAPACK's SVD for pinv, the closest thing to
    MATLAB's own built-in available here.  Returns the quantities the HIP-vs-LAPACK report compares."""
    w, cs = job[0], job[1]
    one_ulp = len(job) > 2 and job[2]          # also: the same reading with every observation moved by one ulp ("*_1ulp")
    if one_ulp:
        w2 = w.select(np.arange(w.B))
        w2.x = np.nextafter(w.x, np.inf)
    out = []
    for c in cs:
        nd = numpy_chain(w, int(c))
        r = {k: np.asarray(nd[k]) for k in ("S_MINUS", "S_PLUS", "S_SMOOTH", "u_opt_smooth", "pinv_rank") if k in nd}
        if one_ulp:
            n2 = numpy_chain(w2, int(c))
            r.update({k + "_1ulp": np.asarray(n2[k]) for k in ("S_MINUS", "S_PLUS")})
        out.append(r)
    return out


def host_call(w, devices=None, outputs=None, extras=True, shape=0, out=None, timing=None, placement_tries=0, report=None):
    """Workload `w` through the HOST-pointer C ABI (classic layout, numpy arrays [T][rows][B]): epi_ekf_run_host on device
    0, or -- devices = list of device ids -- epi_ekf_run_host_multi with one chain block per entry.  Returns dict of arrays.
    `out`: the dict a previous call returned (its arrays are written again instead of allocating and NaN-filling new ones);
    `timing`: a list that gets the seconds the C call itself took appended; `placement_tries` / `report` (a list that gets
    {"tries", "chosen", "ms"} appended): epi_batch_desc.placement_tries and the epi_placement_report the call fills."""
    import ctypes as C
    import time
    from epidemicmodeling_amd import _lib
    names = [n for n in (outputs or OUT_NAMES) if not (w.model.startswith("NewCase") and n == "u_opt_smooth")]
    m, n_npi, B, T = w.m, w.n_npi, w.B, w.T
    rows = {"u_opt": n_npi, "u_opt_smooth": n_npi, "S_MINUS": m, "S_PLUS": m, "S_SMOOTH": m, "P_MINUS": m * m, "P_PLUS": m * m,
            "P_SMOOTH": m * m, "K_GAIN": m}
    if out is None:
        out = {k: np.full((T, rows[k], B) if k in rows else (T, B), np.nan) for k in names}
    mask = 0
    for k in names:
        mask |= L.OUT_BITS[k]
    Sx = w.x.shape[1]; Su = w.u.shape[2]
    d = _lib.make_desc(w.model, B, T, Sx, Su, n_npi, w.L, w.order, w.obs_type, 1 if w.R_series is not None else 0, mask,
                       1 if np.ndim(w.Q) == 3 else 0)
    d.shape = shape
    d.placement_tries = int(placement_tries)
    ins, outs = _lib.Inputs(), _lib.Outputs()
    rep = _lib.PlacementReport()
    outs.placement = C.addressof(rep)
    keep = []
    def ptr(a, dt=np.float64):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
        return a.ctypes.data
    ins.x_series, ins.u_series = ptr(w.x_series, np.int32), ptr(w.u_series, np.int32)
    ins.x, ins.u, ins.R_series, ins.R_scalar, ins.prm = ptr(w.x), ptr(w.u), ptr(w.R_series), ptr(w.R_scalar), ptr(w.prm)
    ins.s_init, ins.Ps_init, ins.s_final, ins.Ps_final, ins.Q = ptr(w.s_init), ptr(w.Ps_init), ptr(w.s_final), ptr(w.Ps_final), ptr(w.Q)
    for k in names:
        setattr(outs, k, out[k].ctypes.data)
    if extras:
        if "pinv_rank" not in out:
            out["pinv_rank"] = np.full((T, B), -7, dtype=np.int32); out["status"] = np.full((B,), -7, dtype=np.int32)
        outs.pinv_rank, outs.status = out["pinv_rank"].ctypes.data, out["status"].ctypes.data
    err = C.create_string_buffer(256)
    t0 = time.perf_counter()
    if devices is None:
        rc = _lib.lib().epi_ekf_run_host(C.byref(d), C.byref(ins), C.byref(outs), 0, err)
    else:
        ids = (C.c_int * len(devices))(*devices)
        rc = _lib.lib().epi_ekf_run_host_multi(C.byref(d), C.byref(ins), C.byref(outs), len(devices), ids, err)
    if timing is not None:
        timing.append(time.perf_counter() - t0)
    _lib.check(rc, err)
    if report is not None:
        report.append({"tries": int(rep.tries), "chosen": int(rep.chosen), "ms": [float(rep.ms[i]) for i in range(rep.tries)]})
    return out


REFEREE_CASES = ["ref_cfg4_dead_400_120", "ref_cfg4_live_400_120", "ref_cfg4_live_60_120", "ref_row3_adaptiveR_30_120",
                 "ref_cfg3_400", "ref_sia3_backward_120", "ref_sia6_backward_40", "ref_newcase_sweep_400_120",
                 "ref_sia6_backward_150", "ref_sia3_totalcases_200", "ref_cfg4_varying_q_90_30", "ref_cfg3_terminal_120"]
REFEREE_GATE_CAP = 1e-2      # an output whose frozen gate exceeds this is rounding-dominated in fp64: reported, not gated


def referee_inner_gate(fx, base):
    """Round 5 (verdict r04 item 5): the frozen gates are 10 x the larger of the C oracle's and the LAPACK reading's distance from
    the exact result, i.e. set by LAPACK's error -- 5 to 7 decades looser than where the library stands (the smoothed epidemic
    states of `ref_cfg4_dead_400_120` 4.5e-13 against a gate of 2.7e-8): kernel and C oracle could lose five digits TOGETHER
    and stay green.  The inner gate is 100 x the distance the C oracle stood at when the fixture was frozen (at least 1e-12);
    None where that exceeds REFEREE_GATE_CAP (rounding-dominated outputs)."""
    t = max(100.0 * float(fx["dist_C_" + base]), 1e-12)
    return t if t <= REFEREE_GATE_CAP else None


def load_referee(name):
    """tests/golden/<name>.npz (tests/golden/make_golden_referee.py) -> (Workload, dict): `ref_*` the reference's formulas
    in 160-digit arithmetic rounded once (MATLAB shapes + a trailing chain axis), `lap_*` the frozen LAPACK reading,
    `dist_C_* / dist_lap_* / tol_*` the distances measured when the fixture was frozen and the gates derived from them."""
    w, _ = load_golden(name)
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return w, {k: d[k] for k in d.files if not k.startswith("in_")}


def referee_compare(w, got, fx, what):
    """Outputs `got` (batched [T][rows][B] arrays of the chains of fixture `fx`) against the frozen referee and LAPACK
    vectors.  Returns (failures, report): every output whose frozen gate is <= REFEREE_GATE_CAP must lie within the gate
    of the exact result and within 2 x the gate of the LAPACK reading; pinv ranks must equal the exact ranks outside the
    steps the referee lists as ambiguous (a singular value within 10 x of the cut-off); where the smoothed costates are
    rounding-dominated the free controls may differ from the exact plan in no more entries than the LAPACK reading's did."""
    m, B = w.m, w.B
    fails, rep = [], {}
    rows = lambda a: a[None] if a.ndim == 1 else a.reshape(-1, a.shape[-1])
    for key in sorted(k for k in fx if k.startswith("ref_")):
        n = key[4:]
        if n in ("pinv_rank", "near_cutoff"):
            continue
        diag = n.endswith("_diag")
        base = n[:-5] if diag else n
        if base not in got:
            continue
        worst, worst_lap = 0.0, 0.0
        for c in range(B):
            g = batch_chain(got, base, c, m)
            if diag:
                g = np.stack([g[i, i] for i in range(m)])
            worst = max(worst, rowwise_abs_rel_err(rows(g), rows(fx[key][..., c])))
            if "lap_" + n in fx:
                worst_lap = max(worst_lap, rowwise_abs_rel_err(rows(g), rows(fx["lap_" + n][..., c])))
        tol = float(fx["tol_" + base])
        gated = tol <= REFEREE_GATE_CAP
        rep[n] = {"vs_exact": worst, "vs_lapack_reading": worst_lap if "lap_" + n in fx else None, "gate": tol if gated else None,
                  "frozen_distance_C": float(fx["dist_C_" + base]), "frozen_distance_lapack": float(fx["dist_lap_" + base])}
        if gated and not worst <= tol:
            fails.append((what, n, "vs exact", worst, tol))
        inner = referee_inner_gate(fx, base)
        rep[n]["inner_gate"] = inner
        if inner is not None and not worst <= inner:
            fails.append((what, n, "vs exact, inner gate (100 x the C oracle's frozen distance)", worst, inner))
        if gated and "lap_" + n in fx and not worst_lap <= 2.0 * tol:
            fails.append((what, n, "vs LAPACK reading", worst_lap, 2.0 * tol))
    # the epidemic states of S_SMOOTH have a gate of their own (the costates beside them may be rounding-dominated)
    if "ref_S_SMOOTH" in fx and "S_SMOOTH" in got:
        worst = max(rowwise_abs_rel_err(batch_chain(got, "S_SMOOTH", c, m)[:3], fx["ref_S_SMOOTH"][:3, :, c]) for c in range(B))
        tol = float(fx["tol_S_SMOOTH_states"])
        inner = referee_inner_gate(fx, "S_SMOOTH_states")
        rep["S_SMOOTH_states"] = {"vs_exact": worst, "gate": tol, "inner_gate": inner}
        if not worst <= tol:
            fails.append((what, "S_SMOOTH(1:3)", "vs exact", worst, tol))
        if inner is not None and not worst <= inner:
            fails.append((what, "S_SMOOTH(1:3)", "vs exact, inner gate (100 x the C oracle's frozen distance)", worst, inner))
    if "ref_pinv_rank" in fx and "pinv_rank" in got:
        T = w.T
        amb = {(int(c), (T - 1 - int(k)) if "Backward" in w.model else int(k)) for c, k, _ in fx["ref_near_cutoff"]}
        mm = [(c, int(k)) for c in range(B) for k in np.flatnonzero(got["pinv_rank"][:, c] != fx["ref_pinv_rank"][:, c])]
        out = [x for x in mm if x not in amb]
        rep["pinv_rank"] = {"mismatch_steps": len(mm), "outside_ambiguous_steps": len(out), "ambiguous_steps": len(amb)}
        if out:
            fails.append((what, "pinv_rank", "differs from the exact rank away from the cut-off", out[:5], 0))
    if "flips_lap" in fx and "u_opt_smooth" in got:
        flips = []
        for c in range(B):
            su = int(w.u_series[c]) if w.u_series is not None else c
            fm = np.isnan(w.u[:, :, su].T)
            fm[:, 0 if "Backward" in w.model else -1] = False
            flips.append(int(np.sum(batch_chain(got, "u_opt_smooth", c, m)[fm] != fx["ref_u_opt_smooth"][..., c][fm])))
        rep["control_flips_vs_exact"] = {"this": flips, "frozen_C": fx["flips_C"].tolist(), "frozen_lapack": fx["flips_lap"].tolist(),
                                         "free_controls": fx["free_controls"].tolist()}
        if float(fx["tol_u_opt_smooth"]) <= REFEREE_GATE_CAP:
            if sum(flips) != 0:
                fails.append((what, "u_opt_smooth", "differs from the exact plan", sum(flips), 0))
        elif sum(flips) > max(int(fx["flips_lap"].sum()), int(fx["flips_C"].sum())):
            fails.append((what, "u_opt_smooth", "more flips vs the exact plan than either frozen fp64 reading", sum(flips),
                          max(int(fx["flips_lap"].sum()), int(fx["flips_C"].sum()))))
    return fails, rep
