"""NumPy-in / NumPy-out bindings of the HOST-pointer entry points that take a whole stage of
Tools/TrainPredictPrescribeNPI.m for all regions at once -- what a MEX gateway binds (matlab/epiekf_pipeline_mex.cpp).
No torch here: the arrays are plain host memory and the library stages them through the device(s) itself."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import layout as L


def _f(a, keep):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    keep.append(a)
    return a.ctypes.data


def sweep_prescribe(x, u, R_series, region, eps, sp, J0_prefix, J1_prefix, t_hist, L_win=21, order=1, obs_type="NEWCASES",
                    devices=(0,), extras=(), shape=0, time_pipe=0, want_S=True, placement_tries=0):
    """epi_sweep_prescribe_host: the cost-weight sweep of ALL regions (TrainPredictPrescribeNPI.m:421-493, 624-633).

    x, R_series [T, R]; u [T, n, R]; region: dict prm [61, R], s_init [6, R], Ps_init / s_final(6) / Ps_final / Q [36, R];
    eps [P]; sp [48, R]; J0_prefix, J1_prefix [R].  Returns dict J0, J1 [R, P], on_front bool [R, P], i_opt [R] (0-based),
    u_opt [T, n, R], S_opt [T, 6, R] and the per-chain extras named in `extras` ([T, rows, R * P]).
    placement_tries > 1 (epi_prescribe_desc.placement_tries): a call that allocates a new device arena keeps the fastest of that
    many candidates; the report of the first device's block comes back as out["placement"] = {tries, chosen, ms}."""
    keep = []
    T, R = np.shape(x)
    n, P = np.shape(u)[1], len(eps)
    d = _lib.PrescribeDesc()
    d.abi_version, d.R, d.P, d.T, d.t_hist, d.n_npi = _lib.ABI_VERSION, R, P, T, int(t_hist), n
    d.L, d.order, d.obs_type = int(L_win), int(order), L.OBS_IDS.get(obs_type, 99) if isinstance(obs_type, str) else int(obs_type)
    d.shape, d.time_pipe, d.placement_tries = int(shape), int(time_pipe), int(placement_tries)
    ins = _lib.PrescribeInputs()
    ins.x, ins.u, ins.R_series, ins.eps = _f(x, keep), _f(u, keep), _f(R_series, keep), _f(eps, keep)
    for k in ("prm", "s_init", "Ps_init", "s_final", "Ps_final", "Q"):
        setattr(ins, k, _f(region[k], keep))
    ins.sp, ins.J0_prefix, ins.J1_prefix = _f(sp, keep), _f(J0_prefix, keep), _f(J1_prefix, keep)
    out = {"J0": np.empty((R, P)), "J1": np.empty((R, P)), "on_front": np.empty((R, P), dtype=np.int32),
           "i_opt": np.empty((R,), dtype=np.int32), "u_opt": np.empty((T, n, R))}
    if want_S:
        out["S_opt"] = np.empty((T, 6, R))
    outs = _lib.PrescribeOutputs()
    for k in ("J0", "J1", "on_front", "i_opt", "u_opt", "S_opt"):
        setattr(outs, k, out[k].ctypes.data if k in out else None)
    mask = 0
    rows = {"u_opt": n, "u_opt_smooth": n, "S_MINUS": 6, "S_PLUS": 6, "S_SMOOTH": 6, "P_MINUS": 36, "P_PLUS": 36, "P_SMOOTH": 36, "K_GAIN": 6}
    ex = {}
    for name in extras:
        mask |= L.OUT_BITS[name]
        ex[name] = np.empty((T, rows[name], R * P) if name in rows else (T, R * P))
        setattr(outs.extras, name, ex[name].ctypes.data)
    d.out_mask = mask
    rep = _lib.PlacementReport()
    outs.placement = C.addressof(rep)
    ids = (C.c_int * len(devices))(*devices)
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_sweep_prescribe_host(C.byref(d), C.byref(ins), C.byref(outs), len(devices), ids, err)
    _lib.check(rc, err)
    out["on_front"] = out["on_front"].astype(bool)
    out.update(ex)
    out["placement"] = {"tries": int(rep.tries), "chosen": int(rep.chosen), "ms": [float(rep.ms[i]) for i in range(rep.tries)]}
    return out


def preprocess(cases, population, deaths=None, ip=None, W=7, min_cases=1.0, first_num_days=7, device=0):
    """epi_preprocess_host (TrainPredictPrescribeNPI.m:142-198, 201-202, 240): cases / deaths [T, S], population [S],
    ip [T, n, S].  Returns the dict of batch.preprocess as NumPy arrays."""
    keep = []
    T, S = np.shape(cases)
    d = _lib.PreDesc()
    d.abi_version, d.S, d.T, d.n_npi = _lib.ABI_VERSION, S, T, 0 if ip is None else np.shape(ip)[1]
    d.W, d.first_num_days, d.min_cases = int(W), int(first_num_days), float(min_cases)
    names = [k for k in _lib.PRE_OUT_NAMES if not (k == "fatality" and deaths is None) and not (k == "ip_filled" and ip is None)]
    out = {k: np.empty((S,) if k == "I0" else (np.shape(ip) if k == "ip_filled" else (T, S))) for k in names}
    outs = _lib.PreOutputs()
    for k in _lib.PRE_OUT_NAMES:
        setattr(outs, k, out[k].ctypes.data if k in out else None)
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_preprocess_host(C.byref(d), _f(cases, keep), _f(deaths, keep), _f(population, keep), _f(ip, keep), C.byref(outs),
                                        int(device), err)
    _lib.check(rc, err)
    return out


def nnls_affine_fit(X, y, max_iters=100, device=0):
    """epi_nnls_affine_fit_host (TrainPredictPrescribeNPI.m:251-276): X [D, n, S], y [D, S] -> dict a [n, S], b, min_err,
    iters, flag [S]."""
    keep = []
    D, n, S = np.shape(X)
    d = _lib.NnlsDesc()
    d.abi_version, d.S, d.D, d.n, d.max_iters = _lib.ABI_VERSION, S, D, n, int(max_iters)
    out = {"a": np.empty((n, S)), "b": np.empty(S), "min_err": np.empty(S), "iters": np.empty(S, dtype=np.int32),
           "flag": np.empty(S, dtype=np.int32)}
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_nnls_affine_fit_host(C.byref(d), _f(X, keep), _f(y, keep), *(out[k].ctypes.data for k in ("a", "b", "min_err", "iters", "flag")),
                                             int(device), err)
    _lib.check(rc, err)
    return out


def random_npi_mc(sp, u_min, n_scen, K, seed=0, z=None, J0_prefix=None, J1_prefix=None, prefix_days=0, store_u=False, device=0):
    """epi_random_npi_mc_host (TrainPredictPrescribeNPI.m:496-521): sp [48, R], u_min [n, R] -> dict J0, J1 [n_scen, R]
    (+ u [K, n, n_scen * R])."""
    keep = []
    n, R = np.shape(u_min)
    d = _lib.McDesc()
    d.abi_version, d.R, d.n_scen, d.K, d.n_npi = _lib.ABI_VERSION, R, int(n_scen), int(K), n
    d.noise, d.prefix_days = int(z is not None), int(prefix_days)
    d.seed_lo, d.seed_hi = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    out = {"J0": np.empty((n_scen, R)), "J1": np.empty((n_scen, R))}
    if store_u:
        out["u"] = np.empty((K, n, n_scen * R))
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_random_npi_mc_host(C.byref(d), _f(sp, keep), _f(u_min, keep), _f(z, keep), _f(J0_prefix, keep), _f(J1_prefix, keep),
                                           out["u"].ctypes.data if store_u else None, out["J0"].ctypes.data, out["J1"].ctypes.data,
                                           int(device), err)
    _lib.check(rc, err)
    return out


def sir(alpha, beta, gamma, s0, i0, r0, K, dt, device=0):
    """testScripts/testSIR01.m:15-36 (BASELINE config 1): the 3-compartment SIR with return flow, forward Euler.  Scalars or
    arrays of B parameter sets; returns (s, i, r), each [K, B] (the first sample is the initial state)."""
    prm = np.ascontiguousarray(np.stack(np.broadcast_arrays(*[np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (alpha, beta, gamma, s0, i0, r0)])))
    B = prm.shape[1]
    out = np.empty((int(K), 3, B))
    err = C.create_string_buffer(256)
    rc = _lib.lib().epi_sir_sim_host(B, int(K), float(dt), prm.ctypes.data, out.ctypes.data, int(device), err)
    _lib.check(rc, err)
    return out[:, 0], out[:, 1], out[:, 2]
