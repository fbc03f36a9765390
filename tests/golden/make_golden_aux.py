"""Generates tests/golden/aux_*.npz: seeded inputs + expected outputs of the stages around the filter
(SURVEY.md 8(f)): Rt_ExpFitEKF, per-region preprocessing, the NNLS regression, random-NPI plans and the Pareto filter.

As for make_golden.py the reference cannot run here, so outputs come from oracle/ekf_oracle.c and generation FAILS
unless the independent reading agrees: oracle/ekf_numpy.py for Rt_ExpFitEKF and the preprocessing (SciPy's lfilter /
filtfilt), SciPy's Lawson-Hanson for the NNLS, a vectorised NumPy restatement for the Pareto filter, the Random123
known-answer vectors for the generator behind the plans.  Only data is stored.

    python tests/golden/make_golden_aux.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from epidemicmodeling_amd import synth  # noqa: E402
from oracle import ekf_numpy as enp  # noqa: E402
from oracle import oracle_lib as olib  # noqa: E402
from tests import helpers as H  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rt_case(order):
    w = synth.make_rt(5, 140, order=order, horizon=14, w_bar=(0.5, 1e-4), seed=order)
    w.x[40:44, 1] = np.nan
    ob = olib.rt_expfit_batch(w.x, w.rp, w.L, order)
    names = ["S_MINUS", "S_PLUS", "P_MINUS", "P_PLUS", "K_GAIN", "S_SMOOTH", "P_SMOOTH", "innovations", "rho"]
    for c in range(w.B):
        rp = w.rp[:, c]
        nd = dict(zip(names, enp.rt_expfit_ekf(w.x[:, c], rp[9:11], rp[0:3], rp[3:5], rp[5], rp[11:15].reshape(2, 2, order="F"),
                                               rp[15:19].reshape(2, 2, order="F"), rp[6], rp[7], rp[8], w.L, order)))
        assert H.rel_err(ob["S_SMOOTH"][:, :, c].T, nd["S_SMOOTH"]) <= 1e-11 and H.rel_err(ob["rho"][:, c], nd["rho"]) <= 1e-11
    d = {"in_x": w.x, "in_rp": w.rp, "in_L": w.L, "in_order": order}
    d.update({"out_" + k: v for k, v in ob.items()})
    return d


def pre_case():
    raw = synth.make_raw_counts(6, 120, seed=8)
    keys = ["new_refined", "new_smoothed", "zero_lag", "x_new", "x_total", "R_v", "fatality"]
    out = {k: np.zeros((120, 6)) for k in keys}
    I0 = np.zeros(6)
    for r in range(6):
        a = olib.preprocess_region(raw["cases"][:, r], raw["deaths"][:, r], raw["population"][r])
        b = enp.preprocess_region(raw["cases"][:, r], raw["deaths"][:, r], raw["population"][r])
        for k in keys:
            assert H.rel_err(a[k], b[k]) <= 1e-13, (r, k)
            out[k][:, r] = a[k]
        I0[r] = a["I0"]
    ipf = np.stack([olib.npi_fill(np.ascontiguousarray(raw["ip"][:, :, r])) for r in range(6)], axis=2)
    assert np.array_equal(ipf[:, :, 2], enp.npi_fill(raw["ip"][:, :, 2]))
    d = {"in_" + k: raw[k] for k in ("cases", "deaths", "population", "ip")}
    d.update({"out_" + k: v for k, v in out.items()})
    d["out_I0"] = I0; d["out_ip_filled"] = ipf
    return d


def nnls_case():
    from scipy.optimize import nnls as sp_nnls
    X, y = H.make_regression_problem(9, 80, 12, seed=4)
    a = np.zeros((12, 9)); b = np.zeros(9); e = np.zeros(9); it = np.zeros(9, dtype=np.int32)
    for s in range(9):
        f = olib.nnls_affine_fit(np.ascontiguousarray(X[:, :, s]), np.ascontiguousarray(y[:, s]))
        ref, rn = sp_nnls(X[:, :, s], y[:, s])
        assert abs(np.linalg.norm(X[:, :, s] @ olib.nnls(X[:, :, s], y[:, s]) - y[:, s]) - rn) <= 1e-12
        a[:, s], b[s], e[s], it[s] = f["a"], f["b"], f["min_err"], f["iters"]
    return {"in_X": X, "in_y": y, "out_a": a, "out_b": b, "out_min_err": e, "out_iters": it}


def scenario_case():
    assert olib.philox4x32_10([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    R, n_scen, K = 3, 10, 12
    plans = np.zeros((n_scen, R, 12, K))
    for j in range(n_scen):
        for r in range(R):
            plans[j, r] = olib.random_npi_plan(0xABCDEF0123, r, j, n_scen, K, np.zeros(12), synth.IP_MAXES)
    rng = np.random.default_rng(6)
    J0, J1 = rng.random((4, 40)), rng.random((4, 40))
    J0[:, 3] = J0[:, 1]; J0[0, 5] = np.nan
    on = np.zeros((4, 40), dtype=bool); io = np.zeros(4, dtype=np.int32)
    for r in range(4):
        on[r], io[r] = olib.pareto_front(J0[r], J1[r])
        dom = (J0[r][None, :] < J0[r][:, None]) & (J1[r][None, :] < J1[r][:, None])
        assert np.array_equal(on[r], dom.sum(axis=1) == 0)
    return {"in_seed": 0xABCDEF0123, "in_n_scen": n_scen, "in_K": K, "out_plans": plans, "in_J0": J0, "in_J1": J1,
            "out_on_front": on, "out_i_opt": io}


def main():
    for name, d in (("aux_rt_order1", rt_case(1)), ("aux_rt_order2", rt_case(2)), ("aux_preprocess", pre_case()),
                    ("aux_nnls", nnls_case()), ("aux_scenarios", scenario_case())):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
