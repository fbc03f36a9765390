"""Measurement of the kernels either side of the filter at the reference's sizes, HIP-event timed.

    python profiles/aux_bench.py [--reps 5] [--pipeline]        # one JSON line per stage
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_aux -o aux -- python3 profiles/aux_bench.py

Sizes: 300 regions, 400 observed + 120 forecast days, 250 cost weights, 500 Monte-Carlo scenarios, 12 NPIs -- the
numbers Tools/TrainPredictPrescribeNPI.m and testPrescribeXPRIZE02.m run with.  `--pipeline` additionally times the whole
chain (epidemicmodeling_amd/pipeline.py) once, cold, end to end including host glue and transfers."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth  # noqa: E402


def regression_problem(S, D, n, seed):
    """X = NPI_MAXES - step-like plans [D, n, S], y = a'x + b + noise [D, S] (some never-changed NPIs)."""
    rng = np.random.default_rng(seed)
    umax = synth.IP_MAXES[:n]
    lvl = np.floor(rng.random((D, n, S)) * (umax[None, :, None] + 1))
    keep = rng.random((D, n, S)) < 0.04
    keep[0] = True
    idx = np.maximum.accumulate(np.where(keep, np.arange(D)[:, None, None], 0), axis=0)
    ip = np.take_along_axis(lvl, idx, axis=0)
    ip[:, 2, ::3] = 1.0; ip[:, 5, ::3] = 2.0
    X = umax[None, :, None] - ip
    a = np.maximum(rng.normal(0.0, 0.02, (n, S)), 0.0)
    return np.ascontiguousarray(X), np.ascontiguousarray(np.einsum("dns,ns->ds", X, a) + 0.08 + 0.004 * rng.standard_normal((D, S)))


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--pipeline", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    S, T, Hh, P, n = 300, 400, 120, 250, 12
    res = []
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(dev)

    raw = synth.make_raw_counts(S, T, seed=0)
    c, d_, N, ip = t(raw["cases"]), t(raw["deaths"]), t(raw["population"]), t(raw["ip"])
    ms = timed(lambda: batch.preprocess(c, N, d_, ip, device=dev), args.reps)
    res.append({"stage": "preprocess_regions + npi_fill", "regions": S, "days": T, "ms": ms,
                "algorithmic_bytes": 8 * T * S * (2 + 12 + 7 + 12), "note": "one lane per region: latency-bound, 5 waves"})

    X, y = regression_problem(S, 120, n, seed=1)
    Xd, yd = t(X), t(y)
    ms = timed(lambda: batch.nnls_affine_fit(Xd, yd, device=dev), args.reps)
    res.append({"stage": "nnls_affine_fit", "regions": S, "days": 120, "ms": ms, "note": "one lane per region, LDS-resident 12x12 solves"})

    for B, draws in ((S, 1), (S * 256, 256)):
        w = synth.make_rt(S, T, n_draws=draws, order=2)
        r = batch.RtRunner(w, dev)
        ms = timed(r.run, args.reps)
        steps = B * T
        res.append({"stage": "rt_expfit_fwd + rt_expfit_bwd (order 2)", "chains": B, "days": T, "ms": ms,
                    "steps_per_s": steps / (ms * 1e-3), "algorithmic_GBs": steps * (8 + 22 * 8) / (ms * 1e-3) / 1e9})
        del r

    sp = np.zeros((batch.SIM_PRM_COUNT, S)); Np = raw["population"]
    sp[0] = 1 - 100 / Np; sp[1] = 100 / Np; sp[2] = synth.ALPHA0; sp[3] = 1e-8; sp[4] = 100.0; sp[5] = 1 / 7; sp[6] = 0.01
    sp[7] = synth.MODEL_BETA; sp[11] = 1.0
    sp[batch.SIM_A:batch.SIM_A + n] = 0.01; sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = synth.IP_MAXES[:, None]
    sp[batch.SIM_W:batch.SIM_W + n] = 1.0
    spd, umin = t(sp), t(np.zeros((n, S)))
    ms = timed(lambda: batch.random_npi_mc(spd, umin, 500, Hh, seed=1, device=dev), args.reps)
    res.append({"stage": "random_npi_mc", "regions": S, "scenarios": 500, "days": Hh, "ms": ms,
                "scenario_days_per_s": S * 500 * Hh / (ms * 1e-3)})

    J0, J1 = torch.rand(S * P, dtype=torch.float64, device=dev), torch.rand(S * P, dtype=torch.float64, device=dev)
    ms = timed(lambda: batch.pareto_front(J0, J1, S), args.reps)
    res.append({"stage": "pareto_front", "regions": S, "points": P, "ms": ms})

    B = S * P
    u = torch.floor(torch.rand((T + Hh, n, B), dtype=torch.float64, device=dev) * 3)
    spb = t(np.repeat(sp, P, axis=1)); z0 = torch.zeros(B, dtype=torch.float64, device=dev)
    ms = timed(lambda: batch.score_sweep(u, T, spb, z0, z0), args.reps)
    res.append({"stage": "sialpha_sim (scoring tail)", "chains": B, "days": Hh, "ms": ms,
                "algorithmic_GBs": B * Hh * n * 8 / (ms * 1e-3) / 1e9})
    del u

    # BASELINE config "SEIRP.m forward RK4, 10k param-set ensemble x 365 days" (dt = 0.1 => 3650 steps), and Euler
    Bs, Ks = 10000, 3650
    rng = np.random.default_rng(3)
    par = t(np.abs(rng.normal(0.2, 0.05, (1, 7, Bs))))
    init = t(np.stack([np.full(Bs, 0.99), np.full(Bs, 0.01), np.zeros(Bs), np.zeros(Bs), np.zeros(Bs)]))
    for integ in ("rk4", "euler"):
        ms = timed(lambda: batch.seirp_sim(par, init, 0.1, Ks, integrator=integ, device=dev), args.reps)
        res.append({"stage": f"seirp_sim ({integ})", "param_sets": Bs, "steps": Ks, "ms": ms,
                    "ensemble_steps_per_s": Bs * Ks / (ms * 1e-3), "written_GBs": Bs * Ks * 5 * 8 / (ms * 1e-3) / 1e9})

    # BASELINE config 1's model (testSIR01.m: SI_Controlled, K = 1500, dt = 0.1) as a 10 000-member ensemble, and
    # NPICost over 75 000 chains x 520 days on its own
    Bi, Ki = 10000, 1500
    al = t(np.full((Ki - 1, Bi), 0.5) * (1 + 0.2 * rng.random((Ki - 1, Bi))))
    ms = timed(lambda: batch.si_controlled(al, np.full(Bi, 0.05), np.full(Bi, 1 - 1e-6), np.full(Bi, 1e-6), Ki, 0.1, device=dev),
               args.reps)
    res.append({"stage": "si_controlled", "chains": Bi, "steps": Ki, "ms": ms, "ensemble_steps_per_s": Bi * Ki / (ms * 1e-3)})
    del al
    Bc, Tc = 75000, 520
    ncs = torch.rand((Tc, Bc), dtype=torch.float64, device=dev)
    uc = t(rng.integers(0, 4, size=(Tc, n, S)).astype(np.float64))
    wc = torch.rand((n, Bc), dtype=torch.float64, device=dev)
    ser = t(np.repeat(np.arange(S), Bc // S).astype(np.int32)).to(torch.int32)
    ms = timed(lambda: batch.npi_cost(ncs, uc, wc, u_series=ser, device=dev), args.reps)
    res.append({"stage": "npi_cost", "chains": Bc, "days": Tc, "ms": ms, "read_GBs": Bc * Tc * 8 / (ms * 1e-3) / 1e9})
    del ncs, uc, wc

    if args.pipeline:
        from epidemicmodeling_amd import pipeline
        raw["cases"][:, -1] = np.cumsum(np.full(T, 40.0))
        times = []
        for _ in range(3):                       # first call pays hipMalloc of ~35 GB and code-object loading
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = pipeline.prescribe(raw["cases"], raw["deaths"], raw["population"], raw["ip"], horizon=Hh, n_eps=P,
                                     num_regression_days=120, device=dev)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        res.append({"stage": "pipeline.prescribe end to end (incl. host glue and PCIe)", "regions": S, "cost_weights": P,
                    "days": T + Hh, "seconds_first_call": times[0], "seconds_later_calls": times[1:],
                    "front_points_mean": float(out["front"].sum(axis=1).mean())})
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
