"""Property-based parity: for randomly drawn problems -- model variant, sizes, missing observations, free controls,
monitor length, adaptive or per-day R, end-point constraints, noise settings, output layout, chunking -- every output of
the HIP path equals the CPU oracle bit for bit (NaN patterns included) and the pinv truncation ranks are identical."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from epidemicmodeling_amd import layout as L_, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

MODELS = ["sia6", "sia3", "sia6", "sia3_bwd", "sia6_bwd", "sia6", "newcase", "newcase_codegen", "row3"]   # the sweep model more often


def build(draw):
    kind = draw(st.sampled_from(MODELS))
    R = draw(st.integers(1, 4)); E = draw(st.integers(1, 5))
    T_hist = draw(st.integers(1, 36)); hor = draw(st.integers(0, 12))
    # one problem in six is long enough (T >= 128) for the launch that pipelines the forward pass in time with the pinv grid
    # (time_pipe = 1 forces it, 0 picks it for batches this small, -1 keeps it off).  The time-flipped wrappers are drawn
    # here too (round 4): they run the epidemic map backwards, and over that many days their covariances overflow (with
    # every observation missing the three-state one reaches 1e152 and then Inf after ~140 days).  A chain that has overflowed
    # carries its non-finite values elsewhere than the dense evaluation when it stays on the packed kernels (a 4 000-example
    # hunt of round 3 found exactly that case; round 3 narrowed the generator); with epi_batch_desc.exact_nonfinite such
    # chains are run again by the dense kernels, and every problem of this generator is run with it, so that an overflow
    # anywhere must come out bit for bit as the oracle has it
    if draw(st.sampled_from([False] * 5 + [True])):
        T_hist = draw(st.integers(128, 170)); hor = draw(st.integers(0, 30))
    seed = draw(st.integers(0, 10 ** 6))
    rng = np.random.default_rng(seed)
    if kind == "sia3":
        w = synth.make_cfg3(R * E, max(T_hist + hor, 1))
    elif kind == "sia6":
        w = synth.make_cfg4(R, E, T_hist, hor)
    elif kind == "sia3_bwd":
        w = synth.as_backward(synth.make_cfg3(R * E, max(T_hist + hor, 2)))
    elif kind == "sia6_bwd":
        w = synth.as_backward(synth.make_cfg4(R, E, max(T_hist, 2), 0))
    elif kind == "row3":
        w = synth.make_row3(R, E, max(T_hist, 2), hor)
    else:
        T = max(T_hist + hor, 3)
        w = synth.make_row4(R, T, min(hor, T - 1), codegen=(kind == "newcase_codegen"))
    generic = not w.model.startswith("NewCase")
    w.L = draw(st.integers(1, 30))
    # missing observations and free (NaN) controls
    if draw(st.booleans()):
        w.x = w.x.copy(); w.x[rng.random(w.x.shape) < draw(st.sampled_from([0.05, 0.3, 1.0]))] = np.nan
    if w.m == 6 and draw(st.booleans()):
        w.u = w.u.copy(); w.u[rng.random(w.u.shape) < 0.2] = np.nan
    # fewer NPIs
    n = draw(st.sampled_from([12, 12, 7, 1]))
    if n < 12:
        w.n_npi = n; w.u = np.ascontiguousarray(w.u[:, :n, :])
        for f in (L_.PRM_A, L_.PRM_U_MIN, L_.PRM_U_MAX, L_.PRM_W_EFF):
            w.prm[f + n:f + 12] = 0.0
    # noise handling: adaptive scalar R vs per-day R (generic models only); fading memory
    w.prm = w.prm.copy()
    w.prm[L_.PRM_BETA_EKF] = draw(st.sampled_from([1.0, 0.9, 0.5]))
    w.prm[L_.PRM_GAMMA_EKF] = draw(st.sampled_from([1.0, 0.995, 0.9]))
    if generic and w.R_series is not None and draw(st.booleans()):
        w.R_scalar = np.full(w.B, float(np.nanmean(w.R_series)) + 1e-14); w.R_series = None
    w.obs_type = draw(st.sampled_from(["NEWCASES", "TOTALCASES"])) if generic else w.obs_type
    # end-point constraints: random (symmetric) NaN pattern / values
    if generic and draw(st.booleans()):
        m = w.m
        w.s_final = w.s_final.copy(); w.Ps_final = w.Ps_final.copy()
        for i in range(m):
            if rng.random() < 0.4:
                w.s_final[i] = rng.random() * 0.5
            for j in range(i, m):
                v = np.nan if rng.random() < 0.5 else (1e-6 * rng.random() if i == j else 1e-9 * (rng.random() - 0.5))
                w.Ps_final[i + m * j] = v; w.Ps_final[j + m * i] = v
    # process noise: diagonal (packed kernels) or full (dense fallback)
    if generic and draw(st.sampled_from([False, False, True])):
        w.Q = w.Q.copy(); w.Q[1] = 1e-13
    lane_block = draw(st.sampled_from([0, 0, 3, 8, 16, 32, 40, "auto"]))
    time_pipe = draw(st.sampled_from([0, 0, 1, -1]))
    # lane mapping of the 6-state generic models: one lane per chain, four lanes per chain, or the library's own choice
    shape = draw(st.sampled_from(["lane", "quad", "wave", "hex", "hex", "auto"]))
    # test hooks of the descriptor: short addressing windows (hex and fixed-descriptor lane kernels), the reverse-time pipeline
    # at a batch size that would not choose it
    hooks = dict(test_window=draw(st.sampled_from([0, 0, 2, 3, 6])), test_flags=draw(st.sampled_from([0, 0, 1, 4, 5, 6])))
    return w, lane_block, time_pipe, kind, shape, hooks


# EPI_FUZZ_EXAMPLES=n runs n freshly drawn examples instead of the fixed 200 (a longer hunt; default stays reproducible)
_N = int(os.environ.get("EPI_FUZZ_EXAMPLES", "0"))


@settings(max_examples=_N or 200, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_problems_match_the_oracle(gpu_device, data):
    from epidemicmodeling_amd import batch
    w, lane_block, time_pipe, kind, shape, hooks = build(data.draw)
    ref = H.oracle_batch(w)
    got = batch.run_workload(w, device=gpu_device, lane_block=lane_block, time_pipe=time_pipe, shape=shape, exact_nonfinite=True, **hooks)
    for n in H.OUT_NAMES:
        if n in ref and n in got:
            assert np.array_equal(got[n], ref[n], equal_nan=True), (kind, w.T, w.B, w.L, lane_block, time_pipe, shape, hooks, n)
    assert np.array_equal(got["pinv_rank"], ref["pinv_rank"]), (kind, w.T, w.B, shape, hooks)
    if not w.model.startswith("NewCase"):
        assert np.array_equal((got["status"] & 1).astype(bool), H.oracle_guard_fired(ref, w.model)), (kind, w.T, w.B, shape)


@settings(max_examples=_N or 80, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_sweeps_through_the_one_call_entry(gpu_device, data):
    """epi_sweep_run_device on randomly drawn sweeps -- observed days from 1, horizons from 1 day (no cut of the smoother) to
    30, long histories (the time-pipelined launch), both lane mappings, layouts, output subsets, missing observations, a
    dense-kernel batch now and then, with and without the Pareto filter: filter outputs equal the oracle's, (J0, J1), front
    and I_opt equal the separate scoring / filter calls on the same u_opt_smooth, bit for bit."""
    import torch
    from epidemicmodeling_amd import batch
    draw = data.draw
    R = draw(st.integers(1, 5)); E = draw(st.integers(1, 9))
    if draw(st.sampled_from([False, False, True])):
        E = draw(st.sampled_from([8, 20, 40]))              # chain counts that are multiples of 40 now and then: with layout block 40 the
        R = draw(st.sampled_from([5, 2, 1])) if E == 8 else R   # one-lane smoother then takes X by LDS-DMA (every lane of every workgroup alive)
    T_hist = draw(st.integers(128, 150)) if draw(st.sampled_from([False] * 4 + [True])) else draw(st.integers(1, 40))
    hor = draw(st.integers(1, 30))
    w = synth.make_cfg4(R, E, T_hist, hor)
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    if draw(st.booleans()):
        w.x = w.x.copy(); w.x[:T_hist][rng.random((T_hist, w.x.shape[1])) < 0.2] = np.nan
    if draw(st.sampled_from([False, False, False, True])):
        w.Q = w.Q.copy(); w.Q[1] = 1e-13                       # non-diagonal Q_w: dense kernels, tail after the smoother
    outputs = draw(st.sampled_from([None, ["u_opt_smooth"], ["u_opt_smooth", "S_SMOOTH"], ["u_opt_smooth", "P_SMOOTH", "rho", "S_PLUS"]]))
    lane_block = draw(st.sampled_from([0, 0, 8, 16, 40, "auto"]))
    shape = draw(st.sampled_from(["lane", "quad", "wave", "hex", "auto"]))
    time_pipe = draw(st.sampled_from([0, 1, -1]))
    hooks = dict(test_window=draw(st.sampled_from([0, 0, 2, 5])), test_flags=draw(st.sampled_from([0, 1, 4, 5])))
    with_front = draw(st.booleans())
    n, B = w.n_npi, w.B
    sp = np.zeros((batch.SIM_PRM_COUNT, B))
    sp[0] = 1.0 - 1e-3 * rng.random(B); sp[1] = 1e-3 * rng.random(B); sp[2] = synth.ALPHA0 * (0.5 + rng.random(B))
    sp[3] = w.prm[L_.PRM_ALPHA_MIN]; sp[4] = w.prm[L_.PRM_ALPHA_MAX]; sp[5] = w.prm[L_.PRM_GAMMA]
    sp[6] = w.prm[L_.PRM_B]; sp[7] = w.prm[L_.PRM_BETA]; sp[11] = 1.0
    sp[batch.SIM_A:batch.SIM_A + n] = w.prm[L_.PRM_A:L_.PRM_A + n]
    sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = w.prm[L_.PRM_U_MAX:L_.PRM_U_MAX + n]
    sp[batch.SIM_W:batch.SIM_W + n] = rng.random((n, B)) + 0.5
    to = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(gpu_device)
    sp_d, j0_d, j1_d = to(sp), to(rng.random(B) * 1e-2), to(rng.random(B) * 40.0)
    ref = H.oracle_batch(w)
    dw = batch.DeviceWorkload(w, gpu_device)
    r = batch.EkfRunner(dw, outputs=outputs, extras=True, lane_block=lane_block, shape=shape, time_pipe=time_pipe, **hooks)
    for t in list(r.out.values()) + [r.ws]:
        t.fill_(float("nan"))
    sc = r.run_sweep(T_hist, sp_d, j0_d, j1_d, n_regions=R if with_front else None)
    torch.cuda.synchronize()
    tag = (R, E, T_hist, hor, outputs, lane_block, shape, time_pipe, hooks, with_front)
    for nme in r.out:
        assert np.array_equal(r.unblocked(nme).cpu().numpy(), ref[nme], equal_nan=True), (tag, nme)
    assert np.array_equal(r.unblocked("pinv_rank").cpu().numpy(), ref["pinv_rank"]), tag
    J0, J1 = sc["J0"].clone(), sc["J1"].clone()
    sc2 = batch.score_sweep(r.out["u_opt_smooth"], T_hist, sp_d, j0_d, j1_d, B=B)
    assert torch.equal(J0, sc2["J0"]) and torch.equal(J1, sc2["J1"]), tag
    if with_front:
        on2, io2 = batch.pareto_front(sc2["J0"], sc2["J1"], R)
        assert torch.equal(sc["on_front"].bool(), on2) and torch.equal(sc["i_opt"], io2), tag


@settings(max_examples=_N or 60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_preprocessing_matches_the_oracle(gpu_device, data):
    """Random cumulative-count columns with gaps, downward corrections, all-missing regions, leading/trailing NaNs and
    random window lengths: every output of epi_preprocess_device equals the oracle bit for bit."""
    from epidemicmodeling_amd import batch
    from oracle import oracle_lib as olib
    draw = data.draw
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    W = draw(st.integers(1, 32)); W2 = int(np.floor(W / 2 + 0.5)); nf = max(1, 3 * (max(W2, 1) - 1))
    T = draw(st.integers(nf + 1, nf + 60)); S = draw(st.integers(1, 9)); n = draw(st.integers(1, 12))
    T = max(T, 2)
    daily = rng.poisson(rng.uniform(0.0, 200.0, (1, S)), (T, S)).astype(np.float64)
    cases = np.cumsum(daily, axis=0); deaths = np.cumsum(rng.poisson(0.5, (T, S)).astype(np.float64), axis=0)
    for arr in (cases, deaths):
        arr[rng.random(arr.shape) < draw(st.sampled_from([0.0, 0.1, 0.6]))] = np.nan
        if draw(st.booleans()):
            arr[rng.integers(0, T):, rng.integers(0, S)] -= 37.0
    if draw(st.booleans()):
        cases[:, 0] = np.nan
    if draw(st.booleans()):
        cases[-1] = np.nan
    pop = 10.0 ** rng.uniform(3, 9, S)
    ip = np.floor(rng.random((T, n, S)) * 4); ip[rng.random(ip.shape) < 0.3] = np.nan
    mc = draw(st.sampled_from([1.0, 0.0, 25.0])); fd = draw(st.integers(0, 9))
    got = {k: v.cpu().numpy() for k, v in batch.preprocess(cases, pop, deaths, ip, W=W, min_cases=mc, first_num_days=fd,
                                                           device=gpu_device).items()}
    for r in range(S):
        ref = olib.preprocess_region(cases[:, r], deaths[:, r], pop[r], W=W, min_cases=mc, first_num_days=fd)
        for k in ("new_refined", "new_smoothed", "zero_lag", "x_new", "x_total", "R_v", "fatality"):
            assert np.array_equal(got[k][:, r], ref[k], equal_nan=True), (W, T, S, r, k)
        assert got["I0"][r] == ref["I0"] or (np.isnan(got["I0"][r]) and np.isnan(ref["I0"]))
        assert np.array_equal(got["ip_filled"][:, :, r], olib.npi_fill(np.ascontiguousarray(ip[:, :, r])))


@settings(max_examples=_N or 60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_regressions_and_fronts_match_the_oracle(gpu_device, data):
    """Random NNLS problems (any rank: duplicated, constant and zero columns, fewer rows than columns, negative targets)
    and random Pareto sets (ties, duplicates, NaNs): device == oracle."""
    import torch
    from epidemicmodeling_amd import batch
    from oracle import oracle_lib as olib
    draw = data.draw
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    S = draw(st.integers(1, 40)); D = draw(st.integers(1, 80)); n = draw(st.integers(1, 12))
    X = np.floor(rng.random((D, n, S)) * draw(st.sampled_from([2, 5])))
    if n > 1 and draw(st.booleans()):
        X[:, 1] = X[:, 0]                                   # duplicated column
    if draw(st.booleans()):
        X[:, n - 1] = draw(st.sampled_from([0.0, 1.0, 3.0]))   # zero / constant column
    a = np.maximum(rng.normal(0, 0.05, (n, S)), 0)
    y = np.einsum("dns,ns->ds", X, a) + rng.normal(draw(st.sampled_from([0.0, 0.1, -0.3])), 0.01, (D, S))
    mit = draw(st.sampled_from([100, 0, 1]))
    got = {k: v.cpu().numpy() for k, v in batch.nnls_affine_fit(X, y, max_iters=mit, device=gpu_device).items()}
    for s in range(S):
        ref = olib.nnls_affine_fit(np.ascontiguousarray(X[:, :, s]), np.ascontiguousarray(y[:, s]), max_iters=mit)
        assert np.array_equal(got["a"][:, s], ref["a"]) and got["b"][s] == ref["b"] and got["iters"][s] == ref["iters"], (S, D, n, s)
        assert got["min_err"][s] == ref["min_err"] and (got["a"][:, s] >= 0).all()
    R = draw(st.integers(1, 12)); P = draw(st.integers(1, 300))
    J0 = np.round(rng.random((R, P)), draw(st.sampled_from([1, 3, 12]))); J1 = np.round(rng.random((R, P)), draw(st.sampled_from([1, 3, 12])))
    J0[rng.random((R, P)) < 0.02] = np.nan; J1[rng.random((R, P)) < 0.02] = np.nan
    on, io = batch.pareto_front(torch.as_tensor(J0.reshape(-1)).to(gpu_device), torch.as_tensor(J1.reshape(-1)).to(gpu_device), R)
    on, io = on.cpu().numpy(), io.cpu().numpy()
    for r in range(R):
        ron, rio = olib.pareto_front(J0[r], J1[r])
        assert np.array_equal(on[r], ron) and io[r] == rio, (R, P, r)


@settings(max_examples=_N or 40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_scenarios_match_the_oracle(gpu_device, data):
    """Random simulator / scenario problems: SIalpha_Controlled with and without noise, NPICost with a historic prefix,
    the random-NPI Monte-Carlo generator (plans and scores), SEIRP (Euler, time-varying parameters): device == oracle."""
    import ctypes as C
    from epidemicmodeling_amd import batch, synth
    from oracle import oracle_lib as olib
    draw = data.draw
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    lib = olib.lib()
    dp = lambda v: v.ctypes.data_as(C.POINTER(C.c_double))
    R = draw(st.integers(1, 6)); n_scen = draw(st.integers(1, 9)); K = draw(st.integers(1, 40)); n = draw(st.integers(1, 12))
    pre = draw(st.sampled_from([0, 0, 17]))
    N = 10.0 ** rng.uniform(3, 9, R)
    sp = np.zeros((batch.SIM_PRM_COUNT, R))
    sp[0] = 1 - 50 / N; sp[1] = 50 / N; sp[2] = rng.uniform(0.05, 1.5, R); sp[3] = draw(st.sampled_from([1e-8, 0.0])); sp[4] = draw(st.sampled_from([100.0, 0.7]))
    sp[5] = rng.uniform(0.01, 0.5, R); sp[6] = rng.uniform(0, 0.05, R); sp[7] = rng.uniform(0.05, 0.4, R)
    sp[8] = 10 / N; sp[9] = 30 / N; sp[10] = 1e-2; sp[11] = draw(st.sampled_from([1.0, 0.5]))
    sp[batch.SIM_A:batch.SIM_A + n] = rng.random((n, R)) * 0.05
    umax = np.floor(rng.random(n) * 4) + 1
    sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = umax[:, None]
    sp[batch.SIM_W:batch.SIM_W + n] = rng.random((n, R)) * 2
    u_min = np.minimum(np.floor(rng.random((n, R)) * 2), umax[:, None])
    noise = draw(st.booleans())
    z = rng.standard_normal((K, 3, n_scen * R)) if noise else None
    J0p = rng.random(R) * 1e-3 if pre else None; J1p = rng.random(R) * 50 if pre else None
    seed = draw(st.integers(0, 2 ** 63 - 1))
    got = batch.random_npi_mc(sp, u_min, n_scen, K, seed=seed, z=z, J0_prefix=J0p, J1_prefix=J1p, prefix_days=pre,
                              store_u=True, device=gpu_device)
    U = got["u"].cpu().numpy(); gJ0 = got["J0"].cpu().numpy(); gJ1 = got["J1"].cpu().numpy()
    for j in range(n_scen):
        for r in range(R):
            c = j * R + r
            plan = olib.random_npi_plan(seed, r, j, n_scen, K, u_min[:, r], umax)
            assert np.array_equal(U[:, :, c].T, plan), (j, r)
            s, i, a = np.zeros(K), np.zeros(K), np.zeros(K)
            aa = np.ascontiguousarray(sp[batch.SIM_A:batch.SIM_A + n, r]); um = np.ascontiguousarray(umax)
            zc = None if z is None else np.ascontiguousarray(z[:, :, c])
            lib.orc_sialpha_controlled(dp(plan), C.c_int(n), *[C.c_double(sp[k, r]) for k in (0, 1, 2)], dp(um),
                                       C.c_double(sp[3, r]), C.c_double(sp[4, r]), C.c_double(sp[5, r]), dp(aa),
                                       C.c_double(sp[6, r]), C.c_double(sp[7, r]), C.c_double(sp[8, r]),
                                       C.c_double(sp[9, r]), C.c_double(sp[10, r]), C.c_int(K), C.c_double(sp[11, r]),
                                       None if zc is None else dp(zc), dp(s), dp(i), dp(a))
            # NPICost continued from the prefix sums, in the reference's summation order
            acc0 = J0p[r] if pre else None; acc1 = J1p[r] if pre else None
            for t in range(K):
                nc = s[t] * i[t] * a[t]
                acc0 = nc if acc0 is None else acc0 + nc
                for k in range(n):
                    term = sp[batch.SIM_W + k, r] * plan[k, t]
                    acc1 = term if acc1 is None else acc1 + term
            assert gJ0[j, r] == acc0 / (K + pre) and gJ1[j, r] == acc1 / (n * (K + pre)), (j, r, pre, noise)
    # SEIRP, Euler, time-varying parameters
    B = draw(st.integers(1, 70)); Ks = draw(st.integers(2, 60)); dt = draw(st.sampled_from([0.1, 1.0, 0.01]))
    par = np.abs(rng.normal(0.2, 0.1, (Ks, 7, B))) if draw(st.booleans()) else np.abs(rng.normal(0.2, 0.1, (1, 7, B)))
    init = np.abs(rng.normal(0.2, 0.1, (5, B)))
    out = batch.seirp_sim(par, init, dt, Ks, device=gpu_device).cpu().numpy()
    for c in {0, B // 2, B - 1}:
        cols = [np.ascontiguousarray(par[:, j, c]) if par.shape[0] > 1 else np.full(Ks, par[0, j, c]) for j in range(7)]
        res = [np.zeros(Ks) for _ in range(5)]
        lib.orc_seirp(*[dp(v) for v in cols], *[C.c_double(init[q, c]) for q in range(5)], C.c_int(Ks), C.c_double(dt),
                      *[dp(v) for v in res])
        for q in range(5):
            assert np.array_equal(out[:, q, c], res[q]), (B, Ks, dt, c, q)
    # SI_Controlled with shared infection-rate series, and NPICost on its own (weights per day or one column, shared inputs)
    Bi = draw(st.integers(1, 90)); Ki = draw(st.integers(1, 50)); Sa = draw(st.integers(1, Bi))
    al = rng.uniform(0.0, 3.0, (max(Ki - 1, 1), Sa)); ser = rng.integers(0, Sa, Bi).astype(np.int32) if Sa != Bi or draw(st.booleans()) else None
    beta = rng.uniform(0.0, 0.5, Bi); s0 = rng.uniform(0.0, 1.0, Bi); i0 = rng.uniform(0.0, 1.0, Bi); dts = draw(st.sampled_from([0.1, 1.0, 2.5]))
    gs, gi = (t.cpu().numpy() for t in batch.si_controlled(al, beta, s0, i0, Ki, dts, alpha_series=ser, device=gpu_device))
    for c in {0, Bi // 2, Bi - 1}:
        col = np.ascontiguousarray(al[:, c if ser is None else ser[c]])
        rs, ri = np.zeros(Ki), np.zeros(Ki)
        lib.orc_si_controlled(dp(col), C.c_double(beta[c]), C.c_double(s0[c]), C.c_double(i0[c]), C.c_int(Ki), C.c_double(dts), dp(rs), dp(ri))
        assert np.array_equal(gs[:, c], rs) and np.array_equal(gi[:, c], ri), (Bi, Ki, Sa, c)
    Bc = draw(st.integers(1, 90)); Tc = draw(st.integers(1, 40)); nn = draw(st.integers(1, 12)); Su = draw(st.integers(1, Bc))
    ub = rng.integers(0, 5, size=(Tc, nn, Su)).astype(np.float64); nc = rng.random((Tc, Bc))
    ser = rng.integers(0, Su, Bc).astype(np.int32) if Su != Bc or draw(st.booleans()) else None
    per_day = draw(st.booleans())
    wb = rng.random((Tc, nn, Bc)) if per_day else rng.random((nn, Bc))
    g0, g1 = (t.cpu().numpy() for t in batch.npi_cost(nc, ub, wb, u_series=ser, device=gpu_device))
    for c in {0, Bc // 2, Bc - 1}:
        uc = np.asfortranarray(ub[:, :, c if ser is None else ser[c]].T)
        wc = np.asfortranarray(wb[:, :, c].T) if per_day else np.asfortranarray(np.repeat(wb[:, c][:, None], Tc, axis=1))
        J0, J1 = C.c_double(), C.c_double()
        lib.orc_npi_cost(dp(np.ascontiguousarray(nc[:, c])), dp(uc), dp(wc), C.c_int(nn), C.c_int(Tc), C.byref(J0), C.byref(J1))
        assert (g0[c], g1[c]) == (J0.value, J1.value), (Bc, Tc, nn, Su, per_day, c)


@settings(max_examples=_N or 40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_rt_expfit_matches_the_oracle(gpu_device, data):
    """Random Rt_ExpFitEKF problems (both orders, gaps, forecast tails, monitor lengths, noise settings, shared series):
    bit for bit against the oracle (exp/tanh are evaluated in one fixed operation order on both sides)."""
    from epidemicmodeling_amd import batch, synth
    from oracle import oracle_lib as olib
    draw = data.draw
    order = draw(st.sampled_from([1, 2])); S = draw(st.integers(1, 8)); T = draw(st.integers(16, 80))
    nd = draw(st.sampled_from([1, 1, 3])); L = draw(st.integers(1, 30)); hor = draw(st.integers(0, 10))
    w = synth.make_rt(S, T, n_draws=nd, order=order, horizon=hor, seed=draw(st.integers(0, 10 ** 6)),
                      w_bar=draw(st.sampled_from([(0.0, 0.0), (0.5, 1e-4), (-2.0, -1e-3)])), L=L)
    cut = draw(st.integers(1, T))
    w.x = np.ascontiguousarray(w.x[:cut])
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    if draw(st.booleans()):
        w.x[rng.random(w.x.shape) < 0.15] = np.nan
    w.rp = w.rp.copy()
    w.rp[7] = draw(st.sampled_from([0.9, 1.0, 0.5])); w.rp[8] = draw(st.sampled_from([0.995, 1.0, 0.9]))
    w.rp[1] = draw(st.sampled_from([0.9, 1.0, 0.3])); w.rp[2] = draw(st.sampled_from([0.1, 1.0, 0.01]))
    got = batch.rt_expfit(w, gpu_device)
    ref = olib.rt_expfit_batch(w.x, w.rp, w.L, order, x_series=w.x_series)
    for n in ref:
        assert np.array_equal(got[n], ref[n], equal_nan=True), (order, cut, S, L, n)


@settings(max_examples=_N or 50, deadline=None, suppress_health_check=list(HealthCheck), derandomize=not _N)
@given(st.data())
def test_random_tools_calls_match_the_oracle(gpu_device, data):
    """The Tools/-named functions with MATLAB-shaped arguments in all the forms the reference accepts -- Q_w scalar,
    m x m, length-T vector, m x m x D pages; R_v scalar or 1 x T; params.w scalar / column / row / n x D (the
    implicit-expansion quirk); observation types -- equal the oracle called with the same arguments."""
    from epidemicmodeling_amd import synth, tools
    from oracle import oracle_lib as olib
    from oracle.ekf_numpy import resolve_w
    draw = data.draw
    rng = np.random.default_rng(draw(st.integers(0, 10 ** 6)))
    name = draw(st.sampled_from(["SIAlphaModelEKF", "SIAlphaModelEKFOptControlled", "SIAlphaModelBackwardEKF",
                                 "SIAlphaModelBackwardEKFOptControlled", "NewCaseEKFEstimatorWithOptimalNPI"]))
    six = "OptControlled" in name or name.startswith("NewCase")
    generic = not name.startswith("NewCase")
    m = 6 if six else 3
    T = draw(st.integers(1, 25)); n = draw(st.sampled_from([12, 12, 4]))
    N = 10.0 ** rng.uniform(4, 8)
    u = np.floor(rng.random((n, T)) * 3)
    if six:
        u[rng.random((n, T)) < 0.3] = np.nan
    x = 1e-5 * (1 + rng.random((1, T)))
    x[0, rng.random(T) < 0.2] = np.nan
    wform = draw(st.sampled_from(["scalar", "col", "row", "mat"]))
    wv = {"scalar": 0.7, "col": rng.random((n, 1)) + 0.1, "row": rng.random((1, n)) + 0.1, "mat": rng.random((n, 3)) + 0.1}[wform]
    params = dict(dt=1.0, a=rng.random(n) * 0.02, b=0.03, u_min=np.zeros(n), u_max=synth.IP_MAXES[:n].copy(),
                  alpha_min=1e-8, alpha_max=100.0, gamma=1 / 7, beta=synth.MODEL_BETA, sigma=1e6,
                  epsilon=draw(st.sampled_from([1e-9, 0.3, 1.0])), w=wv, obs_type=draw(st.sampled_from(["NEWCASES", "TOTALCASES"])),
                  s_min=1 / N, i_min=1 / N)
    base = [0.99, 0.01, 1.1] + [0.0] * (m - 3)
    s_init = np.array(base); Ps_init = np.diag(rng.random(m) * 1e-4 + 1e-8)
    flipped = "Backward" in name
    if flipped or (generic and draw(st.booleans())):
        s_final = np.array(base) * (1 + 0.01 * rng.random(m)); Ps_final = np.diag(rng.random(m) * 1e-4 + 1e-8)
    else:
        s_final = np.full(m, np.nan); Ps_final = np.full((m, m), np.nan)
    qd = rng.random(m) * 1e-6 + 1e-10
    qform = draw(st.sampled_from(["mat", "scalar", "vec", "pages"])) if generic else draw(st.sampled_from(["mat", "scalar"]))
    if qform == "mat":
        Q_w = np.diag(qd); Q_full = Q_w
    elif qform == "scalar":
        Q_w = float(qd[0]); Q_full = qd[0] * np.eye(m)
    elif qform == "vec":
        Q_w = 1e-7 * (1 + rng.random(T)); Q_full = Q_w[None, None, :] * np.eye(m)[:, :, None]
        if T == 1:
            Q_full = Q_full[:, :, 0]          # a length-1 vector IS the scalar form
    else:
        D = draw(st.integers(2, 4)); Q_w = np.stack([np.diag(qd * (k + 1)) for k in range(D)], axis=2)
        Q_full = Q_w[:, :, np.arange(T) % D]
    rform = draw(st.sampled_from(["scalar", "vec"])) if generic and T > 1 else "scalar"
    R_v = 1e-10 if rform == "scalar" else 1e-10 * (1 + rng.random(T))
    beta = draw(st.sampled_from([1.0, 0.9])); gam = draw(st.sampled_from([1.0, 0.995])); Lm = draw(st.integers(1, 25))
    fn = getattr(tools, name)
    got = fn(u, x, params, s_init, Ps_init, s_final, Ps_final, np.zeros(m), 0.0, Q_w, R_v, beta, gam, Lm, 1)
    w_eff = np.zeros(12); w_eff[:n] = resolve_w(wv, n) if six else 0.0
    ref = olib.run(name, u, x.reshape(-1), params, w_eff, s_init, Ps_init, s_final, Ps_final, 0.0, Q_full, R_v, beta, gam, Lm, 1)
    names = [k for k in H.OUT_NAMES if generic or k != "u_opt_smooth"]
    assert len(got) == len(names)
    for k, g in zip(names, got):
        assert np.array_equal(np.asarray(g).reshape(-1), np.asarray(ref[k]).reshape(-1), equal_nan=True), (name, T, n, qform, rform, wform, k)
