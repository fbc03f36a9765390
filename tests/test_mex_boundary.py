"""The MATLAB-facing boundary, executed: the five MEX gateways of matlab/ are compiled against tests/mex_shim/mex.h (a small
IMPLEMENTED stand-in for the MEX / C Matrix API), linked with libepiekf.so and driven by tests/mex_shim/driver.cpp with
MATLAB-shaped column-major arrays.  What is checked: shapes and order of the returned struct fields (the reference's output
order, Tools/SIAlphaModelEKF.m:1), bit-for-bit equality with the CPU oracle, the reference's error texts, and -- without a
GPU -- that the gateways build and link against the real library and that every 1-based offset in
matlab/Tools/epiekf_pack_params.m agrees with include/epiekf_layout.h."""
import os
import re
import shutil
import struct
import subprocess

import numpy as np
import pytest

from epidemicmodeling_amd import layout as L
from epidemicmodeling_amd import synth
from tests import helpers as H

SHIM = os.path.join(H.ROOT, "tests", "mex_shim")
BUILD = os.path.join(SHIM, "build")
GATEWAYS = {"epiekf": "epiekf_mex.cpp", "batch": "epiekf_batch_mex.cpp", "rt": "epiekf_rt_mex.cpp", "sim": "epiekf_sim_mex.cpp",
            "pipeline": "epiekf_pipeline_mex.cpp"}
CLS = {np.dtype(np.float64): 6, np.dtype(np.int32): 12, np.dtype(np.uint8): 4}
DT = {6: np.float64, 12: np.int32, 4: np.uint8}


@pytest.fixture(scope="module")
def driver(hip_lib):
    """Compile the gateways (-Wall -Wextra -Werror) with mexFunction renamed per gateway and link the driver."""
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    os.makedirs(BUILD, exist_ok=True)
    libdir = os.path.join(H.ROOT, "epidemicmodeling_amd")
    objs = []
    common = [cxx, "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + SHIM, "-I" + os.path.join(H.ROOT, "include")]
    for name, src in GATEWAYS.items():
        o = os.path.join(BUILD, name + ".o")
        subprocess.run(common + ["-DmexFunction=mex_" + name, "-c", os.path.join(H.ROOT, "matlab", src), "-o", o], check=True)
        objs.append(o)
    exe = os.path.join(BUILD, "driver")
    subprocess.run(common + [os.path.join(SHIM, "driver.cpp"), os.path.join(SHIM, "mex_shim.cpp"), *objs, "-L" + libdir, "-lepiekf",
                             "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe], check=True)
    return exe


def _write(path, arrays):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(arrays)))
        for a in arrays:
            if isinstance(a, str):
                a = np.frombuffer(a.encode(), dtype=np.uint8).reshape(1, -1)
            a = np.asarray(a)
            if a.dtype not in CLS:
                a = a.astype(np.float64)
            if a.ndim < 2:
                a = a.reshape(-1, 1) if a.ndim == 1 else a.reshape(1, 1)
            f.write(struct.pack("<ii", CLS[a.dtype], a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(np.asfortranarray(a).tobytes(order="F"))


def _read(path):
    out = []
    with open(path, "rb") as f:
        n, = struct.unpack("<i", f.read(4))
        for _ in range(n):
            cls, nd = struct.unpack("<ii", f.read(8))
            dims = struct.unpack("<%dq" % nd, f.read(8 * nd))
            cnt = int(np.prod(dims))
            out.append(np.frombuffer(f.read(cnt * np.dtype(DT[cls]).itemsize), dtype=DT[cls]).reshape(dims, order="F"))
    return out


def _call(driver, gateway, args, nlhs=1, tmp=None, expect_error=None):
    tmp = tmp or BUILD
    fin, fout = os.path.join(tmp, gateway + "_in.bin"), os.path.join(tmp, gateway + "_out.bin")
    _write(fin, args)
    r = subprocess.run([driver, gateway, fin, fout, str(nlhs)], capture_output=True, text=True, timeout=300)
    if expect_error is not None:
        assert r.returncode == 3 and expect_error in r.stderr, (r.returncode, r.stderr[-400:])
        return None
    assert r.returncode == 0, r.stderr[-2000:]
    return _read(fout)


def test_gateways_build_and_link_against_the_library(driver):
    """No GPU needed: the four gateways compile warning-free against the C Matrix API signatures and every epi_* symbol
    they call resolves in libepiekf.so."""
    assert os.path.exists(driver)


def test_pack_params_offsets_match_the_layout_header():
    """matlab/Tools/epiekf_pack_params.m writes the EPI_PRM_* block with 1-based indices: every one of them is parsed out
    of the .m text and checked against include/epiekf_layout.h (0-based)."""
    src = open(os.path.join(H.ROOT, "matlab", "Tools", "epiekf_pack_params.m")).read()
    hdr = open(os.path.join(H.ROOT, "include", "epiekf_layout.h")).read()
    c = dict((k, int(v)) for k, v in re.findall(r"(EPI_PRM_[A-Z_]+)\s*=\s*(\d+)", hdr))
    assert int(re.search(r"prm = zeros\((\d+), 1\)", src).group(1)) == c["EPI_PRM_COUNT"]
    scalars = dict((f, int(i)) for i, f in re.findall(r"prm\((\d+)\)\s*=\s*(?:params\.)?([a-z_]+);", src))
    want = {"dt": "DT", "beta": None, "gamma": None, "b": "B", "alpha_min": "ALPHA_MIN", "alpha_max": "ALPHA_MAX",
            "s_min": "S_MIN", "i_min": "I_MIN", "sigma": "SIGMA", "epsilon": "EPSILON", "v_bar": "V_BAR"}
    for f, name in want.items():
        if name:
            assert scalars[f] == c["EPI_PRM_" + name] + 1, f
    # beta / gamma appear twice: params.beta / params.gamma (model) and the filter's beta / gamma arguments
    both = re.findall(r"prm\((\d+)\)\s*=\s*(params\.)?(beta|gamma);", src)
    got = {(bool(p), f): int(i) for i, p, f in both}
    assert got[(True, "beta")] == c["EPI_PRM_BETA"] + 1 and got[(True, "gamma")] == c["EPI_PRM_GAMMA"] + 1
    assert got[(False, "beta")] == c["EPI_PRM_BETA_EKF"] + 1 and got[(False, "gamma")] == c["EPI_PRM_GAMMA_EKF"] + 1
    vectors = dict((f, int(i)) for i, f in re.findall(r"prm\((\d+) \+ \(1:n_npi\)\)\s*=\s*(?:params\.)?([a-z_]+)\(:\);", src))
    assert vectors == {"a": c["EPI_PRM_A"], "u_max": c["EPI_PRM_U_MAX"], "u_min": c["EPI_PRM_U_MIN"], "w_eff": c["EPI_PRM_W_EFF"]}
    assert c["EPI_PRM_A"] + 12 <= c["EPI_PRM_U_MIN"] and c["EPI_PRM_W_EFF"] + 12 <= c["EPI_PRM_V_BAR"]   # room for 12 NPIs each
    # the Python mirror of the same block (layout.py) is what the tests below pack with
    for k, v in c.items():
        assert getattr(L, k.replace("EPI_PRM_", "PRM_")) == v


ORDER11 = ["u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS", "P_PLUS", "P_SMOOTH", "K_GAIN", "innovations", "rho"]


def _matlab_args_one_chain(w, c):
    """The 13 arguments matlab/Tools/SIAlphaModelEKFOptControlled.m hands to epiekf_mex for chain c (MATLAB shapes)."""
    m, T = w.m, w.T
    su = int(w.u_series[c]) if w.u_series is not None else c
    sx = int(w.x_series[c]) if w.x_series is not None else c
    R = w.R_series[:, sx].reshape(1, T) if w.R_series is not None else np.array([[w.R_scalar[c]]])
    return [float(L.MODEL_IDS[w.model]), w.u[:, :, su].T, w.x[:, sx].reshape(1, T), w.prm[:, c].reshape(-1, 1),
            w.s_init[:, c].reshape(m, 1), w.Ps_init[:, c].reshape(m, m, order="F"), w.s_final[:, c].reshape(m, 1),
            w.Ps_final[:, c].reshape(m, m, order="F"), w.Q[:, c].reshape(m, m, order="F"), R, float(w.L), float(w.order),
            float(L.OBS_IDS[w.obs_type])]


@pytest.mark.gpu
def test_single_call_gateway_returns_the_reference_outputs(gpu_device, driver):
    """epiekf_mex with one chain of the headline sweep, of the 3-state model and of a time-flipped model: eleven struct
    fields in the reference's order, MATLAB's shapes (m x T, m x m x T, m x 1 x T, 1 x T, rho T x 1), values bit for bit
    the oracle's."""
    for w, c in ((synth.make_cfg4(2, 3, 60, 20), 4), (synth.make_cfg3(3, 50), 1), (synth.as_backward(synth.make_cfg4(2, 2, 30, 0)), 3),
                 (synth.make_row3(2, 2, 20, 8), 2)):
        m, T, n = w.m, w.T, w.n_npi
        res = _call(driver, "epiekf", _matlab_args_one_chain(w, c))
        assert len(res) == 11
        ref = H.oracle_batch(w.select(np.array([c])))
        shapes = {"u_opt": (n, T), "u_opt_smooth": (n, T), "S_MINUS": (m, T), "S_PLUS": (m, T), "S_SMOOTH": (m, T), "P_MINUS": (m, m, T),
                  "P_PLUS": (m, m, T), "P_SMOOTH": (m, m, T), "K_GAIN": (m, 1, T), "innovations": (1, T), "rho": (T, 1)}
        for name, got in zip(ORDER11, res):
            assert got.shape == shapes[name], (w.model, name, got.shape)
            want = ref[name][..., 0]                               # [T][rows] of the single chain
            if name.startswith("P_"):
                want = want.T.reshape(m, m, T, order="F")
            elif name == "K_GAIN":
                want = want.T.reshape(m, 1, T)
            elif name == "innovations":
                want = want.reshape(1, T)
            elif name == "rho":
                want = want.reshape(T, 1)
            else:
                want = want.T
            assert np.array_equal(got, want, equal_nan=True), (w.model, name)


@pytest.mark.gpu
def test_gateway_errors_carry_the_reference_texts(gpu_device, driver):
    w = synth.make_cfg4(1, 2, 20, 5)
    a = _matlab_args_one_chain(w, 0)
    bad = list(a); bad[11] = 3.0
    _call(driver, "epiekf", bad, expect_error="Undefined order")                                  # GenericEKF.m:111
    bad = list(a); bad[9] = np.ones((1, w.T + 1))
    _call(driver, "epiekf", bad, expect_error="Observation noise covariance noise mismatch")      # GenericEKF.m:90
    bad = list(a); bad[8] = np.ones((5, 5))
    _call(driver, "epiekf", bad, expect_error="Process noise covariance noise mismatch")          # GenericEKF.m:75
    bad = list(a); bad[12] = 7.0
    _call(driver, "epiekf", bad, expect_error="unknown observation type")                         # SIAlphaModelEKF.m:57


@pytest.mark.gpu
def test_batched_gateway_chain_first_arrays(gpu_device, driver):
    """epiekf_batch_mex as matlab/Tools/SIAlphaModelEKFOptControlledSweep.m calls it: chain index FIRST (B x rows x T), the
    cost weights of a region sharing its single series through zero-based int32 x_series / u_series."""
    w = synth.make_cfg4(1, 9, 40, 12)
    B, T, m, n = w.B, w.T, 6, 12
    args = [1.0, np.transpose(w.u, (2, 1, 0)), w.x.T, w.prm.T, w.s_init.T, w.Ps_init.T, w.s_final.T, w.Ps_final.T, w.Q.T,
            w.R_series.T, float(w.L), float(w.order), 0.0, w.x_series.astype(np.int32).reshape(B, 1), w.u_series.astype(np.int32).reshape(B, 1)]
    res = _call(driver, "batch", args)
    ref = H.oracle_batch(w)
    assert len(res) == 11
    for name, got in zip(ORDER11, res):
        want = np.transpose(ref[name], (2, 1, 0)) if ref[name].ndim == 3 else ref[name].T
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), name


@pytest.mark.gpu
def test_rt_and_simulator_gateways(gpu_device, driver):
    """epiekf_rt_mex (Tools/Rt_ExpFitEKF.m) and epiekf_sim_mex (SIalpha_Controlled, SEIRP, NPICost, SI_Controlled) against
    the Tools/-named Python host mirror, which the parity tests hold bit for bit to the oracle."""
    from epidemicmodeling_amd import batch, tools
    rng = np.random.default_rng(3)
    T = 60
    x = np.abs(rng.standard_normal((1, T))) * 50 + 10
    s_init, prm3 = [30.0, 0.01], [1.0, 0.9, 0.1]
    ref = tools.Rt_ExpFitEKF(x, s_init, prm3, [0, 0], 0.0, np.eye(2), np.diag([1e-1, 1e-4]), 25.0, 0.9, 0.995, 21, 2, device=gpu_device)
    rp = np.zeros((19, 1))                               # the column matlab/Tools/Rt_ExpFitEKF.m builds (EPI_RT_* rows)
    rp[0:3, 0] = prm3; rp[3:5, 0] = [0, 0]; rp[5] = 0.0; rp[6] = 25.0; rp[7], rp[8] = 0.9, 0.995
    rp[9:11, 0] = s_init
    rp[11:15, 0] = np.eye(2).reshape(-1, order="F"); rp[15:19, 0] = np.diag([1e-1, 1e-4]).reshape(-1, order="F")
    res = _call(driver, "rt", [x, rp, 21.0, 2.0])
    assert len(res) == 9
    for got, want in zip(res, ref):
        assert got.shape == np.asarray(want).shape and np.array_equal(got, want, equal_nan=True)
    # SIalpha_Controlled
    n, K = 12, 50
    u = rng.integers(0, 4, size=(n, K)).astype(np.float64)
    a = rng.random(n) * 0.03; um = synth.IP_MAXES.astype(np.float64)
    sp = np.zeros((batch.SIM_PRM_COUNT, 1))
    vals = dict(s0=0.999, i0=1e-3, alpha0=1.1, alpha_min=1e-8, alpha_max=100.0, gamma=1 / 7, b=0.01, beta=synth.MODEL_BETA,
                s_noise_std=1e-5, i_noise_std=3e-5, alpha_noise_std=1e-2, dt=1.0)
    for k, v in vals.items():
        sp[batch.SIM_FIELDS[k], 0] = v
    sp[batch.SIM_A:batch.SIM_A + n, 0] = a; sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n, 0] = um
    z = rng.standard_normal((3, K))
    s, i, al = tools.SIalpha_Controlled(u, vals["s0"], vals["i0"], vals["alpha0"], um, vals["alpha_min"], vals["alpha_max"], vals["gamma"],
                                        a, vals["b"], vals["beta"], vals["s_noise_std"], vals["i_noise_std"], vals["alpha_noise_std"], K, vals["dt"],
                                        noise=z, device=gpu_device)
    res = _call(driver, "sim", ["sialpha", u, sp, z], nlhs=3)
    assert len(res) == 3 and all(r.shape == (1, K) for r in res)
    assert np.array_equal(res[0], s) and np.array_equal(res[1], i) and np.array_equal(res[2], al)
    # NPICost
    wts = rng.random((n, K)); nc = (s * i * al).reshape(1, K)
    J0, J1 = tools.NPICost(nc, u, wts, device=gpu_device)
    res = _call(driver, "sim", ["npicost", nc, u, wts], nlhs=2)
    assert (float(res[0][0, 0]), float(res[1][0, 0])) == (J0, J1)
    # SI_Controlled
    alpha = rng.random(K) * 0.5
    s2, i2 = tools.SI_Controlled(alpha, 0.05, 0.99, 0.01, K, 0.1, device=gpu_device)
    res = _call(driver, "sim", ["si", alpha.reshape(1, K), np.array([[0.05], [0.99], [0.01]]), float(K), 0.1], nlhs=2)
    assert np.array_equal(res[0], np.asarray(s2).reshape(1, K)) and np.array_equal(res[1], np.asarray(i2).reshape(1, K))
    _call(driver, "sim", ["nonsense"], expect_error="unknown command")


def _batch3(driver, w, tmp=None):
    """SIAlphaModelEKF for all regions of workload `w` through epiekf_batch_mex (region index first); returns S_SMOOTH [T, 3, S]."""
    S, T = w.B, w.T
    args = [0.0, np.transpose(w.u, (2, 1, 0)), w.x.T, w.prm.T, w.s_init.T, w.Ps_init.T, w.s_final.T, w.Ps_final.T, w.Q.T,
            w.R_series.T, float(w.L), float(w.order), 0.0, np.zeros((0, 1), dtype=np.int32), np.zeros((0, 1), dtype=np.int32)]
    res = _call(driver, "batch", args, tmp=tmp)
    return np.transpose(res[ORDER11.index("S_SMOOTH")], (2, 1, 0))


@pytest.mark.gpu
def test_prescription_pipeline_through_the_gateways_only(gpu_device, driver, tmp_path):
    """Tools/TrainPredictPrescribeNPI.m's device stages chained through the MEX gateways alone -- no Python device API in
    the chain: preprocess -> SIAlphaModelEKF (zero input) -> regression -> SIAlphaModelEKF (real inputs) -> regression ->
    forecast filter -> the cost-weight sweep of all regions with scoring and Pareto front in ONE call -> random-NPI
    Monte-Carlo.  Every stage's result equals what epidemicmodeling_amd.pipeline.prescribe (the torch-side chain, itself
    held to the oracle stage by stage in tests/test_gpu_parity.py) computes, bit for bit."""
    from epidemicmodeling_amd import batch, pipeline
    tmp = str(tmp_path)
    S, T, n, H_, P = 5, 90, 12, 20, 30
    d = synth.make_raw_counts(S, T, seed=11)
    d["cases"][:, -1] = np.cumsum(np.full(T, 40.0))              # the all-NaN region of the generator: give it data
    ref = pipeline.prescribe(d["cases"], d["deaths"], d["population"], d["ip"], horizon=H_, n_eps=P, num_regression_days=60,
                             device=gpu_device)
    N = np.asarray(d["population"], dtype=np.float64)
    # 1. preprocessing
    pre = _call(driver, "pipeline", ["preprocess", d["cases"].T, d["deaths"].T, N.reshape(S, 1), np.transpose(d["ip"], (2, 1, 0)), 7.0, 7.0,
                                     float(synth.MIN_CASES)], tmp=tmp)
    names = ["new_refined", "new_smoothed", "zero_lag", "x_new", "x_total", "R_v", "fatality", "I0", "ip_filled"]
    pre = dict(zip(names, pre))
    for k in names:
        want = ref["pre"][k]
        got = pre[k].T if pre[k].ndim == 2 and k != "I0" else (pre[k].reshape(-1) if k == "I0" else np.transpose(pre[k], (2, 1, 0)))
        assert np.array_equal(got, want, equal_nan=True), k
    x, R, u, I0 = pre["x_new"].T, pre["R_v"].T, np.transpose(pre["ip_filled"], (2, 1, 0)), pre["I0"].reshape(-1)
    # 2. round 1 (zero input) and the first regression
    S1 = _batch3(driver, pipeline.workload3(x, R, np.zeros_like(u), N, I0, np.zeros((n, S)), np.zeros(S)), tmp)
    assert np.array_equal(S1[:, 2], ref["alpha_round1"])
    D = 60
    X = np.ascontiguousarray(synth.IP_MAXES[None, :n, None] - u[T - D:])
    a1, b1, _, _ = _call(driver, "pipeline", ["nnls", np.transpose(X, (2, 1, 0)), S1[T - D:, 2].T, 100.0], nlhs=4, tmp=tmp)
    assert np.array_equal(a1.T, ref["fit1"]["a"]) and np.array_equal(b1.reshape(-1), ref["fit1"]["b"])
    # 3. round 2 and the second regression
    S2 = _batch3(driver, pipeline.workload3(x, R, u, N, I0, a1.T, b1.reshape(-1)), tmp)
    a2, b2, me2, it2 = _call(driver, "pipeline", ["nnls", np.transpose(X, (2, 1, 0)), S2[T - D:, 2].T, 100.0], nlhs=4, tmp=tmp)
    assert np.array_equal(a2.T, ref["fit2"]["a"]) and np.array_equal(b2.reshape(-1), ref["fit2"]["b"])
    assert np.array_equal(me2.reshape(-1), ref["fit2"]["min_err"]) and np.array_equal(it2.reshape(-1), ref["fit2"]["iters"])
    a2, b2 = a2.T, b2.reshape(-1)
    # 4. forecast filter (last plan held over the horizon)
    R_mean = R.sum(axis=0) / T
    xh = np.concatenate([x, np.full((H_, S), np.nan)]); Rh = np.concatenate([R, np.repeat(R_mean[None], H_, 0)])
    Sf = _batch3(driver, pipeline.workload3(xh, Rh, np.concatenate([u, np.repeat(u[-1:], H_, 0)]), N, I0, a2, b2), tmp)
    hist = Sf[:T]
    assert np.array_equal(hist, ref["historic"])
    # 5. the sweep of all regions in one call: per-region inputs only
    reg = pipeline.sweep_region_inputs(N, I0, a2, b2, n)
    wts = np.ones((n, S))
    sp = pipeline.scoring_region_inputs(hist[T - 1], a2, b2, synth.IP_MAXES[:n], wts)
    J0p = np.cumsum(hist[:, 0] * hist[:, 1] * hist[:, 2], axis=0)[-1]
    J1p = np.cumsum((wts[None] * u).reshape(T * n, S), axis=0)[-1]
    eps = synth.epsilon_grid(P)
    u_nan = np.concatenate([u, np.full((H_, n, S), np.nan)])
    res = _call(driver, "pipeline", ["prescribe", xh.T, np.transpose(u_nan, (2, 1, 0)), Rh.T, reg["prm"].T, reg["s_init"].T, reg["Ps_init"].T,
                                     reg["s_final"].T, reg["Ps_final"].T, reg["Q"].T, eps.reshape(-1, 1), sp.T, J0p.reshape(-1, 1),
                                     J1p.reshape(-1, 1), float(T), 21.0, 1.0, 0.0, np.zeros((0, 0))], tmp=tmp)
    J0, J1, on_front, I_opt, u_opt, S_opt = res
    assert J0.shape == (P, S) and u_opt.shape == (S, n, T + H_) and S_opt.shape == (S, 6, T + H_)
    assert np.array_equal(J0.T, ref["J0"]) and np.array_equal(J1.T, ref["J1"])
    assert np.array_equal(on_front.T.astype(bool), ref["front"])
    assert np.array_equal(I_opt.reshape(-1) - 1, ref["i_opt"])                     # the gateway returns MATLAB's 1-based index
    assert np.array_equal(np.transpose(u_opt, (2, 1, 0))[T:], ref["prescription"])
    # 6. random-NPI Monte-Carlo scenarios of every region
    nz = 40
    mcr = batch.random_npi_mc(sp, np.zeros((n, S)), nz, H_, seed=5, J0_prefix=J0p, J1_prefix=J1p, prefix_days=T, device=gpu_device)
    m0, m1 = _call(driver, "pipeline", ["mc", sp.T, np.zeros((S, n)), float(nz), float(H_), 5.0, np.zeros((0, 0)), J0p.reshape(-1, 1),
                                        J1p.reshape(-1, 1), float(T)], nlhs=2, tmp=tmp)
    assert np.array_equal(m0.T, mcr["J0"].cpu().numpy()) and np.array_equal(m1.T, mcr["J1"].cpu().numpy())
    _call(driver, "pipeline", ["nonsense"], expect_error="unknown command", tmp=tmp)
