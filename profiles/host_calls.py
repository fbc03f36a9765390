"""Wall time of the HOST-pointer entry points (what a MATLAB host pays), written to gpurun_out/host_calls.json:
  * epi_sweep_prescribe_host: the headline sweep (300 regions x 250 cost weights x (400 + 120) days) from per-region host
    arrays to (J0, J1), front, I_opt and the optimum's plan -- the whole of TrainPredictPrescribeNPI.m:421-493, 624-633;
  * epi_ekf_run_host_multi: one 9 375-chain block (the shard of the sweep one of 8 GPUs runs), reduced outputs;
  * epi_ekf_run_host: ONE reference-shaped call (B = 1) and the 250 cost weights of one region (B = 250), all outputs.
python profiles/host_calls.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicmodeling_amd import hostapi, pipeline, synth  # noqa: E402
from tests import helpers as H  # noqa: E402

rep = {}
S, P, T_hist, hor = 300, 250, 400, 120
w0 = synth.make_cfg4(S, 1, T_hist, hor)
rng = np.random.default_rng(0)
reg_tab = synth.make_regions(S)
N = np.asarray(reg_tab["N"], dtype=np.float64)
I0 = np.full(S, 100.0)
a, b = np.ascontiguousarray(reg_tab["a"].T), reg_tab["b"]
eps = synth.epsilon_grid(P)
reg = pipeline.sweep_region_inputs(N, I0, a, b, 12)
sp = pipeline.scoring_region_inputs(np.stack([1.0 - 100.0 / N, 100.0 / N, np.full(S, synth.ALPHA0)]), a, b, synth.IP_MAXES[:12], np.ones((12, S)))
J0p, J1p = rng.random(S) * 1e-2, rng.random(S) * 40.0
args = (w0.x, w0.u, w0.R_series, reg, eps, sp, J0p, J1p, T_hist)
hostapi.sweep_prescribe(*args)                                   # first call: contexts, arena
ts = []
for _ in range(5):
    t0 = time.perf_counter(); got = hostapi.sweep_prescribe(*args); ts.append(time.perf_counter() - t0)
in_mb = sum(np.asarray(v).nbytes for v in (w0.x, w0.u, w0.R_series, sp, *reg.values())) / 1e6
out_mb = sum(v.nbytes for v in got.values() if hasattr(v, "nbytes")) / 1e6
rep["sweep_prescribe_host_300x250x520"] = {
    "ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts), "host_to_device_MB": in_mb, "device_to_host_MB": out_mb,
    "region_day_steps": S * P * (T_hist + hor), "steps_per_s": S * P * (T_hist + hor) / float(np.median(ts)),
    "front_points_mean": float(got["on_front"].sum(axis=1).mean()),
    "note": "per-region inputs expanded on the device; filter + scoring + Pareto filter + gather of the optimum's plan; python packing of the ctypes call included"}
print(json.dumps(rep["sweep_prescribe_host_300x250x520"]), flush=True)

def timed(w, n, **kw):
    """median / min ms of the C call alone (output arrays of the first call written again) and of the python call that
    allocates and NaN-fills fresh output arrays every time (what a MEX gateway's mxCreateDoubleMatrix costs as well)"""
    out = H.host_call(w, **kw)
    tc, tw = [], []
    for _ in range(n):
        H.host_call(w, out=out, timing=tc, **kw)
    for _ in range(max(3, n // 2)):
        t0 = time.perf_counter(); H.host_call(w, **kw); tw.append(time.perf_counter() - t0)
    mb = sum(v.nbytes for v in out.values()) / 1e6
    return {"c_call_ms_median": 1e3 * float(np.median(tc)), "c_call_ms_min": 1e3 * min(tc),
            "with_fresh_output_arrays_ms_median": 1e3 * float(np.median(tw)), "device_to_host_MB": mb,
            "c_call_GB_per_s_of_outputs": mb / 1e3 / float(np.median(tc))}


full = synth.make_cfg4(75, 125, 400, 120)                         # 9 375 chains
for tag, devs in (("one_block", [0]), ("two_blocks_strided", [0, 0])):
    r = timed(full, 5, devices=devs, outputs=["u_opt_smooth", "S_SMOOTH"], extras=False)
    r["note"] = ("epi_ekf_run_host_multi, 9 375 chains (the shard of the sweep one of 8 GPUs runs), reduced outputs into pageable memory; "
                 + ("one block on device 0" if len(devs) == 1 else "two chain blocks, both on device 0, one after the other: every array is moved as a strided piece"))
    rep["run_host_multi_9375_chains_reduced_outputs_" + tag] = r
    print(json.dumps(r), flush=True)

# all 11 outputs of the 9 375-chain shard (6.2 GB back over PCIe): into arrays the previous call already wrote (resident pages) and into
# fresh, never-touched ones (np.empty: the pages are faulted in under the copy -- what a MEX gateway's freshly created outputs cost)
def timed_cold(w, n, **kw):
    names = [k for k in H.OUT_NAMES]
    m, n_npi, B, T = w.m, w.n_npi, w.B, w.T
    rows = {"u_opt": n_npi, "u_opt_smooth": n_npi, "S_MINUS": m, "S_PLUS": m, "S_SMOOTH": m, "P_MINUS": m * m, "P_PLUS": m * m, "P_SMOOTH": m * m, "K_GAIN": m}
    tc = []
    for _ in range(n):
        out = {k: np.empty((T, rows[k], B) if k in rows else (T, B)) for k in names}
        H.host_call(w, out=out, timing=tc, extras=False, **kw)
        mb = sum(v.nbytes for v in out.values()) / 1e6
        del out
    return {"c_call_ms_median": 1e3 * float(np.median(tc)), "c_call_ms_min": 1e3 * min(tc), "device_to_host_MB": mb,
            "c_call_GB_per_s_of_outputs": mb / 1e3 / float(np.median(tc))}


r = timed(full, 4, extras=False)
r["note"] = "epi_ekf_run_host, 9 375 chains, ALL 11 outputs, output arrays of the previous call written again (resident pages)"
rep["run_host_9375_chains_all_outputs_reused_arrays"] = r
print(json.dumps(r), flush=True)
r = timed_cold(full, 3)
r["note"] = "the same into fresh never-touched arrays (np.empty): page faults under the copy"
rep["run_host_9375_chains_all_outputs_fresh_arrays"] = r
print(json.dumps(r), flush=True)

two = synth.make_cfg4(2, 250, 400, 120)
for tag, w in (("B1", two.select(np.array([137]))), ("B250", two.select(np.arange(250)))):
    r = timed(w, 9, extras=False)
    t0 = time.perf_counter(); H.oracle_batch(w, n_threads=1); t_cpu = time.perf_counter() - t0
    r.update({"chains": w.B, "days": w.T, "cpu_oracle_one_thread_ms": 1e3 * t_cpu, "note": "epi_ekf_run_host, all 11 outputs"})
    rep["run_host_" + tag] = r
    print(json.dumps(r), flush=True)
    if tag == "B250":
        r = timed_cold(w, 7)
        r["note"] = "B = 250, all 11 outputs into fresh never-touched arrays"
        rep["run_host_B250_fresh_arrays"] = r
        print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "host_calls.json"), "w"), indent=1)
