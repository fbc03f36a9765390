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
out_mb = sum(v.nbytes for v in got.values()) / 1e6
rep["sweep_prescribe_host_300x250x520"] = {
    "ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts), "host_to_device_MB": in_mb, "device_to_host_MB": out_mb,
    "region_day_steps": S * P * (T_hist + hor), "steps_per_s": S * P * (T_hist + hor) / float(np.median(ts)),
    "front_points_mean": float(got["on_front"].sum(axis=1).mean()),
    "note": "per-region inputs expanded on the device; filter + scoring + Pareto filter + gather of the optimum's plan; python packing of the ctypes call included"}
print(json.dumps(rep["sweep_prescribe_host_300x250x520"]), flush=True)

full = synth.make_cfg4(75, 125, 400, 120)                         # 9 375 chains
H.host_call(full, devices=[0], outputs=["u_opt_smooth", "S_SMOOTH"], extras=False)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); H.host_call(full, devices=[0], outputs=["u_opt_smooth", "S_SMOOTH"], extras=False); ts.append(time.perf_counter() - t0)
rep["run_host_multi_9375_chains_reduced_outputs"] = {"ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts),
                                                     "device_to_host_MB": 9375 * 520 * 18 * 8 / 1e6,
                                                     "note": "epi_ekf_run_host_multi, one block on device 0; 702 MB of selected outputs over PCIe into pageable memory"}
print(json.dumps(rep["run_host_multi_9375_chains_reduced_outputs"]), flush=True)

two = synth.make_cfg4(2, 250, 400, 120)
for tag, w in (("B1", two.select(np.array([137]))), ("B250", two.select(np.arange(250)))):
    H.host_call(w, extras=False)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); H.host_call(w, extras=False); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); H.oracle_batch(w, n_threads=1); t_cpu = time.perf_counter() - t0
    rep["run_host_" + tag] = {"chains": w.B, "days": w.T, "gpu_call_ms_median": 1e3 * float(np.median(ts)), "gpu_call_ms_min": 1e3 * min(ts),
                              "cpu_oracle_one_thread_ms": 1e3 * t_cpu, "note": "all 11 outputs; python packing of the ctypes call included (~0.1 ms)"}
    print(json.dumps(rep["run_host_" + tag]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "host_calls.json"), "w"), indent=1)
