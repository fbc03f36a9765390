"""Per-stage time of small batches in each lane mapping (HIP events, median of 9; stages enqueued one by one) and the wall
time of the host-pointer call:   python profiles/shape_latency.py [out.json]
B = 1 (one reference-shaped call, TrainPredictPrescribeNPI.m:460), 16, 250 (a region's cost weights), 300, 1024, 2048, and
(round 5) 4 096, 9 375 with the hex shape; then one row per model x R-mode at B = 1 / 16 / 250 (verdict r04 item 4): the
6-state sweep with a per-day R_v, the same with a scalar adaptive R_v (testPrescribeXPRIZE01.m:211), NewCaseEKFEstimatorWithOptimalNPI."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth  # noqa: E402
from tests import helpers as H  # noqa: E402

full = synth.make_cfg4(9, 250, 400, 120)
out = {}
def measure(w, shapes):
    row = {}
    for shape in shapes:
        r = batch.EkfRunner(batch.DeviceWorkload(w, "cuda:0"), lane_block="auto", shape=shape)
        r.run(); torch.cuda.synchronize()
        ms = {}
        for name, ph in (("ekf_fwd", 1), ("eks_pinv", 3), ("eks_bwd", 4), ("whole_call", 0)):
            ts = []
            for _ in range(9):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); r.run(phase=ph); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ms[name] = round(float(np.median(ts)), 4)
        ts = []
        sh = {"wave": 3, "quad": 2, "lane": 1, "hex": 4, "auto": 0}[shape]
        H.host_call(w, extras=False, shape=sh)
        for _ in range(9):
            t0 = time.perf_counter(); H.host_call(w, extras=False, shape=sh); ts.append(time.perf_counter() - t0)
        ms["host_call_all_outputs_wall"] = round(1e3 * float(np.median(ts)), 4)
        ms["lane_block_chosen"] = r.blk
        row[shape] = ms
        del r
    if w.B <= 2048:
        t0 = time.perf_counter(); H.oracle_batch(w, n_threads=1); row["cpu_oracle_one_thread_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
    return row


for B in (1, 16, 250, 300, 1024, 2048, 4096, 9375):
    src = full if B <= full.B else synth.make_cfg4(75, 125, 400, 120)
    w = src.select(np.linspace(0, src.B - 1, B).astype(np.int64)) if B > 1 else src.select(np.array([137]))
    row = measure(w, ("wave", "hex", "quad", "lane") if B <= 2048 else ("hex", "quad", "lane"))
    out[f"B={B}, T=520"] = row
    print(B, json.dumps(row), flush=True)
# one row per model x R-mode (shape auto = what an unchanged caller gets)
adapt = synth.make_row3(9, 250, 400, 120) if hasattr(synth, "make_row3") else None
newc = synth.make_newcase_sweep(9, 250, 400, 120)
for tag, src in (("SIAlphaModelEKFOptControlled, scalar adaptive R_v (testPrescribeXPRIZE01.m:211)", adapt),
                 ("NewCaseEKFEstimatorWithOptimalNPI", newc)):
    if src is None:
        continue
    for B in (1, 16, 250):
        w = src.select(np.linspace(0, src.B - 1, B).astype(np.int64)) if B > 1 else src.select(np.array([137]))
        row = measure(w, ("auto", "quad") if "adaptive" in tag else ("auto",))
        out[f"{tag}: B={B}, T={w.T}"] = row
        print(tag, B, json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
