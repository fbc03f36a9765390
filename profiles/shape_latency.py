"""Per-stage time of small batches in each lane mapping (HIP events, median of 9; stages enqueued one by one) and the wall
time of the host-pointer call:   python profiles/shape_latency.py [out.json]
B = 1 (one reference-shaped call, TrainPredictPrescribeNPI.m:460), 16, 250 (a region's cost weights), 300, 1024, 2048."""
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
for B in (1, 16, 250, 300, 1024, 2048):
    w = full.select(np.linspace(0, full.B - 1, B).astype(np.int64)) if B > 1 else full.select(np.array([137]))
    row = {}
    for shape in ("wave", "quad", "lane"):
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
        sh = {"wave": 3, "quad": 2, "lane": 1}[shape]
        H.host_call(w, extras=False, shape=sh)
        for _ in range(9):
            t0 = time.perf_counter(); H.host_call(w, extras=False, shape=sh); ts.append(time.perf_counter() - t0)
        ms["host_call_all_outputs_wall"] = round(1e3 * float(np.median(ts)), 4)
        row[shape] = ms
        del r
    t0 = time.perf_counter(); H.oracle_batch(w, n_threads=1); row["cpu_oracle_one_thread_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
    out[f"B={B}, T=520"] = row
    print(B, json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
