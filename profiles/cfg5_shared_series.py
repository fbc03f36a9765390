"""BASELINE config 5's forward kernel waits 57 % of its cycles on memory although its store pattern alone runs at 6.6 TB/s.
Are the per-chain observation series (x, R_v: [T][B], read from HBM under the kernel's own write load) what it waits for?
Times the stages of config 5 as it is and with the SAME chains reading their x / R_v from 300 shared series (L2 hits), as the
headline sweep does.  Timing only -- the shared variant is another filtering problem.
    python profiles/cfg5_shared_series.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicmodeling_amd import batch, synth  # noqa: E402


def stage_times(r, reps=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f, p, b = [], [], []
    for rep in range(reps + 1):
        ev[0].record(); r.run(phase=1); ev[1].record(); r.run(phase=3); ev[2].record(); r.run(phase=4); ev[3].record()
        torch.cuda.synchronize()
        if rep:
            f.append(ev[0].elapsed_time(ev[1])); p.append(ev[1].elapsed_time(ev[2])); b.append(ev[2].elapsed_time(ev[3]))
    return round(float(np.median(f)), 2), round(float(np.median(p)), 2), round(float(np.median(b)), 2)


dev = torch.device("cuda:0")
w = synth.make_cfg5(300, 1024, 400)
r = batch.EkfRunner(batch.DeviceWorkload(w, dev), lane_block="auto", storage="f32")
print("config 5 as it is (x, R_v per chain)        fwd / pinv / bwd ms:", stage_times(r), flush=True)
del r
torch.cuda.empty_cache()
reg = np.repeat(np.arange(300, dtype=np.int32), 1024)
w.x = np.ascontiguousarray(w.x[:, ::1024]); w.R_series = np.ascontiguousarray(w.R_series[:, ::1024]); w.x_series = reg
r = batch.EkfRunner(batch.DeviceWorkload(w, dev), lane_block="auto", storage="f32")
print("the same chains on 300 shared x / R_v series  fwd / pinv / bwd ms:", stage_times(r), flush=True)
