"""A/B helper: median HIP-event time of one kernel stage of the headline sweep for the library named by EPIEKF_LIB.
    EPIEKF_LIB=$PWD/ab/variant.so python profiles/ab_phase.py 4 [reps]      # 1 = forward, 3 = pinv, 4 = backward"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth  # noqa: E402

phase = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
r = batch.EkfRunner(batch.DeviceWorkload(synth.make_cfg4(), "cuda:0"), lane_block=(lambda v: "auto" if v == "auto" else int(v))(os.environ.get("EPI_LANE_BLOCK", "auto")))
for ph in (1, 3, 4):
    r.run(phase=ph)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r.run(phase=phase); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(os.path.basename(os.environ.get("EPIEKF_LIB", "default")), "phase", phase, "median %.3f min %.3f max %.3f ms" % (np.median(ts), min(ts), max(ts)))
