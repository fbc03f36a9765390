"""Quick parity + timing check of the four-lanes-per-chain kernels against the oracle (development aid)."""
import sys, time
import numpy as np
import torch
from epidemicmodeling_amd import batch, synth
from tests import helpers as H

def check(w, tag, **kw):
    ref = H.oracle_batch(w)
    got = batch.run_workload(w, device="cuda:0", shape="quad", **kw)
    bad = [n for n in H.OUT_NAMES + ["pinv_rank"] if n in got and not np.array_equal(got[n], ref[n], equal_nan=True)]
    worst = {n: H.rel_err(got[n], ref[n]) for n in bad if n != "pinv_rank"}
    print(tag, "OK" if not bad else f"MISMATCH {worst}", flush=True)
    return not bad

ok = True
ok &= check(synth.make_cfg4(3, 7, 40, 12), "cfg4 small")
ok &= check(synth.make_cfg4(5, 13, 60, 20), "cfg4 blocked16", lane_block=16)
ok &= check(synth.make_cfg4(5, 13, 60, 20), "cfg4 blocked8", lane_block=8)
ok &= check(synth.as_backward(synth.make_cfg4(4, 5, 40, 0)), "sia6 backward")
ok &= check(synth.make_row3(3, 6), "row3 adaptive R")
ok &= check(synth.make_cfg4(12, 50, 200, 60), "cfg4 600x260", lane_block="auto")
sys.exit(0 if ok else 1)
