"""Quick parity check of the two lane mappings and the launch modes against the oracle (GPU box; development aid):
    PYTHONPATH=. python tools/shape_check.py"""
import sys
import numpy as np
from epidemicmodeling_amd import batch, synth
from tests import helpers as H

def check(w, tag, **kw):
    ref = H.oracle_batch(w)
    got = batch.run_workload(w, device="cuda:0", **kw)
    bad = [n for n in H.OUT_NAMES + ["pinv_rank"] if n in got and not np.array_equal(got[n], ref[n], equal_nan=True)]
    worst = {n: H.rel_err(got[n], ref[n]) for n in bad if n != "pinv_rank"}
    print(tag, kw, "OK" if not bad else f"MISMATCH {worst}", flush=True)
    return not bad

ok = True
for shape in ("quad", "lane"):
    for tp in (0, 1, -1):
        ok &= check(synth.make_cfg4(3, 7, 140, 12), "cfg4 small", shape=shape, time_pipe=tp)
        ok &= check(synth.make_cfg4(5, 13, 160, 20), "cfg4 blocked", shape=shape, time_pipe=tp, lane_block="auto")
        ok &= check(synth.as_backward(synth.make_cfg4(4, 5, 40, 0)), "sia6 backward", shape=shape, time_pipe=tp)
        ok &= check(synth.as_backward(synth.make_cfg3(6, 160)), "sia3 backward, long", shape=shape, time_pipe=tp)
        ok &= check(synth.make_cfg3(9, 150), "cfg3", shape=shape, time_pipe=tp)
ok &= check(synth.make_row3(3, 6), "row3 adaptive R", shape="quad")
ok &= check(synth.make_cfg4(12, 50, 200, 60), "cfg4 600x260", shape="quad", lane_block="auto")
sys.exit(0 if ok else 1)
