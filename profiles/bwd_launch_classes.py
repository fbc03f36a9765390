"""Splits the eks_bwd launches of a rocprofv3 kernel trace of bench.py into the three kinds the run contains:
  full     -- the whole recursion in one launch (the staged passes the per-kernel HIP events and `roofline` are measured on)
  horizon  -- the first of the two launches of a timed pass (smoother steps T-2 .. t_hist: the horizon days)
  observed -- the second one (steps t_hist-1 .. 0), beside which the scoring tail and the Pareto filter run
rocprofv3 --stats averages all three under one kernel name; this script separates them by duration (they differ by > 2 x).
    python profiles/bwd_launch_classes.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import sys

import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "eks_bwd" in r["Kernel_Name"]])
full_ms = d.max()
cls = {"full": d[d > 0.9 * full_ms], "observed": d[(d <= 0.9 * full_ms) & (d > 0.5 * full_ms)], "horizon": d[d <= 0.5 * full_ms]}
res = {k: {"launches": int(v.size), "mean_ms": float(v.mean()) if v.size else None, "min_ms": float(v.min()) if v.size else None,
           "max_ms": float(v.max()) if v.size else None} for k, v in cls.items()}
res["all_launches_mean_ms_as_rocprof_stats_reports"] = float(d.mean())
res["horizon_plus_observed_ms"] = (res["horizon"]["mean_ms"] or 0.0) + (res["observed"]["mean_ms"] or 0.0)
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
