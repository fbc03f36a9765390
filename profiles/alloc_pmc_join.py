"""Joins rocprofv3's counter_collection.csv and kernel_trace.csv of a profiles/alloc_probe.py run: per dispatch of the
forward kernel / smoother its duration and the counters, sorted by duration.  python profiles/alloc_pmc_join.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for key in ("ekf_fwd_sym", "eks_bwd_sym"):
    rows = [(dur[i][1], cnt[i]) for i in cnt if i in dur and key in dur[i][0]]
    rows.sort(key=lambda x: x[0])
    print(key, len(rows), "dispatches")
    for t, c in rows:
        print("  %7.3f ms  " % t + "  ".join("%s=%.4g" % (k.replace("_sum", ""), v) for k, v in sorted(c.items())))
