"""Which launch of the innovation monitor is the slow one, and what ran beside it?  (verdict r05, weak #4: ekf_monitor<0,21> min 0.66 /
max 9.45 ms in profiles/r05/bench_cfg4_staged_kernel_stats.csv)

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o p -- python3 bench.py --no-cpu-baseline   (EPI_BENCH_STAGED=1)
    python3 profiles/monitor_outlier.py OUT

Prints every monitor launch (start relative to the first kernel, duration) and, for the three longest, the kernels whose execution
overlaps it on the device."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t0 = rows[0]["s"]
name = lambda r: r["Kernel_Name"].replace("void ", "").replace("epi::", "").split("(")[0][:44]
mon = [r for r in rows if "ekf_monitor" in r["Kernel_Name"]]
print("%d monitor launches; durations ms: min %.3f median %.3f max %.3f" % (
    len(mon), min(r["e"] - r["s"] for r in mon) / 1e6, sorted(r["e"] - r["s"] for r in mon)[len(mon) // 2] / 1e6, max(r["e"] - r["s"] for r in mon) / 1e6))
for i, r in enumerate(mon):
    print("  #%02d at %9.3f ms  dur %7.3f ms  queue %s" % (i, (r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r.get("Queue_Id", "")))
for r in sorted(mon, key=lambda r: r["s"] - r["e"])[:3]:
    print("\nlaunch at %.3f ms, %.3f ms long; on the device during it:" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6))
    for o in rows:
        if o is r or o["e"] <= r["s"] or o["s"] >= r["e"]:
            continue
        print("    %-44s %9.3f .. %9.3f ms (dur %7.3f) queue %s" % (name(o), (o["s"] - t0) / 1e6, (o["e"] - t0) / 1e6, (o["e"] - o["s"]) / 1e6, o.get("Queue_Id", "")))
    prev = [o for o in rows if o["e"] <= r["s"]][-3:]
    print("    before it: " + "; ".join("%s (ended %.3f ms earlier)" % (name(o), (r["s"] - o["e"]) / 1e6) for o in prev))
