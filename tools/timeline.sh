#!/bin/bash
# Kernel timeline of one timed pass of the sweep at a given shard size:  tools/timeline.sh REGIONS EPS [extra bench args]
# (rocprofv3 --kernel-trace; prints start / duration / queue of each kernel of the second pass)
set -e
R=${1:-75}; E=${2:-125}; shift 2 || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_${R}x${E}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --regions $R --eps $E "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<PY
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "calib" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sims = [i for i, r in enumerate(rows) if "sialpha_sim" in r["Kernel_Name"]]
lo, hi = sims[1] + 1, sims[2] + 3          # the second pass (warm-up is the first), scoring tail included
while "ekf_fwd" not in rows[lo]["Kernel_Name"]: lo += 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print("%-46s start %7.3f dur %6.3f ms q%s" % (r["Kernel_Name"][:46], (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "")))
PY
