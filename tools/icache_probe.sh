#!/bin/bash
# instruction-cache counters of the pass's kernels, time-pipelined or not:  tools/icache_probe.sh TAG [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/ic_$TAG; mkdir -p $O
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O.log 2>&1
cd $R
python3 - "$O" > $R/gpurun_out/r05/ic_$TAG.txt <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "epi::" not in k or "calib" in k: continue
        k = k.replace("void ", "").split("(")[0].replace("epi::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "launches", len(next(iter(v.values()))))
    for n in sorted(v):
        print("   %-22s mean %14.0f  max %14.0f" % (n, sum(v[n]) / len(v[n]), max(v[n])))
PY
rm -rf $O
cat $R/gpurun_out/r05/ic_$TAG.txt
