#!/bin/bash
# SQ counters of the kernels of one probe workload, stage by stage, in three passes:
#   tools/sq_shard.sh TAG WAVES DAYS [traffic_probe.py args]      (EPIEKF_LIB selects another build)
# prints every counter per launch and per wave-day (WAVES x DAYS); output gpurun_out/r05/sq_TAG.txt
TAG=$1; WAVES=$2; DAYS=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/sq_$TAG; mkdir -p $O
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
C="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
for P in A B C; do
  rocprofv3 --pmc ${!P} --kernel-trace --output-format csv -d $O/$P -o p -- python3 $R/profiles/traffic_probe.py "$@" > $O/$P.log 2>&1 && echo pass $P ok
done
cd $R
python3 - "$O" "$WAVES" "$DAYS" > $R/gpurun_out/r05/sq_$TAG.txt <<'PY'
import csv, glob, collections, sys
O, W, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "epi::" not in k or "calib" in k: continue
        k = k.replace("void ", "").split("(")[0].replace("epi::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for n in sorted(v):
        m = sum(v[n]) / len(v[n])
        print("   %-26s %14.0f   per wave-day (%d waves x %d days) %10.1f" % (n, m, W, D, m / (W * D)))
PY
rm -rf $O
cat $R/gpurun_out/r05/sq_$TAG.txt
