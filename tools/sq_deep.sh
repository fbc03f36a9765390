# SQ counters of the 9 375-chain shard's kernels, stage by stage (profiles/traffic_probe.py shard), in three passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/sq_deep; mkdir -p $O
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SMEM"
C="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES"
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/profiles/traffic_probe.py shard > /dev/null 2>&1 && echo a
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/profiles/traffic_probe.py shard > /dev/null 2>&1 && echo b
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/c -o p -- python3 $R/profiles/traffic_probe.py shard > /dev/null 2>&1 && echo c
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03/sq_deep/*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "epi::" not in k or "calib" in k: continue
        k = k.replace("void ", "").split("(")[0].replace("epi::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for n in sorted(v):
        m = sum(v[n]) / len(v[n])
        print("   %-26s %14.0f   per wave-day (586 waves x 520 days) %10.1f" % (n, m, m / (586 * 520)))
PY
