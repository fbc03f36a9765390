cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_cfg5 -o p -- python3 $R/profiles/traffic_probe.py cfg5 > /dev/null 2>&1 && echo sq cfg5
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $O/pmc_sq2_cfg5 -o p -- python3 $R/profiles/traffic_probe.py cfg5 > /dev/null 2>&1 && echo sq2 cfg5
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_cfg5 -o p -- python3 $R/profiles/traffic_probe.py cfg5 > /dev/null 2>&1 && echo fetch
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_cfg5 -o p -- python3 $R/profiles/traffic_probe.py cfg5 > /dev/null 2>&1 && echo write
cd $R
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
python3 profiles/valu_summary.py $(f $O/pmc_sq_cfg5) $O/valu_summary_cfg5.json > /dev/null && echo valu ok
python3 profiles/traffic_summary.py $(f $O/pmc_fetch_cfg5) $(f $O/pmc_write_cfg5) $O/traffic_summary_cfg5.json > /dev/null && echo traffic ok
python3 - $(f $O/pmc_sq2_cfg5) <<'P'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1]+'/p_counter_collection.csv')):
    k=r['Kernel_Name'][:40]; d[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in d.items():
    if 'fwd' in k or 'bwd' in k or 'pinv' in k: print(k, {a:'%.4g'%b for a,b in v.items()})
P
