# Re-measures profiles/r02/traffic_summary.json (and the SQ summary) for the CURRENT kernel sources: run through gpurun as the
# last step after any kernel change, then copy gpurun_out/r02/{traffic,valu}_summary*.json into profiles/r02/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/profiles/traffic_probe.py > /dev/null 2>&1 && echo fetch
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/profiles/traffic_probe.py > /dev/null 2>&1 && echo write
cd $R
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
python3 profiles/traffic_summary.py $(f $O/pmc_fetch) $(f $O/pmc_write) $O/traffic_summary.json | head -3
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_check.json 2>/dev/null; cp $O/traffic_summary.json profiles/r02/traffic_summary.json; python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('traffic in bench line:', r['roofline']['traffic'], r['roofline']['traffic_GBs'])"
