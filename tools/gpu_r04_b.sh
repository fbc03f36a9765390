# round 4, GPU call B: profiles.  (1) rocprofv3 --kernel-trace --stats of a STAGED-ONLY bench run (EPI_BENCH_STAGED=1: one kind
# of launch per kernel, so the CSV's averages ARE the roofline kernel times) on both workloads; (2) the one-call timeline
# separately; (3) FETCH_SIZE / WRITE_SIZE passes for the survey's workload, the living epidemic and reduced outputs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_staged -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg4_staged_under_rocprof.json 2>/dev/null && echo staged
EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_staged_live -o bench -- python3 $R/bench.py --no-cpu-baseline --workload cfg4-live > $O/bench_cfg4_live_staged_under_rocprof.json 2>/dev/null && echo staged live
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_onecall -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg4_onecall_under_rocprof.json 2>/dev/null && echo onecall
for v in "" live reduced; do
  t=${v:+_$v}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$t -o p -- python3 $R/profiles/traffic_probe.py $v > /dev/null 2>&1 && echo fetch $v
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$t -o p -- python3 $R/profiles/traffic_probe.py $v > /dev/null 2>&1 && echo write $v
done
cd $R
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
for v in "" live reduced; do
  t=${v:+_$v}
  python3 profiles/traffic_summary.py $(f $O/pmc_fetch$t) $(f $O/pmc_write$t) $O/traffic_summary$t.json > /dev/null && echo traffic $v ok
done
find $O -name "*kernel_stats.csv" -exec sh -c 'cp "$1" $2/$(basename $(dirname $(dirname "$1")))_kernel_stats.csv' _ {} $O \; 2>/dev/null
python3 profiles/bwd_launch_classes.py $O/stats_onecall $O/bwd_launch_classes_onecall.json > /dev/null && echo bwd classes ok
bash tools/timeline.sh 300 250 > $O/timeline_75000.txt 2>&1; echo timeline ok
ls $O
