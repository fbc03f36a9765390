# round 4, GPU call G: final collection for the current kernel sources -- PMC traffic (survey's series, living epidemic, reduced
# outputs), SQ counters (all outputs, reduced outputs, a 250-chain batch in the wave shape), staged-only and one-call stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_staged -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg4_staged_under_rocprof.json 2>/dev/null && echo staged
EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_staged_live -o bench -- python3 $R/bench.py --no-cpu-baseline --workload cfg4-live > $O/bench_cfg4_live_staged_under_rocprof.json 2>/dev/null && echo staged live
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_onecall -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg4_onecall_under_rocprof.json 2>/dev/null && echo onecall
for v in "" live reduced; do
  t=${v:+_$v}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$t -o p -- python3 $R/profiles/traffic_probe.py $v > /dev/null 2>&1 && echo fetch $v
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$t -o p -- python3 $R/profiles/traffic_probe.py $v > /dev/null 2>&1 && echo write $v
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq$t -o p -- python3 $R/profiles/traffic_probe.py $v > /dev/null 2>&1 && echo sq $v
done
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_wave250 -o p -- python3 $R/profiles/traffic_probe.py wave250 > /dev/null 2>&1 && echo sq wave250
cd $R
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
for v in "" live reduced; do
  t=${v:+_$v}
  python3 profiles/traffic_summary.py $(f $O/pmc_fetch$t) $(f $O/pmc_write$t) $O/traffic_summary$t.json > /dev/null && echo traffic $v ok
  python3 profiles/valu_summary.py $(f $O/pmc_sq$t) $O/valu_summary$t.json > /dev/null && echo valu $v ok
done
python3 profiles/valu_summary.py $(f $O/pmc_sq_wave250) $O/valu_summary_wave250.json > /dev/null && echo valu wave250 ok
python3 profiles/bwd_launch_classes.py $O/stats_onecall $O/bwd_launch_classes_onecall.json > /dev/null && echo bwd classes ok
bash tools/timeline.sh 300 250 > $O/timeline_75000.txt 2>&1; echo timeline ok
python3 bench.py > $O/bench_cfg4.json 2>/dev/null && echo bench ok
python3 bench.py --workload cfg4-live > $O/bench_cfg4_live.json 2>/dev/null && echo live ok
python3 bench.py --no-cpu-baseline --outputs reduced > $O/bench_cfg4_reduced.json 2>/dev/null && echo reduced ok
python3 bench.py --no-cpu-baseline --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>/dev/null && echo cfg3 ok
python3 bench.py --no-cpu-baseline --workload cfg5 --eps 1024 --storage f32 > $O/bench_cfg5_f32.json 2>/dev/null && echo cfg5 ok
python3 bench.py --no-cpu-baseline --workload newcase > $O/bench_newcase.json 2>/dev/null && echo newcase ok
python3 profiles/host_calls.py > $O/host_calls.txt 2>&1; cp gpurun_out/host_calls.json $O/ 2>/dev/null; echo host calls ok
