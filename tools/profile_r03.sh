# Collects the round-3 profiles on the GPU box (run through gpurun); results land in gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats75k -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_cfg4_under_rocprof.json 2>/dev/null && echo stats75k
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats9375 -o bench -- python3 $R/bench.py --no-cpu-baseline --regions 75 --eps 125 > $O/bench_shard9375_under_rocprof.json 2>/dev/null && echo stats9375
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/profiles/traffic_probe.py > /dev/null 2>&1 && echo fetch
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/profiles/traffic_probe.py > /dev/null 2>&1 && echo write
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/profiles/traffic_probe.py > /dev/null 2>&1 && echo sq75k
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_shard -o p -- python3 $R/profiles/traffic_probe.py shard > /dev/null 2>&1 && echo sqshard
cd $R
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
python3 profiles/traffic_summary.py $(f $O/pmc_fetch) $(f $O/pmc_write) $O/traffic_summary.json > /dev/null && echo traffic ok
python3 profiles/valu_summary.py $(f $O/pmc_sq) $O/valu_summary.json > /dev/null && echo valu ok
python3 profiles/valu_summary.py $(f $O/pmc_sq_shard) $O/valu_summary_shard9375.json > /dev/null && echo valu shard ok
find $O -name "*kernel_stats.csv" -exec sh -c 'cp "$1" $2/$(basename $(dirname $(dirname "$1")))_kernel_stats.csv' _ {} $O \; 2>/dev/null
ls $O
python3 profiles/bwd_launch_classes.py $O/stats75k $O/bwd_launch_classes.json > /dev/null && echo bwd classes ok
python3 bench.py > $O/bench_cfg4_final.json 2>/dev/null && echo bench ok
python3 bench.py --no-cpu-baseline --outputs reduced > $O/bench_cfg4_reduced.json 2>/dev/null && echo reduced ok
python3 bench.py --no-cpu-baseline --workload cfg5 --eps 1024 --storage f32 > $O/bench_cfg5_f32.json 2>/dev/null && echo cfg5 ok
python3 bench.py --no-cpu-baseline --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>/dev/null && echo cfg3 ok
python3 bench.py --no-cpu-baseline --workload newcase > $O/bench_newcase.json 2>/dev/null && echo newcase ok
( echo "bench.py --steps 8 --warmup 2 --no-cpu-baseline --regions R --eps E   (one MI355X, round 3; pass = ONE epi_sweep_run_device call: filter + scoring tail + Pareto filter, wall clock;"
  echo "fwd / pinv / bwd = HIP-event durations of the stages enqueued one by one; fwd includes the monitor kernel)"; echo
  echo "== shape and launch chosen by the library"; bash tools/batch_sweep.sh; echo
  echo "== time pipeline off (--time-pipe -1)"; EXTRA="--time-pipe -1" bash tools/batch_sweep.sh; echo
  echo "== one lane per chain forced (--shape lane)"; EXTRA="--shape lane" bash tools/batch_sweep.sh; echo
  echo "== four lanes per chain forced (--shape quad)"; EXTRA="--shape quad" bash tools/batch_sweep.sh ) > $O/batch_size_sweep.txt 2>&1 && echo sweep ok
bash tools/timeline.sh 75 125 > $O/timeline_9375.txt 2>&1; bash tools/timeline.sh 300 250 > $O/timeline_75000.txt 2>&1; echo timelines ok
python3 profiles/pinv_by_days.py > $O/pinv_by_days.txt 2>/dev/null && echo pinv days ok
