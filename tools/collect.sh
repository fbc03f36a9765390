#!/bin/bash
# One parameterised collection script (replaces the per-round one-off scripts):   tools/collect.sh ROUND STAGE [args]
# run through gpurun; results land in gpurun_out/ROUND/ and are copied into profiles/ROUND/ by hand afterwards.
#   suite              the GPU test suite                                  -> gpu_tests.txt
#   hunt N TAG         a fresh N-example random hunt of tests/test_gpu_fuzz.py -> fuzz_hunt_TAG.txt
#   bench              the bench lines of every BASELINE config            -> bench_*.json
#   stats              rocprofv3 --kernel-trace --stats of the staged headline bench and of the 9 375-chain shard
#   traffic [probe args]   FETCH_SIZE / WRITE_SIZE in separate --pmc passes of profiles/traffic_probe.py + summary (tag = args joined)
#   sq [probe args]    SQ counters (profiles/valu_summary.py)
#   sweep              ms per pass by batch size, every shape              -> batch_size_sweep.txt
#   timeline           kernel timelines of the shard and of the full batch
#   host               host-pointer entry points, PCIe included            -> host_calls.json
#   shapes             profiles/shape_latency.py                           -> shape_latency.json
set -o pipefail
ROUND=$1; STAGE=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
f() { find $1 -name "p_counter_collection.csv" | head -1 | xargs dirname; }
tag() { local t; t=$(echo "$*" | tr ' ' '_'); echo ${t:+_$t}; }
cd $R
case $STAGE in
  suite)
    timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -4 $O/gpu_tests.txt; exit $rc ;;
  hunt)
    EPI_FUZZ_EXAMPLES=${1:-3000} timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_hunt_${2:-a}.txt 2>&1; rc=$?
    tail -4 $O/fuzz_hunt_${2:-a}.txt; exit $rc ;;
  bench)
    python3 bench.py > $O/bench_cfg4.json 2>/dev/null && echo cfg4 ok
    python3 bench.py --workload cfg4-live > $O/bench_cfg4_live.json 2>/dev/null && echo live ok
    python3 bench.py --no-cpu-baseline --outputs reduced > $O/bench_cfg4_reduced.json 2>/dev/null && echo reduced ok
    python3 bench.py --no-cpu-baseline --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>/dev/null && echo cfg3 ok
    python3 bench.py --no-cpu-baseline --workload cfg5 --eps 1024 --storage f32 > $O/bench_cfg5_f32.json 2>/dev/null && echo cfg5 ok
    python3 bench.py --no-cpu-baseline --workload newcase > $O/bench_newcase.json 2>/dev/null && echo newcase ok
    python3 bench.py --no-cpu-baseline --regions 75 --eps 125 > $O/bench_shard9375.json 2>/dev/null && echo shard ok ;;
  stats)
    cd /tmp && export TMPDIR=/tmp
    EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats75k -o bench -- python3 $R/bench.py --no-cpu-baseline --placement-tries 1 > $O/bench_cfg4_staged_under_rocprof.json 2>/dev/null && echo stats75k
    EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats9375 -o bench -- python3 $R/bench.py --no-cpu-baseline --placement-tries 1 --regions 75 --eps 125 > $O/bench_shard9375_staged_under_rocprof.json 2>/dev/null && echo stats9375
    cd $R
    cp $(find $O/stats75k -name "*kernel_stats.csv" | head -1) $O/bench_cfg4_staged_kernel_stats.csv
    cp $(find $O/stats9375 -name "*kernel_stats.csv" | head -1) $O/bench_shard9375_staged_kernel_stats.csv
    rm -rf $O/stats75k $O/stats9375; head -8 $O/bench_cfg4_staged_kernel_stats.csv ;;
  traffic)
    T=$(tag "$@"); cd /tmp && export TMPDIR=/tmp; rm -rf $O/pmc_fetch $O/pmc_write
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/profiles/traffic_probe.py "$@" > /dev/null 2>&1 && echo fetch
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/profiles/traffic_probe.py "$@" > /dev/null 2>&1 && echo write
    cd $R; python3 profiles/traffic_summary.py $(f $O/pmc_fetch) $(f $O/pmc_write) $O/traffic_summary$T.json | head -12; rm -rf $O/pmc_fetch $O/pmc_write ;;
  sq)
    T=$(tag "$@"); cd /tmp && export TMPDIR=/tmp; rm -rf $O/pmc_sq
    SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
    rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/profiles/traffic_probe.py "$@" > /dev/null 2>&1 && echo sq
    cd $R; python3 profiles/valu_summary.py $(f $O/pmc_sq) $O/valu_summary$T.json | head -30; rm -rf $O/pmc_sq ;;
  sweep)
    ( echo "bench.py --steps 8 --warmup 2 --no-cpu-baseline --regions R --eps E   (one MI355X; pass = ONE epi_sweep_run_device call: filter + scoring tail + Pareto filter, wall clock;"
      echo "fwd / pinv / bwd = HIP-event durations of the stages enqueued one by one; fwd includes the monitor kernel)"; echo
      echo "== shape and launch chosen by the library"; bash tools/batch_sweep.sh; echo
      echo "== six lanes per chain forced (--shape hex)"; EXTRA="--shape hex" bash tools/batch_sweep.sh; echo
      echo "== four lanes per chain forced (--shape quad)"; EXTRA="--shape quad" bash tools/batch_sweep.sh; echo
      echo "== one lane per chain forced (--shape lane)"; EXTRA="--shape lane" bash tools/batch_sweep.sh ) > $O/batch_size_sweep.txt 2>&1; cat $O/batch_size_sweep.txt ;;
  timeline)
    bash tools/timeline.sh 75 125 > $O/timeline_9375.txt 2>&1; bash tools/timeline.sh 300 250 > $O/timeline_75000.txt 2>&1; cat $O/timeline_9375.txt ;;
  host)
    python3 profiles/host_calls.py > $O/host_calls.txt 2>&1; cp gpurun_out/host_calls.json $O/ 2>/dev/null; tail -5 $O/host_calls.txt ;;
  shapes)
    timeout -k 10 700 python3 profiles/shape_latency.py $O/shape_latency.json > $O/shape_latency.log 2>&1; tail -3 $O/shape_latency.log | cut -c1-300 ;;
  *) echo "unknown stage $STAGE"; exit 2 ;;
esac
