# round 4, GPU call J: after the last source change -- the GPU suite, a fresh random hunt, then the bench lines (which now find
# the PMC summaries of these sources under profiles/r04) and the host-call profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -4 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
python3 bench.py > $O/bench_cfg4.json 2>/dev/null && echo bench ok
python3 bench.py --workload cfg4-live > $O/bench_cfg4_live.json 2>/dev/null && echo live ok
python3 bench.py --no-cpu-baseline --outputs reduced > $O/bench_cfg4_reduced.json 2>/dev/null && echo reduced ok
python3 bench.py --no-cpu-baseline --workload cfg3 --steps 20 > $O/bench_cfg3.json 2>/dev/null && echo cfg3 ok
python3 bench.py --no-cpu-baseline --workload cfg5 --eps 1024 --storage f32 > $O/bench_cfg5_f32.json 2>/dev/null && echo cfg5 ok
python3 bench.py --no-cpu-baseline --workload newcase > $O/bench_newcase.json 2>/dev/null && echo newcase ok
python3 profiles/host_calls.py > $O/host_calls.txt 2>&1; cp gpurun_out/host_calls.json $O/ 2>/dev/null; echo host calls ok
EPI_FUZZ_EXAMPLES=${1:-1000} timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_hunt.txt 2>&1; rc=$?; tail -4 $O/fuzz_hunt.txt
exit $rc
