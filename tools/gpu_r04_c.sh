# round 4, GPU call C: exact_nonfinite (dense re-run of overflowed chains) -- the overflow test, the property tests with the
# time-flipped wrappers back in the long draws, the other-configs LAPACK report; reduced outputs with the time pipeline forced.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overflow or other_baseline or status_and_guard or hip_graph or host" > $O/gpu_tests_c.txt 2>&1; rc=$?; tail -15 $O/gpu_tests_c.txt
[ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/gpu_fuzz_c.txt 2>&1; rc=$?; tail -15 $O/gpu_fuzz_c.txt
[ $rc -eq 0 ] || exit $rc
for tp in 0 1; do
python3 bench.py --no-cpu-baseline --outputs reduced --time-pipe $tp 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('reduced time_pipe $tp', round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['kernels'].items() if k.endswith('_ms')}, int(r['roofline']['measured_copy']['copy_8B_per_lane_GBs']))"
done
