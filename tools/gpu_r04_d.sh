# round 4, GPU call D: the one-wavefront-per-chain shape: parity, then latency against the other shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg4_sia6 or backward_sia6 or blocked or sweep_entry or time_pipelined or referee or overflow or row3" > $O/gpu_tests_d.txt 2>&1; rc=$?; tail -15 $O/gpu_tests_d.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 profiles/shape_latency.py $O/shape_latency.json 2>&1 | tail -8
