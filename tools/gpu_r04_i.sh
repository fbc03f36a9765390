# round 4, GPU call I: host-pointer entry points after the dense-copy change: their tests, then the wall-time profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "host" > $O/gpu_tests_host.txt 2>&1; rc=$?; tail -5 $O/gpu_tests_host.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 profiles/host_calls.py > $O/host_calls.txt 2>&1; tail -8 $O/host_calls.txt; cp gpurun_out/host_calls.json $O/ 2>/dev/null; echo host calls done
