# a fresh random hunt of tests/test_gpu_fuzz.py:  bash tools/gpu_hunt.sh EXAMPLES TAG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${EPI_ROUND:-r05}; mkdir -p $O
cd $R
EPI_FUZZ_EXAMPLES=${1:-5000} timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_hunt_${2:-a}.txt 2>&1; rc=$?; tail -4 $O/fuzz_hunt_${2:-a}.txt
exit $rc
