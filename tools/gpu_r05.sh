#!/bin/bash
# Round-5 collection script, one stage per call:  tools/gpu_r05.sh STAGE [args]
#   hextest      the GPU tests that run the hex shape (-k hex) + smoke
#   sweep [args] tools/batch_sweep.sh with extra bench args (e.g. --shape quad)
#   suite        the whole GPU suite
#   bench [args] one bench line
# Output under gpurun_out/r05/.
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
stage=$1; shift
case $stage in
  hextest)
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hex" > $O/hextest.txt 2>&1; rc=$?
    tail -15 $O/hextest.txt; exit $rc ;;
  sweep)
    tag=$1; shift
    EXTRA="$*" bash tools/batch_sweep.sh > $O/sweep_$tag.txt 2>&1; rc=$?
    cat $O/sweep_$tag.txt; exit $rc ;;
  suite)
    timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?
    tail -8 $O/gpu_tests.txt; exit $rc ;;
  bench)
    tag=$1; shift
    timeout -k 10 600 python bench.py "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; rc=$?
    tail -c 1500 $O/bench_$tag.json; exit $rc ;;
esac
