#!/bin/bash
# bit-exactness of A/B builds of the library (files under ab/) on the tests that reach the headline's kernels, then the A/B itself
# on one box (tools/ab_variants.sh):   tools/ab_parity.sh "parity_lib.so ..." "ab libs" REPS [bench args]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for L in $1; do
  EPIEKF_LIB=$GRAFT_REPO_ROOT/ab/$L timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane6 or headline_sweep_at_full_size" > gpurun_out/r06/parity_$L.txt 2>&1 || { tail -40 gpurun_out/r06/parity_$L.txt; exit 1; }
  tail -2 gpurun_out/r06/parity_$L.txt
done
LIBS=$2; REPS=$3; shift 3
bash tools/ab_variants.sh "$LIBS" $REPS "$@"
