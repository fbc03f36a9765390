#!/bin/bash
# A/B/C... of several builds of the library on one box (EPIEKF_LIB selects the build; files under ab/):
#   tools/ab_variants.sh "A.so B.so C.so" REPS [bench args]
LIBS=$1; REPS=${2:-2}; shift 2
cd $GRAFT_REPO_ROOT
for rep in $(seq $REPS); do
  for L in $LIBS; do
    EPIEKF_LIB=$GRAFT_REPO_ROOT/ab/$L timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%-22s pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f  %s'%('$L', r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms'], r['config']['shape']))"
  done
done
