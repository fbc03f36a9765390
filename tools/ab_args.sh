#!/bin/bash
# one library, several bench argument sets on one box:  tools/ab_args.sh LIB.so REPS "args1" "args2" ...
LIB=$1; REPS=$2; shift 2
cd $GRAFT_REPO_ROOT
for rep in $(seq $REPS); do
  for A in "$@"; do
    EPIEKF_LIB=$GRAFT_REPO_ROOT/ab/$LIB timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $A 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%-44s pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f  %s'%('$A', r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms'], r['config']['shape']))"
  done
done
