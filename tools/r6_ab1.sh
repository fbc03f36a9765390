#!/bin/bash
# round 6, step 1: parity of the fixed-descriptor smoother (both variants), then A/B of the three builds on this box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for v in 1 2; do
  EPIEKF_LIB=$GRAFT_REPO_ROOT/ab/l6_bwd$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane6 or headline_sweep_at_full_size" > gpurun_out/r06/parity_l6_bwd$v.txt 2>&1 || { tail -30 gpurun_out/r06/parity_l6_bwd$v.txt; exit 1; }
  tail -3 gpurun_out/r06/parity_l6_bwd$v.txt
done
bash tools/ab_variants.sh "l6_bwd0.so l6_bwd1.so l6_bwd2.so" 3 --placement-tries 1 | tee gpurun_out/r06/ab_l6_bwd.txt
