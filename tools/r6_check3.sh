#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane6 or monitor_kernels or fastest_of_several or reverse_time" 2>&1 | tail -5 || exit 1
bash tools/ab_variants.sh "l6_bwd0.so final_a.so" 3 --placement-tries 1 --regions 150 --eps 250 | tee gpurun_out/r06/ab_n2_shard.txt
bash tools/ab_variants.sh "l6_bwd0.so final_a.so" 2 --placement-tries 1 | tee gpurun_out/r06/ab_headline_final_a.txt
