# copies what tools/profile_r03.sh left in gpurun_out/r03 into profiles/r03 (run here, after the gpurun call)
S=${1:-gpurun_out/r03}; D=profiles/r03
for f in bench_cfg3.json bench_cfg4_final.json bench_cfg4_reduced.json bench_cfg4_under_rocprof.json bench_cfg5_f32.json bench_newcase.json \
         bench_shard9375_under_rocprof.json bwd_launch_classes.json batch_size_sweep.txt pinv_by_days.txt timeline_75000.txt timeline_9375.txt \
         traffic_summary.json valu_summary.json valu_summary_shard9375.json; do cp $S/$f $D/$f; done
if [ -f $S/stats75k/bench_kernel_stats.csv ]; then cp $S/stats75k/bench_kernel_stats.csv $D/bench_cfg4_kernel_stats.csv; cp $S/stats9375/bench_kernel_stats.csv $D/bench_shard9375_kernel_stats.csv
else cp $S/bench_cfg4_kernel_stats.csv $S/bench_shard9375_kernel_stats.csv $D/; fi
