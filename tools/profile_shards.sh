# rocprofv3 --kernel-trace --stats of the N = 2 and N = 4 shards of the headline sweep (run through gpurun); the per-kernel
# summaries land in gpurun_out/r02/ as bench_shard{18750,37500}_kernel_stats.csv next to the bench line of the same run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats18750 -o bench -- python3 $R/bench.py --no-cpu-baseline --regions 75 --eps 250 > $O/bench_shard18750_under_rocprof.json 2>/dev/null && echo stats18750
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats37500 -o bench -- python3 $R/bench.py --no-cpu-baseline --regions 150 --eps 250 > $O/bench_shard37500_under_rocprof.json 2>/dev/null && echo stats37500
cd $R
for n in 18750 37500; do f=$(find $O/stats$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_shard${n}_kernel_stats.csv; head -8 $O/bench_shard${n}_kernel_stats.csv | cut -c1-160; done
