# is the forward stage's 6.05 / 7.05 ms bimodality (same box, same library, different processes) the kernel or the monitor?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/bimodal; mkdir -p $O
for i in 1 2 3 4 5 6; do
  EPI_BENCH_STAGED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$i -o b -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench$i.json 2>/dev/null
  python3 - $O/run$i $O/bench$i.json <<'P'
import csv,glob,json,sys
f=glob.glob(sys.argv[1]+'/**/b_kernel_stats.csv',recursive=True)[0]
r=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
row={}
for x in csv.DictReader(open(f)):
    n=x['Name']
    for k in ('ekf_fwd_sym','ekf_monitor','eks_pinv','eks_bwd_sym'):
        if k in n: row[k]=(round(float(x['AverageNs'])/1e6,3),round(float(x['MinNs'])/1e6,3),round(float(x['MaxNs'])/1e6,3))
print('pass %.2f'%r['ms_per_step'], 'bench fwd %.2f'%r['kernels']['ekf_fwd_ms'], row, flush=True)
P
done
