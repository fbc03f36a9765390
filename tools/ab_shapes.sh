cd $GRAFT_REPO_ROOT
for cfg in "150 125" "128 160" "164 125" "175 125"; do
  set -- $cfg
  for sh in auto hex lane; do
    for rep in 1 2; do
      X=""; [ $sh != auto ] && X="--shape $sh"
      timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --regions $1 --eps $2 --placement-tries 1 $X 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%6d chains %-5s pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f  %s'%($1*$2,'$sh', r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms'], r['config']['shape']))"
    done
  done
done
