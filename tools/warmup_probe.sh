cd $GRAFT_REPO_ROOT
for ws in "2 8" "5 20" "50 50" "200 100" "2 8"; do set -- $ws
python bench.py --no-cpu-baseline --regions 75 --eps 125 --placement-tries 1 --warmup $1 --steps $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('warmup %4d steps %4d  pass %6.3f ms  fwd %5.2f pinv %5.2f bwd %5.2f'%($1,$2, r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))"
done
