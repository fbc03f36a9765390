for cfg in "300 1" "75 16" "75 32" "75 64" "75 125" "150 125"; do
  set -- $cfg
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --regions $1 --eps $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%6d chains  pass %.2f ms  fwd %.2f pinv %.2f bwd %.2f'%(r['config']['chains_per_gpu'], r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))"
done
