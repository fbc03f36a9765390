# A/B of quad-kernel builds (EPI_QUAD_WAVES = 1..4) on three batch sizes; usage: bash tools/quad_ab.sh "1 2 3 4"
for w in $1; do
  echo "== EPI_QUAD_WAVES=$w"
  for cfg in "75 125" "150 125" "150 250" "300 250"; do
    set -- $cfg
    EPIEKF_LIB=$PWD/epidemicmodeling_amd/libepiekf_q$w.so EPIEKF_SHAPE=2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --regions $1 --eps $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%6d chains  %.3e steps/s  pass %.2f ms  fwd %.2f pinv %.2f bwd %.2f'%(r['config']['chains_per_gpu'], r['value'], r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))"
  done
done
