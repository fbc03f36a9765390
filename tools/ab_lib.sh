# A/B of library builds on the headline sweep: bash tools/ab_lib.sh libA.so libB.so ...
for L in "$@"; do
  EPIEKF_LIB=$PWD/epidemicmodeling_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('$L  pass %.2f ms  fwd %.2f pinv %.2f bwd %.2f'%(r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))"
done
