# alternate several builds of the library on one box:  tools/ab3.sh "a.so b.so c.so" [bench args]
set -e
LIBS=$1; shift
cd $GRAFT_REPO_ROOT/epidemicmodeling_amd
for rep in 1 2 3; do
  for L in $LIBS; do
    cp $L libepiekf.so
    ( cd .. && timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('$L  pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f'%(r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))" )
  done
done
