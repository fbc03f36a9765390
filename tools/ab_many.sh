# A/B of several builds of the library on one box:  tools/ab_many.sh "A.so B.so C.so" [bench args]   (files under epidemicmodeling_amd/)
set -e
LIBS=$1; shift
cd $GRAFT_REPO_ROOT/epidemicmodeling_amd
cp libepiekf.so /tmp/libepiekf_keep.so
for rep in $(seq 1 ${REPS:-3}); do
  for L in $LIBS; do
    cp $L libepiekf.so
    ( cd .. && timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('$L  pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f'%(r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))" )
  done
done
cp /tmp/libepiekf_keep.so libepiekf.so
