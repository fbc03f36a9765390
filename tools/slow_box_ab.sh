# run the cache-policy variants only on a box / placement in the slow mode (base pass > 16.5 ms); exit at once otherwise
cd $GRAFT_REPO_ROOT
one() { ( cd epidemicmodeling_amd && cp $1 libepiekf.so ); python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('$1  pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f'%(r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms']))"; }
cp epidemicmodeling_amd/libepiekf.so /tmp/keep.so
first=$(one lib_v_base.so); echo "$first"
ms=$(echo "$first" | awk '{print $3}')
if python3 -c "import sys; sys.exit(0 if float('$ms') > 16.5 else 1)"; then
  echo SLOW BOX
  for rep in 1 2; do for L in lib_v_base.so lib_v_ld0.so lib_v_ld19.so lib_v_st19.so lib_v_st17.so lib_v_st2.so lib_v_r03.so; do one $L; done; done
else echo fast box, nothing to do; fi
cp /tmp/keep.so epidemicmodeling_amd/libepiekf.so
