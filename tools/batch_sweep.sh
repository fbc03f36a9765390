# ms per pass of the headline sweep's shards:  tools/batch_sweep.sh [extra bench.py args, e.g. --shape lane | --time-pipe -1]
for cfg in "300 1" "75 125" "150 125" "128 256" "150 250" "300 250"; do
  set -- $cfg
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --regions $1 --eps $2 $EXTRA "${@:3}" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('%6d chains  %.3e steps/s  pass %6.2f ms  fwd %5.2f pinv %5.2f bwd %5.2f   %s'%(r['config']['chains_per_gpu'], r['value'], r['ms_per_step'], k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms'], r['config']['shape']))"
done
