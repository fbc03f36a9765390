# round 4, GPU call H: the 3-state wave shape (seven chains per wavefront): parity, then BASELINE config 3 in both shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_h.txt 2>&1; rc=$?; tail -12 $O/gpu_tests_h.txt
[ $rc -eq 0 ] || exit $rc
for sh in wave lane; do
for reg in 300 2000 7000 20000; do
python3 bench.py --no-cpu-baseline --workload cfg3 --regions $reg --steps 20 --shape $sh 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('cfg3 $reg chains shape $sh', round(r['ms_per_step'],4), {k:round(v,4) for k,v in r['kernels'].items() if k.endswith('_ms')}, r['config']['shape'], r['config']['lane_block'])"
done
done
