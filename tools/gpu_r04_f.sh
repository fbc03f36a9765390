# round 4, GPU call F: reduced-output mode, time pipeline forced on / off / auto, alternating (same box, same process order)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for rep in 1 2 3; do
for tp in -1 1; do
python3 bench.py --no-cpu-baseline --outputs reduced --time-pipe $tp --steps 10 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('reduced time_pipe $tp', round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['kernels'].items() if k.endswith('_ms')}, int(r['roofline']['measured_copy']['copy_8B_per_lane_GBs']))"
done
done
for tp in -1 1; do
python3 bench.py --no-cpu-baseline --time-pipe $tp --steps 10 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('all outputs time_pipe $tp', round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['kernels'].items() if k.endswith('_ms')}, int(r['roofline']['measured_copy']['copy_8B_per_lane_GBs']))"
done
