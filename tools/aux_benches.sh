# secondary bench lines of the round (GPU box): cfg3, NewCase sweep, reduced outputs, cfg5 both storages
mkdir -p gpurun_out/r02
run() { out=$1; shift; python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/r02/$out; python -c "
import json;r=json.load(open('gpurun_out/r02/$out'));k=r['kernels'];print('$out  %.3e steps/s  pass %.3f ms  fwd %.3f pinv %.3f bwd %.3f  frac %.3f'%(r['value'],r['ms_per_step'],k['ekf_fwd_ms'],k['eks_pinv_ms'],k['eks_bwd_ms'],r['roofline']['frac']))"; }
run bench_cfg3.json --workload cfg3 --regions 300 --t-hist 400 --steps 10 --warmup 3
run bench_newcase_sweep.json --workload newcase --steps 5 --warmup 2
run bench_cfg4_reduced_outputs.json --outputs reduced --steps 5 --warmup 2
run bench_cfg5_f64.json --workload cfg5 --regions 300 --eps 1024 --t-hist 400 --steps 5 --warmup 2
run bench_cfg5_f32.json --workload cfg5 --regions 300 --eps 1024 --t-hist 400 --steps 5 --warmup 2 --storage f32
