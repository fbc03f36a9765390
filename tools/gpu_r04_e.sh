# round 4, GPU call E: the whole GPU suite with the wave shape and exact_nonfinite in, a 2000-example fuzz hunt, benches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -6 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
EPI_FUZZ_EXAMPLES=2000 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz_hunt.txt 2>&1; rc=$?; tail -6 $O/fuzz_hunt.txt
[ $rc -eq 0 ] || exit $rc
python3 bench.py > $O/bench_cfg4.json 2>$O/bench_cfg4.err && echo bench ok
python3 bench.py --no-cpu-baseline --workload cfg4-live > $O/bench_cfg4_live.json 2>/dev/null && echo live ok
python3 profiles/shape_latency.py $O/shape_latency.json > $O/shape_latency.txt 2>&1 && echo latency ok
python3 - <<'PY'
import json
for f in ("bench_cfg4", "bench_cfg4_live"):
    r = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
    print(f, round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernels"].items() if k.endswith("_ms")}, r["roofline"]["measured_copy"], r.get("box_normalised"))
PY
