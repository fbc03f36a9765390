# round 4, GPU call A: the whole GPU suite (new: live full-size parity, live LAPACK report, frozen referee vectors), then the
# bench on the survey's workload and on the living epidemic.  Results under gpurun_out/r04/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
python3 bench.py > $O/bench_cfg4.json 2>$O/bench_cfg4.err && echo bench ok
python3 bench.py --workload cfg4-live > $O/bench_cfg4_live.json 2>$O/bench_cfg4_live.err && echo live ok
python3 bench.py --no-cpu-baseline --outputs reduced > $O/bench_cfg4_reduced.json 2>/dev/null && echo reduced ok
python3 bench.py --no-cpu-baseline --outputs reduced --workload cfg4-live > $O/bench_cfg4_live_reduced.json 2>/dev/null && echo live reduced ok
python3 profiles/pinv_rank_histogram.py $O/pinv_rank_histogram.json > $O/pinv_rank_histogram.txt 2>&1 && echo hist ok
python3 - <<'PY'
import json
for f in ("bench_cfg4", "bench_cfg4_live", "bench_cfg4_reduced", "bench_cfg4_live_reduced"):
    try:
        r = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernels"].items() if k.endswith("_ms")}, r["roofline"]["measured_copy"])
    except Exception as e:
        print(f, "failed", e)
PY
