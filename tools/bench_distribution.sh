#!/bin/bash
# The default bench and the 9 375-chain shard on N freshly acquired boxes (one gpurun call each; run from the build container):
#   tools/bench_distribution.sh N OUTFILE
N=${1:-8}; OUT=${2:-profiles/r06/bench_distribution.txt}
cd /root/repo
mkdir -p gpurun_out/dist
{
echo "# python bench.py --no-cpu-baseline, python bench.py --no-cpu-baseline --regions 75 --eps 125 and the latter with --spinup-ms 0 on $N freshly acquired boxes, one gpurun call each:"
echo "# ms per pass, stages, this box's 8 B/lane copy bandwidth, the placement report of the headline run (staged ms of the allocations tried, mode),"
echo "# and what 8 GPUs would make of the fixed sweep on boxes like this one (headline pass / shard pass)"
for i in $(seq $N); do
  /usr/local/graft/bin/gpurun --timeout 500 -- "mkdir -p gpurun_out/dist && python bench.py --no-cpu-baseline > gpurun_out/dist/h_$i.json 2>/dev/null && python bench.py --no-cpu-baseline --regions 75 --eps 125 > gpurun_out/dist/s_$i.json 2>/dev/null && python bench.py --no-cpu-baseline --regions 75 --eps 125 --spinup-ms 0 > gpurun_out/dist/n_$i.json 2>/dev/null" > /dev/null 2>&1
  python3 - $i <<'PY'
import json, sys
i = sys.argv[1]
try:
    h = json.loads(open(f"/root/repo/gpurun_out/dist/h_{i}.json").read().strip().splitlines()[-1])
    s = json.loads(open(f"/root/repo/gpurun_out/dist/s_{i}.json").read().strip().splitlines()[-1])
except Exception as e:
    print(f"box {i}: no result ({e})"); sys.exit(0)
hk, sk, pl = h["kernels"], s["kernels"], h["config"].get("placement") or {}
try:
    n = json.loads(open(f"/root/repo/gpurun_out/dist/n_{i}.json").read().strip().splitlines()[-1])["ms_per_step"]
except Exception:
    n = float("nan")
print("box %s  headline %6.2f ms (fwd %.2f pinv %.2f bwd %.2f)  copy %.2f TB/s  tries %s mode %s   shard %5.2f ms (fwd %.2f pinv %.2f bwd %.2f; %.2f without the spin-up)   ratio %.2f" % (
    i, h["ms_per_step"], hk["ekf_fwd_ms"], hk["eks_pinv_ms"], hk["eks_bwd_ms"], h["roofline"]["measured_copy"]["copy_8B_per_lane_GBs"] / 1e3,
    pl.get("tries"), pl.get("mode"), s["ms_per_step"], sk["ekf_fwd_ms"], sk["eks_pinv_ms"], sk["eks_bwd_ms"], n, h["ms_per_step"] / s["ms_per_step"]))
PY
done
} > $OUT 2>&1
cat $OUT
