"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 profiles/traffic_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 profiles/traffic_probe.py
    python3 profiles/traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/traffic_summary.json

1. a calibration copy of a known byte count with the filter kernels' access shape (8 B per lane): the guide
   (MI355X_MICROARCH.md, HBM) calibrates FETCH_SIZE only for 16 B/lane streams and says other widths must be
   calibrated on a known byte count in one's own access pattern;
2. two passes of the headline sweep, stage by stage (ekf_fwd_sym, eks_pinv, eks_bwd_sym)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from epidemicmodeling_amd import _lib, batch, synth  # noqa: E402

CALIB_DOUBLES = 1 << 29          # 4 GiB read + 4 GiB written: far beyond the 256 MiB Infinity Cache

dev = torch.device("cuda:0")
src = torch.rand(CALIB_DOUBLES, dtype=torch.float64, device=dev)
dst = torch.empty_like(src)
err = C.create_string_buffer(256)
st = torch.cuda.current_stream(dev)
for _ in range(2):
    _lib.check(_lib.lib().epi_calib_copy_f64_device(src.data_ptr(), dst.data_ptr(), CALIB_DOUBLES,
                                                    C.c_void_p(st.cuda_stream), err), err)
torch.cuda.synchronize()
del src, dst
# `python traffic_probe.py shard`: the 9 375-chain shard one of 8 GPUs runs (four lanes per chain) instead of the whole sweep;
# `live`: the living multi-wave epidemic (bench.py --workload cfg4-live); `reduced`: the two outputs the caller consumes
args = set(sys.argv[1:])
if "wave250" in args:             # a region's 250 cost weights: the one-wavefront-per-chain shape (round 4)
    w = synth.make_cfg4(1, 250)
elif "cfg5" in args:              # BASELINE config 5: 307 200 3-state chains x 400 days, fp32 storage
    w = synth.make_cfg5(300, 1024, 400)
elif "shard4" in args:            # the 18 750-chain shard one of 4 GPUs runs: the hex kernels at two waves per SIMD
    w = synth.make_cfg4(150, 125)
else:
    w = synth.make_cfg4(75, 125, live="live" in args) if "shard" in args else synth.make_cfg4(live="live" in args)
outs = ["u_opt_smooth", "S_SMOOTH"] if "reduced" in args else None
r = batch.EkfRunner(batch.DeviceWorkload(w, dev), outputs=outs, lane_block="auto",    # bench.py's default layout and lane mapping
                    storage="f32" if "cfg5" in args else "f64")
for _ in range(20 if "wave250" in args else 2):
    for ph in (1, 3, 4):
        r.run(phase=ph)
torch.cuda.synchronize()
print("probe done: calib doubles", CALIB_DOUBLES, "steps per pass", w.B * w.T)
