"""What one library call costs at a given batch size, four ways (ms per call, 20 back-to-back calls, wall clock):
the filter alone (epi_ekf_run_device), the same replayed from a captured HIP graph, the sweep call (filter + scoring
tail, epi_sweep_run_device) without and with the Pareto filter.
    python profiles/graph_probe.py [regions eps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth, layout as L_

R, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (75, 125)
dev = torch.device("cuda:0")
w = synth.make_cfg4(R, E)
dw = batch.DeviceWorkload(w, dev)
r = batch.EkfRunner(dw, lane_block="auto")
for _ in range(3):
    r.run()
torch.cuda.synchronize()
# the scoring step's inputs, as bench.py prepares them
n, Bc, th = w.n_npi, w.B, w.meta.get("T_hist", w.T)
sp = torch.zeros((batch.SIM_PRM_COUNT, Bc), dtype=torch.float64, device=dev)
prm = dw.prm
sp[3], sp[4], sp[5] = prm[L_.PRM_ALPHA_MIN], prm[L_.PRM_ALPHA_MAX], prm[L_.PRM_GAMMA]
sp[6], sp[7], sp[11] = prm[L_.PRM_B], prm[L_.PRM_BETA], 1.0
sp[batch.SIM_A:batch.SIM_A + n] = prm[L_.PRM_A:L_.PRM_A + n]
sp[batch.SIM_U_MAX:batch.SIM_U_MAX + n] = prm[L_.PRM_U_MAX:L_.PRM_U_MAX + n]
sp[batch.SIM_W:batch.SIM_W + n] = 1.0
S = r.unblocked("S_SMOOTH")
sp[0:3].copy_(r.unblocked_at("S_SMOOTH", th - 1)[0:3])
J0p = (S[:th, 0] * S[:th, 1] * S[:th, 2]).sum(dim=0)
J1p = r.unblocked("u_opt_smooth")[:th].sum(dim=(0, 1))


def timed(fn, k=20):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        r.run(stream=s)
for rep in range(3):
    print("%d chains x %d days: filter %.3f ms, graph replay of it %.3f, sweep call %.3f, sweep call + Pareto filter %.3f" % (
        w.B, w.T, timed(r.run), timed(g.replay), timed(lambda: r.run_sweep(th, sp, J0p, J1p)),
        timed(lambda: r.run_sweep(th, sp, J0p, J1p, n_regions=w.Sx))))

# transient: groups of five back-to-back sweep calls after an idle second, wall clock per call
time.sleep(1.0)
out = []
for grp in range(10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        r.run_sweep(th, sp, J0p, J1p, n_regions=w.Sx)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / 5 * 1e3)
print("after 1 s idle, ms per call in groups of five:", " ".join("%.3f" % v for v in out))
