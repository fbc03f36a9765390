"""python profiles/layout_probe/run_cfg5.py [trials] -- BASELINE config 5's forward store pattern (307 200 chains x 400 days,
fp32 outputs + fp64 workspace) as the arrays are today against fused variants, over fresh allocations (measurement only)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from epidemicmodeling_amd import _lib  # noqa: E402

so, src = os.path.join(HERE, "layout_probe.so"), os.path.join(HERE, "layout_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", src, "-o", so])
_lib._preload_torch_hip_runtime()
h = C.CDLL(so)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, T = 307200, 400
nblk = B // 64
st = torch.cuda.current_stream()
F32 = [(3, 4), (3, 4), (9, 4), (9, 4), (3, 4), (1, 4), (1, 4), (12, 4)]       # S-, S+, P-, P+, K, innov, rho, u_opt
VARIANTS = {
    "today: 8 fp32 outputs + 3 fp64 workspace arrays (3, 6, 6 rows)": F32 + [(3, 8), (6, 8), (6, 8)],
    "workspace as one 15-row record": F32 + [(15, 8)],
    "fp32 outputs as one 41-row record + workspace record": [(41, 4), (15, 8)],
    "fp32 outputs alone (8 arrays)": F32,
    "workspace alone (3 arrays)": [(3, 8), (6, 8), (6, 8)],
}


def timed(fn):
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = fn(); b.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))


res = {k: [] for k in VARIANTS}
rng = np.random.default_rng(0)
for tr in range(trials):
    for name, arrs in VARIANTS.items():
        torch.cuda.empty_cache()
        pad = torch.empty(int(rng.integers(1, 1024)) << 20, dtype=torch.uint8, device="cuda:0")
        bufs = [torch.empty(T * nblk * r * 64 * e, dtype=torch.uint8, device="cuda:0") for r, e in arrs]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[b.data_ptr() for b in bufs])
        rows = (C.c_int * n)(*[r for r, _ in arrs]); es = (C.c_int * n)(*[e for _, e in arrs])
        res[name].append(timed(lambda: h.run_probeN(ptrs, rows, es, n, B, T, C.c_void_p(st.cuda_stream))))
        del bufs, pad
    print(tr, {k[:24]: round(v[-1], 2) for k, v in res.items()}, flush=True)
for name, arrs in VARIANTS.items():
    v = np.array(res[name]); gb = B * T * sum(r * e for r, e in arrs) / 1e9
    print(f"{name:66s} {gb:5.1f} GB  median {np.median(v):5.2f} ms = {gb / np.median(v):.2f} TB/s  (min {v.min():.2f}, max {v.max():.2f})")
