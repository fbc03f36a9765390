"""python profiles/layout_probe/run_cfg5_occ.py -- BASELINE config 5's forward store pattern under the real kernel's
constraints: resident waves capped by LDS (all / 4 / 3 / 2 / 1 per SIMD), with and without one awaited vector load per step,
with and without ~350 dependent FMAs per step.  (measurement only)"""
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
B, T = 307200, 400
nblk = B // 64
st = torch.cuda.current_stream()
arrs = [(3, 4), (3, 4), (9, 4), (9, 4), (3, 4), (1, 4), (1, 4), (12, 4), (3, 8), (6, 8), (6, 8)]
bufs = [torch.empty(T * nblk * r * 64 * e, dtype=torch.uint8, device="cuda:0") for r, e in arrs]
n = len(arrs)
ptrs = (C.c_void_p * n)(*[b.data_ptr() for b in bufs])
rows = (C.c_int * n)(*[r for r, _ in arrs]); es = (C.c_int * n)(*[e for _, e in arrs])
xin = torch.rand(T * B, dtype=torch.float64, device="cuda:0")
gb = B * T * sum(r * e for r, e in arrs) / 1e9
# per-workgroup LDS that leaves w waves per SIMD (4 SIMDs, 160 KB per CU; 64 KB is the default dynamic limit)
occ = {"all": 0, "4/SIMD": 10 * 1024 - 256, "3/SIMD": 13 * 1024, "2/SIMD": 20 * 1024 - 256, "1/SIMD": 40 * 1024 - 256}
for wait in (0, 1):
    for work in (0, 350):
        for name, lds in occ.items():
            ts = []
            for _ in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = h.run_probeNW(ptrs, rows, es, n, C.c_void_p(xin.data_ptr()), B, T, work, wait, lds, C.c_void_p(st.cuda_stream))
                b.record(); torch.cuda.synchronize(); assert rc == 0
                ts.append(a.elapsed_time(b))
            t = float(np.median(ts[1:]))
            print(f"awaited load per step: {wait}   FMAs per step: {work:3d}   waves {name:7s}  {t:6.2f} ms = {gb / t:.2f} TB/s", flush=True)
