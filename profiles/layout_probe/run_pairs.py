"""python profiles/layout_probe/run_pairs.py -- the headline's forward store pattern with arithmetic between the stores, one
wave per SIMD: 104 stores of 8 B per lane and step against 52 of 16 B (row pairs interleaved).  (measurement only)"""
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
B, T, blk = 75000, 520, 40
nblk = (B + blk - 1) // blk
rows = [6, 6, 36, 36, 6, 2, 2, 12]
st = torch.cuda.current_stream()
arrs = [torch.empty(T * nblk * r * blk, dtype=torch.float64, device="cuda:0") for r in rows]
ptrs = (C.c_void_p * 8)(*[a.data_ptr() for a in arrs])
xin = torch.rand(256, dtype=torch.float64, device="cuda:0")
gb = nblk * blk * T * sum(rows) * 8 / 1e9
for lds, occ in ((40 * 1024 - 256, "1 wave/SIMD"), (0, "all resident")):
    for work in (0, 400, 800, 1600):
        for pair in (0, 1, 2):
            ts = []
            for _ in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if pair < 2:
                    rc = h.run_probeP(ptrs, C.c_void_p(xin.data_ptr()), B, T, blk, work, pair, lds, C.c_void_p(st.cuda_stream))
                else:       # the shipped layout, rows transposed through LDS into 16-byte stores
                    rc = h.run_probeT(ptrs, C.c_void_p(xin.data_ptr()), B, T, blk, work, max(lds, 36 * blk * 8), C.c_void_p(st.cuda_stream))
                b.record()
                torch.cuda.synchronize(); assert rc == 0
                ts.append(a.elapsed_time(b))
            t = float(np.median(ts[1:]))
            what = [" 8 B/lane, 106 stores", "16 B/lane, 53 stores (row pairs interleaved)", "16 B/lane, 53 stores (shipped layout, through LDS)"][pair]
            print(f"{occ:13s} FMAs/step {work:4d}  {what:50s} {t:6.2f} ms = {gb / t:.2f} TB/s", flush=True)
