"""python profiles/layout_probe/run_rows.py -- store bandwidth of the forward kernels' pattern as a function of the rows one
wave writes contiguously per step ([t][B/blk][rows][blk], kernel `probe` mode 1): the three-state kernels write 3- and 9-row
arrays (1.3 / 4 KB per wave and step), the six-state ones 6- and 36-row arrays (1.9 / 11.5 KB).  Measurement only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from epidemicmodeling_amd import _lib  # noqa: E402

so = os.path.join(HERE, "layout_probe.so")
src = os.path.join(HERE, "layout_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", src, "-o", so])
_lib._preload_torch_hip_runtime()
h = C.CDLL(so)
out = torch.empty(int(4.2e9), dtype=torch.float64, device="cuda:0")     # 33.6 GB
st = torch.cuda.current_stream()
for B, blk in ((75000, 40), (307200, 56), (307200, 64)):
    for rows in (1, 3, 9, 12, 41, 104):
        T = int(4.0e9 // (rows * ((B + blk - 1) // blk) * blk))
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = h.run_probe(C.c_void_p(out.data_ptr()), B, T, rows, blk, 1, 0, 0, C.c_void_p(st.cuda_stream)); b.record()
            torch.cuda.synchronize(); assert rc == 0
            ts.append(a.elapsed_time(b))
        gb = B * T * rows * 8 / 1e9
        print(f"B {B:6d} blk {blk:2d} rows {rows:3d} T {T:5d}: {gb:5.1f} GB in {np.median(ts[1:]):6.2f} ms = {gb / np.median(ts[1:]) * 1e3:5.0f} GB/s", flush=True)

# round 3, second question: 40 lanes per wave (what the six-state kernels want) in rows PADDED to 48 / 64 doubles (mode 4)
print("padded rows, 40 lanes per wave:")
B, blk = 75000, 40
for lb in (40, 48, 64):
    for rows in (1, 6, 12, 36, 104):
        T = int(4.0e9 // (rows * ((B + blk - 1) // blk) * lb))
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = h.run_probe(C.c_void_p(out.data_ptr()), B, T, rows, blk, 4, 0, lb, C.c_void_p(st.cuda_stream)); b.record()
            torch.cuda.synchronize(); assert rc == 0
            ts.append(a.elapsed_time(b))
        gb = B * T * rows * 8 / 1e9
        print(f"pitch {lb:2d} rows {rows:3d} T {T:5d}: {gb:5.1f} GB useful in {np.median(ts[1:]):6.2f} ms = {gb / np.median(ts[1:]) * 1e3:5.0f} GB/s", flush=True)
