"""python profiles/layout_probe/run.py  -- see layout_probe.hip (measurement only, not part of the product)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from epidemicmodeling_amd import _lib  # noqa: E402  (loads torch's HIP runtime first)

so = os.path.join(HERE, "layout_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared",
                           os.path.join(HERE, "layout_probe.hip"), "-o", so])
_lib._preload_torch_hip_runtime()
h = C.CDLL(so)
B, T, rows, blk = 75000, 520, 104, 40
nblk = (B + blk - 1) // blk
out = torch.empty(T * rows * (B + 512), dtype=torch.float64, device="cuda:0")
st = torch.cuda.current_stream()
def run(mode, lb, label):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = h.run_probe(C.c_void_p(out.data_ptr()), B, T, rows, blk, mode, 0, lb, C.c_void_p(st.cuda_stream)); b.record()
        torch.cuda.synchronize(); assert rc == 0
        ts.append(a.elapsed_time(b))
    gb = B * T * rows * 8 / 1e9
    print(f"{label:28s} median {np.median(ts[1:]):.2f} ms = {gb / np.median(ts[1:]) * 1e3:.0f} GB/s (min {min(ts[1:]):.2f})")


run(0, 0, "[t][row][B]")
run(1, 0, "[t][B/40][row][40]")
run(3, 0, "[t][B/40][row/2][40][2] 16B")
for lb in (8, 16, 32, 64, 128, 256):
    run(2, lb, f"[t][B/{lb}][row][{lb}]")
