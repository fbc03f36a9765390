"""python profiles/layout_probe/run_rw.py  -- the smoother's read/write pattern, 8 B vs 16 B per lane (layout_probe.hip, probe_rw;
measurement only, not part of the product)."""
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
src = os.path.join(HERE, "layout_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", src, "-o", so])
_lib._preload_torch_hip_runtime()
h = C.CDLL(so)
B, T, rr, wr, blk = 75000, 520, 76, 54, 40
nblk = (B + blk - 1) // blk
inp = torch.rand(T * rr * nblk * blk, dtype=torch.float64, device="cuda:0")
out = torch.empty(T * wr * nblk * blk, dtype=torch.float64, device="cuda:0")
st = torch.cuda.current_stream()


def run(rmode, wmode, work, label):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = h.run_probe_rw(C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), B, T, rr, wr, blk, rmode, wmode, work, C.c_void_p(st.cuda_stream))
        b.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(a.elapsed_time(b))
    gb = B * T * (rr + wr) * 8 / 1e9
    print(f"{label:44s} work {work:5d}  median {np.median(ts[1:]):.2f} ms = {gb / np.median(ts[1:]) * 1e3:.0f} GB/s (min {min(ts[1:]):.2f})", flush=True)


print(f"eks_bwd-like pattern: {B} chains x {T} steps, {rr} doubles read + {wr} written per chain and step = {B * T * (rr + wr) * 8 / 1e9:.1f} GB, one wave per SIMD")
for work in (0, 600, 1400):
    run(0, 0, work, "reads 8 B/lane, writes 8 B/lane")
    run(1, 0, work, "reads 16 B/lane, writes 8 B/lane")
    run(1, 1, work, "reads 16 B/lane, writes 16 B/lane")
