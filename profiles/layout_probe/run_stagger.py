"""python profiles/layout_probe/run_stagger.py -- the forward kernel's store pattern into eight arrays carved out of ONE
allocation, array k starting k * step bytes later than in the dense packing, for a range of steps: can a caller who owns one
slab pick the placement instead of taking what the allocator gives?  Each configuration timed twice (reproducibility); the
whole sweep repeated on a second, fresh slab.  (measurement only)"""
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
rows = [6, 6, 36, 36, 6, 1, 1, 12]
sizes = [T * nblk * r * blk * 8 for r in rows]
st = torch.cuda.current_stream()


def timed(fn):
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = fn(); b.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))


steps = [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 8 << 20, 13 << 20, 16 << 20, 21 << 20, 32 << 20, 34 << 20, 55 << 20, 64 << 20, 89 << 20, 128 << 20]
for slab_no in range(2):
    torch.cuda.empty_cache()
    pad = torch.empty((slab_no * 777 + 1) << 20, dtype=torch.uint8, device="cuda:0")
    slab = torch.empty(sum(sizes) + 8 * max(steps) + (64 << 20), dtype=torch.uint8, device="cuda:0")
    base = slab.data_ptr()
    base = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    out = []
    for step in steps:
        off, ptrs = 0, []
        for k, sz in enumerate(sizes):
            ptrs.append(base + off + k * step)
            off += (sz + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        arr = (C.c_void_p * 8)(*ptrs)
        t1 = timed(lambda: h.run_probe8(arr, B, T, blk, 0, C.c_void_p(st.cuda_stream)))
        t2 = timed(lambda: h.run_probe8(arr, B, T, blk, 0, C.c_void_p(st.cuda_stream)))
        out.append((step, t1, t2))
        print(f"slab {slab_no}  step {step / (1 << 20):9.3f} MiB   {t1:.2f}  {t2:.2f} ms", flush=True)
    del slab, pad
