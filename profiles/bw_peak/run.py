"""python profiles/bw_peak/run.py -- write-only / read-only / copy bandwidth of this box for 8 and 16 B per lane, default and
non-temporal policy (bw_peak.hip; measurement only).  GB/s of bytes moved (copy: read + written)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from epidemicmodeling_amd import _lib  # noqa: E402

so, src = os.path.join(HERE, "bw_peak.so"), os.path.join(HERE, "bw_peak.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", src, "-o", so])
_lib._preload_torch_hip_runtime()
h = C.CDLL(so)
h.bw_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
N = 1 << 32                                    # 4 GiB per buffer: far beyond the 256 MiB Infinity Cache
a = torch.empty(N, dtype=torch.uint8, device="cuda:0"); b = torch.empty_like(a); a.fill_(1)
sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream()
h.bw_run_tile.argtypes = h.bw_run.argtypes
for grid, fn in (("2 048 persistent workgroups, grid-stride", h.bw_run), ("one-shot grid, a 4 x 256-element tile per workgroup", h.bw_run_tile)):
    print(grid)
    for kind, name, mult in ((0, "write only", 1), (1, "read only", 1), (2, "copy (read + write)", 2)):
        for width in (8, 16):
            for nt in (0, 1):
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(kind, width, nt, a.data_ptr(), b.data_ptr(), N, sink.data_ptr(), st.cuda_stream)
                    e1.record(); torch.cuda.synchronize(); assert rc == 0
                    ts.append(e0.elapsed_time(e1))
                print(f"  {name:20s} {width:2d} B/lane {'nt     ' if nt else 'default'}  {mult * N / (min(ts[1:]) * 1e-3) / 1e9:7.0f} GB/s", flush=True)
# the library kernels torch ships, on the same buffers
af, bf = a.view(torch.float64), b.view(torch.float64)
for name, op, mult in (("torch fill_", lambda: bf.fill_(1.0), 1), ("torch copy_", lambda: bf.copy_(af), 2), ("torch max()", lambda: af.max(), 1)):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); op(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:24s} {mult * N / (min(ts[1:]) * 1e-3) / 1e9:7.0f} GB/s", flush=True)
