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
for kind, name, mult in ((0, "write only", 1), (1, "read only", 1), (2, "copy (read + write)", 2)):
    for width in (8, 16):
        for nt in (0, 1):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = h.bw_run(kind, width, nt, a.data_ptr(), b.data_ptr(), N, sink.data_ptr(), st.cuda_stream)
                e1.record(); torch.cuda.synchronize(); assert rc == 0
                ts.append(e0.elapsed_time(e1))
            print(f"{name:20s} {width:2d} B/lane {'nt     ' if nt else 'default'}  {mult * N / (min(ts[1:]) * 1e-3) / 1e9:7.0f} GB/s", flush=True)
