"""python profiles/layout_probe/run_streams.py [trials] -- the forward kernel's store pattern (104 rows of 320 B per wave and
step) written into EIGHT separately allocated arrays, as today, against ONE array of 104-row records, each over fresh
allocations: does the spread between placements come with the number of concurrently written arrays?  (measurement only)"""
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
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, T, blk = 75000, 520, 40
nblk = (B + blk - 1) // blk
rows = [6, 6, 36, 36, 6, 1, 1, 12]
st = torch.cuda.current_stream()
rng = np.random.default_rng(0)
gb = nblk * blk * T * 104 * 8 / 1e9


def timed(fn):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = fn(); b.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))


res = {"eight arrays": [], "eight arrays, nt": [], "one record array": []}
for tr in range(trials):
    torch.cuda.empty_cache()
    pad = torch.empty(int(rng.integers(1, 2048)) << 20, dtype=torch.uint8, device="cuda:0")
    arrs = [torch.empty(T * nblk * r * blk, dtype=torch.float64, device="cuda:0") for r in rows]
    ptrs = (C.c_void_p * 8)(*[a.data_ptr() for a in arrs])
    res["eight arrays"].append(timed(lambda: h.run_probe8(ptrs, B, T, blk, 0, C.c_void_p(st.cuda_stream))))
    res["eight arrays, nt"].append(timed(lambda: h.run_probe8(ptrs, B, T, blk, 1, C.c_void_p(st.cuda_stream))))
    del arrs
    torch.cuda.empty_cache()
    one = torch.empty(T * nblk * 104 * blk, dtype=torch.float64, device="cuda:0")
    res["one record array"].append(timed(lambda: h.run_probe(C.c_void_p(one.data_ptr()), B, T, 104, blk, 1, 1, 0, C.c_void_p(st.cuda_stream))))
    del one, pad
    print(tr, {k: round(v[-1], 3) for k, v in res.items()}, flush=True)
for k, v in res.items():
    v = np.array(v)
    print(f"{k:18s} median {np.median(v):.2f} ms ({gb / np.median(v):.2f} TB/s)  min {v.min():.2f}  max {v.max():.2f}  spread {100 * (v.max() - v.min()) / v.min():.0f} %")
