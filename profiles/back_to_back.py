"""Is the B = 1 per-day latency a clock artefact?  The same one-chain call timed (a) alone with a synchronisation after each
(the GPU idles between calls), (b) 50 calls enqueued back to back, (c) alone while a long-running kernel of another stream
keeps the chip busy.   python profiles/back_to_back.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth
full = synth.make_cfg4(2, 250, 400, 120)
w = full.select(np.array([137]))
for shape in ("wave", "quad"):
    r = batch.EkfRunner(batch.DeviceWorkload(w, "cuda:0"), lane_block="auto", shape=shape)
    r.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r.run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    alone = float(np.median(ts))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        r.run()
    b.record(); torch.cuda.synchronize()
    b2b = a.elapsed_time(b) / 50
    side = torch.cuda.Stream()
    big = torch.rand(1 << 28, dtype=torch.float64, device="cuda:0")
    ts = []
    for _ in range(5):
        with torch.cuda.stream(side):
            for _ in range(6):
                big.mul_(1.0000001)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r.run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    busy = float(np.median(ts))
    print(f"{shape}: one call {alone:.3f} ms; 50 back to back {b2b:.3f} ms each; one call beside a streaming kernel {busy:.3f} ms", flush=True)
    del big
