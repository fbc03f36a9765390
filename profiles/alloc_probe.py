"""Does the forward kernel's time depend on WHERE its output arrays lie?  The headline's ekf_fwd stage measures 6.05 or
7.05 ms in different processes on one box with one library.  Here one process allocates the runner's arrays several times
(a dummy allocation of varying size in front moves every base address), times the forward stage alone with HIP events, and
prints the addresses modulo a few powers of two next to the time.
    python profiles/alloc_probe.py [trials]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicmodeling_amd import batch, synth  # noqa: E402


# SLAB="align,stagger" (bytes): one allocation for all arrays, array i at a multiple of align plus i * stagger
SLAB = None
if os.environ.get("SLAB"):
    _a, _s = os.environ["SLAB"].split(",")
    SLAB = {"align": int(_a), "stagger": int(_s)}


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    w = synth.make_cfg4()
    dw = batch.DeviceWorkload(w, dev)
    rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
    rows = []
    for tr in range(trials):
        torch.cuda.empty_cache()
        pad_mb = 0 if tr == 0 else int(rng.integers(1, 4096))
        pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
        r = batch.EkfRunner(dw, extras=False, lane_block="auto", shape="auto", slab=SLAB)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ts = {"fwd": [], "pinv": [], "bwd": []}
        for rep in range(6):
            ev[0].record(); r.run(phase=1); ev[1].record(); r.run(phase=3); ev[2].record(); r.run(phase=4); ev[3].record()
            torch.cuda.synchronize()
            if rep:
                ts["fwd"].append(ev[0].elapsed_time(ev[1])); ts["pinv"].append(ev[1].elapsed_time(ev[2])); ts["bwd"].append(ev[2].elapsed_time(ev[3]))
        # per-array streaming rates of the same placement: a plain fill (write) and a plain read of every array
        rates = {}
        arrs = dict(r.out); arrs["ws"] = r.ws
        for n, t in arrs.items():
            if t.numel() * t.element_size() < (1 << 30) or os.environ.get("NOFILL"):
                continue
            flat = t.view(-1)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            best_w = best_r = 1e9
            for _ in range(3):
                e0.record(); flat.fill_(0.0); e1.record(); _ = flat.max(); e2.record(); torch.cuda.synchronize()
                best_w = min(best_w, e0.elapsed_time(e1)); best_r = min(best_r, e1.elapsed_time(e2))
            gb = t.numel() * t.element_size() / 1e9
            rates[n] = (round(gb / best_w, 2), round(gb / best_r, 2))     # TB/s (GB per ms)
        ptr = {n: t.data_ptr() for n, t in r.out.items()}
        ptr["ws"] = r.ws.data_ptr()
        row = {"trial": tr, "pad_MiB": pad_mb, "fwd_ms": round(float(np.median(ts["fwd"])), 3), "pinv_ms": round(float(np.median(ts["pinv"])), 3),
               "bwd_ms": round(float(np.median(ts["bwd"])), 3), "fill_read_TBps": rates,
               "ptr_mod_1GiB_in_MiB": {n: (p % (1 << 30)) >> 20 for n, p in ptr.items()},
               "ptr_hex": {n: hex(p) for n, p in ptr.items()}}
        rows.append(row)
        print(os.environ.get("SLAB", "separate"), json.dumps({k: row[k] for k in ("trial", "fwd_ms", "bwd_ms", "fill_read_TBps")}), flush=True)
        del r, pad
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04", "alloc_probe_%s_%s.json" % (os.environ.get("SLAB", "separate").replace(",", "_"), os.environ.get("SEED", "0"))), "w"), indent=1)


if __name__ == "__main__":
    main()
