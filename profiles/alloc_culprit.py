"""Which array's PLACEMENT makes a slow allocation slow?  (profiles/alloc_probe.py: with one library on one box the forward
stage takes 5.9 or 7.0 ms and the smoother 7.1 or 8.0 ms depending on where the allocator put the arrays.)
Allocates the runner's arrays until a slow placement turns up, then replaces ONE array at a time by a fresh allocation
(the old one kept alive, so the new one lies elsewhere) and times the stages again.
    python profiles/alloc_culprit.py [max_trials]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicmodeling_amd import batch, synth  # noqa: E402
from epidemicmodeling_amd.batch import _ptr  # noqa: E402


def stage_times(r, reps=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f, b = [], []
    for rep in range(reps + 1):
        ev[0].record(); r.run(phase=1); ev[1].record(); r.run(phase=3); ev[2].record(); r.run(phase=4); ev[3].record()
        torch.cuda.synchronize()
        if rep:
            f.append(ev[0].elapsed_time(ev[1])); b.append(ev[2].elapsed_time(ev[3]))
    return round(float(np.median(f)), 3), round(float(np.median(b)), 3)


def main():
    max_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    w = synth.make_cfg4()
    dw = batch.DeviceWorkload(w, dev)
    rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
    log = {"trials": [], "replacements": []}
    keep = []
    for tr in range(max_trials):
        torch.cuda.empty_cache()
        pad = torch.empty(int(rng.integers(1, 2048)) << 20, dtype=torch.uint8, device=dev)
        r = batch.EkfRunner(dw, extras=False, lane_block="auto", shape="auto")
        f, b = stage_times(r)
        log["trials"].append((f, b))
        print("trial", tr, f, b, flush=True)
        if f >= 6.6 or b >= 7.7:
            base = (f, b)
            for n in list(r.out) + ["ws"]:
                old = r.ws if n == "ws" else r.out[n]
                if old.numel() * old.element_size() < (1 << 28):
                    continue
                new = torch.empty_like(old)
                if n == "ws":
                    r.ws = new
                else:
                    r.out[n] = new
                    setattr(r.outs, n, _ptr(new))
                keep.append(old)
                f2, b2 = stage_times(r)
                log["replacements"].append({"trial": tr, "array": n, "before": base, "after": (f2, b2)})
                print("   replaced", n, "->", f2, b2, flush=True)
                base = (f2, b2)
                free, total = torch.cuda.mem_get_info()
                if free < (40 << 30):
                    keep.clear(); torch.cuda.empty_cache()
            break
        del r, pad
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
    json.dump(log, open(os.path.join(ROOT, "gpurun_out", "r04", "alloc_culprit_%s.json" % os.environ.get("SEED", "0")), "w"), indent=1)


if __name__ == "__main__":
    main()
