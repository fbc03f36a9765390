"""What the host-pointer entry points can expect from the box: device-to-host copy rates into pinned and pageable memory,
and the host's own memcpy rate (the staged download copies pinned -> caller's arrays).  Prints one JSON object."""
import json
import time

import numpy as np
import torch


def rate(fn, nbytes, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return nbytes / best / 1e9


def main():
    out = {}
    dev = torch.device("cuda:0")
    for mb in (1, 16, 64, 256):
        n = mb << 20
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        hp = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        hq = torch.empty(n, dtype=torch.uint8)
        hq.fill_(1)
        out[f"{mb}MiB"] = {
            "d2h_pinned_GBps": round(rate(lambda: hp.copy_(d, non_blocking=True), n), 2),
            "d2h_pageable_GBps": round(rate(lambda: hq.copy_(d), n), 2),
            "h2d_pinned_GBps": round(rate(lambda: d.copy_(hp, non_blocking=True), n), 2),
            "h2d_pageable_GBps": round(rate(lambda: d.copy_(hq), n), 2),
        }
        a = np.frombuffer(hp.numpy(), dtype=np.uint8)
        b = hq.numpy()
        t = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            np.copyto(b, a)
            t = min(t, time.perf_counter() - t0)
        out[f"{mb}MiB"]["host_memcpy_pinned_to_pageable_GBps"] = round(n / t / 1e9, 2)
    import os
    out["cpus"] = os.cpu_count()
    out["affinity"] = len(os.sched_getaffinity(0))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
