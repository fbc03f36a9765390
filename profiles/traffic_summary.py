"""Turns the two rocprofv3 PMC passes of profiles/traffic_probe.py into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are in KiB.  Correction: the calibration copy moved a KNOWN 8*N bytes each way with the
same 8 B/lane access shape, so  factor_read = 8N / (FETCH_SIZE_calib * 1024)  (the guide measures exactly 2.0 for
16 B/lane streams on gfx950) and likewise for writes; every kernel's raw counter is multiplied by that factor."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import _build  # noqa: E402

CALIB_DOUBLES = 1 << 29


def per_kernel(path, counter):
    acc, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path + "/p_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        acc[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}       # KiB per launch


def main(fetch_dir, write_dir, out):
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    calib = [k for k in f if "calib_copy" in k][0]
    known = 8.0 * CALIB_DOUBLES
    fr, fw = known / (f[calib] * 1024.0), known / (w[calib] * 1024.0)
    res = {"kernel_src_sha16": _build.source_hash(),     # bench.py quotes these numbers only for the same kernel sources
           "calibration": {"bytes_each_way": known, "FETCH_SIZE_KiB": f[calib], "WRITE_SIZE_KiB": w[calib],
                           "read_factor": fr, "write_factor": fw}, "kernels": {}}
    for k in f:
        if "epi::" not in k or "calib" in k or "precheck" in k:
            continue
        short = k.replace("void epi::", "").split("(")[0]
        rd, wr = f[k] * 1024.0 * fr, w.get(k, 0.0) * 1024.0 * fw
        res["kernels"][short] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                                 "raw_FETCH_SIZE_KiB": f[k], "raw_WRITE_SIZE_KiB": w.get(k, 0.0)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
