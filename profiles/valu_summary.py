"""Turns one rocprofv3 SQ-counter pass of profiles/traffic_probe.py into per-kernel VALU occupancy figures.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o p -- python3 profiles/traffic_probe.py
    python3 profiles/valu_summary.py gpurun_out/pmc_sq profiles/r01/valu_summary.json

SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* count quad-cycles summed over all waves; GRBM_GUI_ACTIVE counts clock cycles summed
over the 8 XCDs.  Derived per launch:
  valu_active_per_wave  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of a resident wave's time in which it has a
                          VALU instruction executing)
  simd_valu_busy        = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (share of the kernel's
                          duration in which an average SIMD's VALU is busy, launch ramp and tail included)
  waves_per_simd        = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   (average resident waves per SIMD)"""
import csv
import json
import sys
from collections import defaultdict

N_XCD, N_SIMD = 8, 1024


def main(pmc_dir, out):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(pmc_dir + "/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "epi::" not in k:
            continue
        k = k.replace("void ", "").split("(")[0].replace("epi::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, v in acc.items():
        c = {n: sum(x) / len(x) for n, x in v.items()}
        if "SQ_WAVE_CYCLES" not in c or c.get("GRBM_GUI_ACTIVE", 0) < 1e6:
            continue
        simd_cycles = c["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD
        res[k] = {"launches": len(v["SQ_WAVE_CYCLES"]), **{n: c[n] for n in sorted(c)},
                  "valu_active_per_wave": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                  "wait_any_per_wave": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                  "simd_valu_busy": 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                  "waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / simd_cycles,
                  "valu_insts_per_launch": c["SQ_INSTS_VALU"]}
    json.dump({"kernels": res}, open(out, "w"), indent=1)
    print(json.dumps({"kernels": res}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
