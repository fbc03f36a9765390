#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's resource remarks.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/epiekf.s \
          epidemicmodeling_amd/csrc/epiekf.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/remarks.txt
    python tools/kernel_resources.py /tmp/remarks.txt [substring ...]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
    rows = []
    for b in blocks[1:]:
        name = b.split("\n")[0].strip().split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g("SGPRs"), g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.split("\n")
    print("%-58s %5s %5s %8s %4s %5s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPR", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", n)).replace("epi::", "")
        if want and not any(w in n for w in want):
            continue
        print("%-58s %5d %5d %8d %4d %5d %6d" % (n[:58], *r[1:]))


if __name__ == "__main__":
    main()
