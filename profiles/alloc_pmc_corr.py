"""Correlation of each counter with the kernel's duration over the dispatches of a profiles/alloc_probe.py run (output of
profiles/alloc_pmc_join.py).  python profiles/alloc_pmc_corr.py JOIN.txt"""
import re
import sys

import numpy as np

rows, cur = [], None
for l in open(sys.argv[1]):
    if l.startswith("ekf_fwd_sym") or l.startswith("eks_bwd_sym"):
        cur = l.split()[0]
        continue
    m = re.match(r"\s+([\d.]+) ms\s+(.*)", l)
    if m:
        rows.append((cur, float(m.group(1)), {kv.split("=")[0]: float(kv.split("=")[1]) for kv in m.group(2).split()}))
for k in ("ekf_fwd_sym", "eks_bwd_sym"):
    r = [x for x in rows if x[0] == k]
    t = np.array([x[1] for x in r])
    print(k, len(r), "dispatches, duration min / median / max %.3f / %.3f / %.3f ms" % (t.min(), np.median(t), t.max()))
    for c in r[0][2]:
        v = np.array([x[2][c] for x in r])
        cc = np.corrcoef(t, v)[0, 1] if v.std() > 0 else float("nan")
        fast, slow = v[t <= np.percentile(t, 25)].mean(), v[t >= np.percentile(t, 90)].mean()
        print("   %-36s corr %+.2f   fastest quarter %.4g   slowest tenth %.4g" % (c, cc, fast, slow))
