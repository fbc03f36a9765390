#!/usr/bin/env python3
"""Which VGPRs a kernel's hot loop only reads (loop-invariant live-ins) and which it writes:
    python tools/vgpr_live.py /tmp/epiekf.s 'ekf_fwd_sym<3, 0, 0, 1, 0, 0>'"""
import re, subprocess, sys
out = subprocess.run([sys.executable, __file__.replace("vgpr_live.py", "isa_mix.py"), sys.argv[1], sys.argv[2], "--dump"],
                     capture_output=True, text=True, timeout=200).stdout.split("\n")
start = next(i for i, l in enumerate(out) if l.startswith("  VALU total")) + 1
txt = out[start:]
rx = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")
def regs(s):
    r = set()
    for m in rx.finditer(s):
        if m.group(1): r.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: r.add(int(m.group(3)))
    return r
used, written, wl = set(), set(), set()
for l in txt:
    parts = l.split(None, 1)
    if len(parts) < 2: continue
    op, args = parts
    used |= regs(args)
    if op.startswith(("buffer_store", "global_store", "ds_write", "s_", "v_cmp", "scratch_store")) and not op.startswith("v_cmpx"):
        if op.startswith("v_readlane") is False: pass
        continue
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        continue
    first = args.split(",")[0]
    written |= regs(first)
    if op.startswith("v_writelane"): wl |= regs(first)
print(out[0])
print("VGPRs referenced in loop %d, written %d (of them SGPR-spill slots %d), read-only %d" % (len(used), len(written), len(wl), len(used - written)))
rl = set()
for l in txt:
    if l.startswith("v_readlane"):
        rl |= regs(l.split(",")[1])
print("read-only VGPRs that are v_readlane sources (SGPR spill slots): %d" % len((used - written) & rl))
