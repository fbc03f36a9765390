#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop from hipcc's -S output (see tools/kernel_resources.py for the compile line).

    python tools/isa_mix.py /tmp/epiekf.s 'ekf_fwd_quad<0, 16, 21, 0, 1>' [--dump] [--ops] [--inner]

Finds the function, takes the largest backward-branch span as the loop and counts instructions by class."""
import collections
import re
import subprocess
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    # function starts: "<mangled>:" following a .type ...,@function
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    names = subprocess.run(["c++filt"] + [s[1] for s in starts], capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.split("\n")
    sel = [(s, n) for s, n in zip(starts, names) if want in n.replace("epi::", "")]
    if not sel:
        sys.exit("no kernel matches " + want)
    (i0, mangled), name = sel[0]
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[i0:i1]
    labels = {m.group(1): k for k, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    # the compiler marks loop headers ("Loop Header") and the blocks of a loop ("in Loop: Header=BBn_m"): the outermost
    # loop with the largest extent, from its header to the end of its last block
    best = (0, 0, 0)
    inner = "--inner" in sys.argv          # the largest INNER loop instead (a kernel whose day loop sits inside a window loop)
    for lab, a in labels.items():
        head = " ".join(body[a:a + 4])       # "Parent Loop ..." and "=> This Inner Loop Header" follow the label on their own lines
        if "Loop Header" not in head.split(".LBB", 2)[1] if head.count(".LBB") > 1 else "Loop Header" not in head:
            continue
        if inner != ("Parent Loop" in head):
            continue
        tag = "Header=" + lab[2:]
        blocks = [k for k, l in enumerate(body) if k > a and tag in l and re.match(r"^\.LBB", l)]
        last = blocks[-1] if blocks else a
        end = next((k for k in range(last + 1, len(body)) if re.match(r"^\.LBB", body[k])), len(body)) - 1
        if end - a > best[0]:
            best = (end - a, a, end)
    _, a, b = best
    loop = [l.strip() for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    cnt = collections.Counter()
    ops = collections.Counter()
    for l in loop:
        op = l.split()[0]
        ops[op] += 1
        if "dpp" in l and op.startswith("v_"): c = "valu dpp"
        elif op.startswith("v_div_") or op.startswith("v_rcp") or op.startswith("v_sqrt") or op.startswith("v_rsq"): c = "valu div/rcp"
        elif op.startswith("v_fma_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64"): c = "valu fp64 arith"
        elif op.startswith("v_accvgpr"): c = "valu agpr move"
        elif op.startswith("v_cndmask") or op.startswith("v_cmp") : c = "valu cmp/select"
        elif op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): c = "valu lane<->sgpr"
        elif op.startswith("v_mov"): c = "valu mov"
        elif op.startswith("v_"): c = "valu other"
        elif op.startswith("s_waitcnt"): c = "s_waitcnt"
        elif op.startswith("s_nop"): c = "s_nop"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): c = "branch"
        elif op.startswith("s_"): c = "salu"
        elif op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load") or op.startswith("scratch_load"): c = "vmem load"
        elif op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("flat_store") or op.startswith("scratch_store"): c = "vmem store"
        elif op.startswith("ds_"): c = "lds"
        else: c = "other"
        cnt[c] += 1
    print("%s\n  loop: %d instructions (lines %d..%d of the function)" % (name.split("(")[0], len(loop), a, b))
    for c, n in sorted(cnt.items(), key=lambda x: -x[1]):
        print("  %-18s %5d" % (c, n))
    print("  VALU total %d" % sum(n for c, n in cnt.items() if c.startswith("valu")))
    if "--ops" in sys.argv:
        for o, n in ops.most_common(40):
            print("    %-28s %4d" % (o, n))
    if dump:
        print("\n".join(loop))


if __name__ == "__main__":
    main()
