#!/usr/bin/env python3
"""Order of vector-memory operations and vmcnt waits in a kernel's loop, from hipcc's -S output (see tools/kernel_resources.py for
the compile line):

    python tools/vm_trace.py /tmp/epiekf.s '_ZN3epi11ekf_fwd_hexILi0ELi10ELi1EEEvNS_5KArgsEPKi:' [FIRST LAST]

prints e.g. `LDx4 W21 STx7 | W27 STx9 W35 | W34 | STx2`: LD / ST / DMA (`buffer_load ... lds`) runs, Wn = `s_waitcnt vmcnt(n)`,
`|` = a branch (operations behind it are on some paths only: the wait-count pass can rely on the operations of the path with the
fewest).  Without FIRST LAST (instruction lines relative to the symbol) the largest backward-branch span is taken."""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    start = [i for i, l in enumerate(lines) if l.startswith(want)][0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    print("backward branches (first, last):", sorted(loops, key=lambda x: x[0] - x[1])[:6], "of", len(body), "lines")
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else max(loops, key=lambda x: x[1] - x[0])
    ops = []
    for i in range(lo, hi + 1):
        l = body[i].strip()
        if l.startswith("buffer_load") and " lds" in l:
            ops.append("DMA")
        elif l.startswith(("buffer_load", "global_load")):
            ops.append("LD")
        elif l.startswith(("buffer_store", "global_store")):
            ops.append("ST")
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            ops.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
        elif l.startswith(("s_cbranch", "s_branch")):
            ops.append("|")
    out, prev, cnt = [], None, 0
    for o in ops:
        if o == prev and not o.startswith("W"):
            cnt += 1
        else:
            if prev is not None:
                out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
            prev, cnt = o, 1
    out.append(f"{prev}x{cnt}" if cnt > 1 else str(prev))
    print(" ".join(out))


if __name__ == "__main__":
    main()
