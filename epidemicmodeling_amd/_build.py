"""In-tree build of libepiekf.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "epiekf.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "ekf_device.hpp"), os.path.join(HERE, "csrc", "ekf_sym.hpp"), os.path.join(HERE, "csrc", "ekf_quad.hpp"), os.path.join(HERE, "csrc", "ekf_wave.hpp"), os.path.join(HERE, "csrc", "ekf_hex.hpp"), os.path.join(HERE, "csrc", "ekf_lane6.hpp"),
        os.path.join(HERE, "csrc", "scenario_kernels.hpp"), os.path.join(HERE, "csrc", "rt_expfit.hpp"), os.path.join(HERE, "csrc", "preprocess.hpp"), os.path.join(HERE, "csrc", "nnls.hpp"),
        os.path.join(HERE, "..", "include", "epiekf.h"), os.path.join(HERE, "..", "include", "epiekf_layout.h")]
LIB = os.path.join(HERE, "libepiekf.so")

# -ffp-contract=off: the kernels' arithmetic contract is one IEEE rounding per written operation
# (DESIGN.md "Arithmetic contract"); the CPU oracle is built the same way.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def source_hash() -> str:
    """sha256 (16 hex digits) of every source the library is built from: stored with profile summaries
    (profiles/traffic_summary.py) so that bench.py can tell a measurement of THESE kernels from a stale one."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libepiekf.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = [hipcc(), *HIPCC_FLAGS, SRC, "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB
