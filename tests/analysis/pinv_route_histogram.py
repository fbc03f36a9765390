"""Which routes the smoother's pinv takes on the headline workload (test-side analysis: it runs the C oracle).
    python tests/analysis/pinv_route_histogram.py  -> route (0 factorisation + one-sided Jacobi, 2 certified full rank), rank kept, sweeps"""
import sys, ctypes as C, collections
import numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
from epidemicmodeling_amd import synth
w = synth.make_cfg4()
idx = np.linspace(0, w.B - 1, 40).astype(np.int64)
ws = w.select(idx)
ref = H.oracle_batch(ws)
Pm = ref["P_MINUS"]            # [T, 36, B]
lib = C.CDLL(os.path.join(ROOT, 'oracle', 'libekf_oracle.so'))
f = lib.orc_sym_pinv_ex
f.restype = C.c_int
hist = collections.Counter(); byday = collections.defaultdict(collections.Counter)
T = Pm.shape[0]
for c in range(Pm.shape[2]):
    for t in range(1, T):
        A = np.ascontiguousarray(Pm[t, :, c].reshape(6, 6).T)   # column-major rows e = i + 6 j
        X = np.zeros(36); route = C.c_int(0); sw = C.c_int(0)
        rk = f(6, A.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p), C.byref(route), C.byref(sw))
        hist[(route.value, rk, sw.value)] += 1
        byday[t // 65][(route.value, rk)] += 1
tot = sum(hist.values())
for k, v in sorted(hist.items(), key=lambda x: -x[1])[:20]:
    print("route %d rank %d sweeps %d : %5.1f %%" % (k[0], k[1], k[2], 100.0 * v / tot))
for seg in sorted(byday):
    print("days %3d-%3d" % (seg * 65, seg * 65 + 64), {k: round(100.0 * v / sum(byday[seg].values())) for k, v in sorted(byday[seg].items())})
