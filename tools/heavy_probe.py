import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from lancet_amd import abi, engine, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
big = workload.make_scan_batch(n, 30, 30, seed=22)
eng = engine.Engine(abi.default_params(min_k=11, max_k=101))
eng.upload(big); eng.run()
v, st = eng.results()
out = (C.c_uint32 * (4 * n))()
L = engine.lib()
L.lancet_debug_pre_headers.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
rc = L.lancet_debug_pre_headers(eng.h if hasattr(eng, "h") else eng._h, out)
h = np.ctypeslib.as_array(out).reshape(n, 4)
nb = np.array([s["n_builds"] for s in st]); fk = np.array([s["final_k"] for s in st])
heavy = h[:, 2] != 0; multi = nb > 1
print("rc", rc, "windows", n, "heavy", heavy.sum(), "multi", multi.sum(), "heavy&multi", (heavy & multi).sum())
print("K (first) histogram", np.unique(h[:, 1], return_counts=True))
for b in range(1, 7):
    m = nb == b
    print("builds", b, "count", m.sum(), "heavy", (heavy & m).sum())
