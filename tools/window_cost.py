"""Tuning aid (GPU box): what a window costs the window kernel against what the build kernel knows about it when the processing
order is made (order_class, engine.hip).  Writes gpurun_out/window_cost_<case>.npz: per window the busy time, the phase split, the
statistics and the hand-off header's numbers.
    python tools/window_cost.py [bench|bench60|bench4] [windows]"""
import os as _os; _os.environ.setdefault("LANCET_PHASE_TIMES", "1")      # (the engine accounts per-phase ticks only on request)
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import abi, engine, workload

case = sys.argv[1] if len(sys.argv) > 1 else "bench"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
if case == "bench4": big = workload.make_scan_batch(nw, 100.0, 40.0, seed=22, str_fraction=0.30, lowcomplex_fraction=0.05)
else: big = workload.make_scan_batch(nw, 60 if case == "bench60" else 30, 60 if case == "bench60" else 30, seed=22)
eng = engine.Engine(abi.default_params(min_k=11, max_k=101))
eng.upload(big)
for _ in range(2): eng.run()
v, st = eng.results()
ph = eng.phase_times()
hd = np.zeros(8 * nw, dtype=np.uint32); cm = np.zeros(4 * nw, dtype=np.uint32)
eng.L.lancet_debug_pre_headers.argtypes = [C.c_void_p, C.c_void_p]; eng.L.lancet_debug_pre_cmp.argtypes = [C.c_void_p, C.c_void_p]
assert eng.L.lancet_debug_pre_headers(eng.h, hd.ctypes.data) == 0 and eng.L.lancet_debug_pre_cmp(eng.h, cm.ctypes.data) == 0
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/window_cost_{case}.npz", phase=ph, hdr=hd.reshape(-1, 8), cmp=cm.reshape(-1, 4),
                    builds=np.array([s["n_builds"] for s in st]), final_k=np.array([s["final_k"] for s in st]), nvar=np.array([s["n_variants"] for s in st]),
                    max_nodes=np.array([s["max_nodes"] for s in st]), kernel_ms=np.array(eng.kernel_times()))
print(case, nw, "kernel ms", eng.kernel_times(), "busy s", ph.sum())
