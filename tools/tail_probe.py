"""Per-window slot time of the window kernel against what the build kernel knows about the window (tuning aid for order_kernel)."""
import os as _os; _os.environ.setdefault("LANCET_PHASE_TIMES", "1")      # (the engine accounts per-phase ticks only on request)
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import abi, engine, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
big = workload.make_scan_batch(n, cov, cov, seed=22)
eng = engine.Engine(abi.default_params(min_k=11, max_k=101))
eng.upload(big)
for _ in range(2): eng.run()
v, st = eng.results()
ph = eng.phase_times()
hd = np.zeros(8 * n, dtype=np.uint32)
eng.L.lancet_debug_pre_headers.argtypes = [C.c_void_p, C.c_void_p]
eng.L.lancet_debug_pre_headers(eng.h, hd.ctypes.data)
hd = hd.reshape(n, 8)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "tail_probe.npz"), phase=ph, hdr=hd, builds=np.array([s["n_builds"] for s in st]), nvar=np.array([s["n_variants"] for s in st]),
                    reads=np.diff(big.read_begin), kernel=np.array(eng.kernel_times()))
print("kernel ms", eng.kernel_times(), "svc", eng.svc_counts())
import collections
why = collections.Counter(int(x) >> 8 for x in hd[:, 0] if (int(x) & 0xFF) != 1)
print("windows not built in LDS by reason (BLW_*: 2 size, 5 table, 6 nodes, 7 tracked, 8 cand, 9 qv, 10 surv, 11 mate, 12 names, 13 pairs):", dict(why))
t = ph.sum(axis=1) * 1000
nb = (hd[:, 0] & 0xFF) != 1
print("time ms: built mean %.2f max %.2f ; not built mean %.2f max %.2f ; reads not built mean %.0f" % (t[~nb].mean(), t[~nb].max(), t[nb].mean() if nb.any() else 0, t[nb].max() if nb.any() else 0, np.diff(big.read_begin)[nb].mean() if nb.any() else 0))
order = np.argsort(-t)[:16]
builds = np.array([s["n_builds"] for s in st])
print("slowest windows (ms, builds, built in LDS, reads):", [(round(float(t[w]), 1), int(builds[w]), bool(~nb[w]), int(np.diff(big.read_begin)[w])) for w in order])
for q in (50, 90, 99, 99.9): print("slot time percentile %.1f: %.2f ms" % (q, np.percentile(t, q)))
print("slot time by number of builds:", {int(k): (int((builds == k).sum()), round(float(t[builds == k].mean()), 2)) for k in np.unique(builds)})
print("windows in the re-run tier", eng.rerun_count(), "ahead", eng.ahead_counts())
