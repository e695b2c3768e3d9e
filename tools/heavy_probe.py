import sys, os, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from lancet_amd import abi, engine, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
big = workload.make_scan_batch(n, 30, 30, seed=22)
eng = engine.Engine(abi.default_params(min_k=11, max_k=101))
eng.upload(big); eng.run()
v, st = eng.results()
out = (C.c_uint32 * (8 * n))()
L = engine.lib()
L.lancet_debug_pre_headers.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
rc = L.lancet_debug_pre_headers(eng.h if hasattr(eng, "h") else eng._h, out)
h = np.ctypeslib.as_array(out).reshape(n, 8)
nb = np.array([s["n_builds"] for s in st]); fk = np.array([s["final_k"] for s in st])
heavy = h[:, 2] != 0; multi = nb > 1
print("rc", rc, "windows", n, "heavy", heavy.sum(), "multi", multi.sum(), "heavy&multi", (heavy & multi).sum())
print("K (first) histogram", np.unique(h[:, 1], return_counts=True))
for b in range(1, 7):
    m = nb == b
    print("builds", b, "count", m.sum(), "heavy", (heavy & m).sum())

# per-window slot time of the window kernel against what the build kernel knows (scheduling order)
eng2 = engine.Engine(abi.default_params(min_k=11, max_k=101))
eng2.upload(big); eng2.run(); eng2.run()
t = eng2.phase_times().sum(axis=1) * 1000
for name, col in (("N", 3), ("nsurv", 4), ("numcomp", 5), ("ncand", 6)):
    x = h[:, col].astype(float)
    print("corr(time,", name, ") =", round(float(np.corrcoef(t, x)[0, 1]), 3))
nxt = h[:, 7] != 0
print("mean ms: all", round(float(t.mean()), 2), "heavy", round(float(t[heavy].mean()), 2), "chained", round(float(t[nxt].mean()), 2), "K==11", round(float(t[h[:, 1] == 11].mean()), 2))
order = np.argsort(-t)[:200]
print("of the 200 slowest: heavy", int(heavy[order].sum()), "K==11", int((h[order, 1] == 11).sum()), "median nsurv", float(np.median(h[order, 4])), "vs all", float(np.median(h[:, 4])), "median numcomp", float(np.median(h[order, 5])), "vs", float(np.median(h[:, 5])))
# how good would a linear proxy be: LPT by predicted cost
import itertools
for w_surv, w_heavy in ((1, 0), (1, 500), (1, 2000), (0, 1)):
    pred = w_surv * h[:, 4].astype(float) + w_heavy * heavy.astype(float)
    o = np.argsort(-pred, kind="stable")
    # list scheduling of the measured times on 5120 slots in that order
    import heapq
    slots = [0.0] * min(5120, n)
    heapq.heapify(slots)
    for i in o:
        s0 = heapq.heappop(slots); heapq.heappush(slots, s0 + float(t[i]))
    print("order by %d*nsurv + %d*heavy: makespan %.2f ms (sum/slots %.2f)" % (w_surv, w_heavy, max(slots), t.sum() / len(slots)))
v2, st2 = eng2.results()
nb2 = np.array([s["n_builds"] for s in st2])
ph = eng2.phase_times()
import collections
print("200 slowest: builds", sorted(collections.Counter(nb2[order].tolist()).items()), "chained", int(nxt[order].sum()))
gen = ph[:, 2:9].sum(axis=1) * 1000
print("200 slowest: mean total ms %.1f, of which general-build phases %.1f" % (t[order].mean(), gen[order].mean()))
one = order[nb2[order] == 1][:100]
print("slowest single-build windows: n", len(one), "mean ms", round(float(t[one].mean()), 1) if len(one) else 0, "phase split (ms):", [round(float(x), 2) for x in (ph[one].mean(axis=0) * 1000)] if len(one) else [])
