"""Phase breakdown of the coverage pile-up windows of the 500 kb scan BAMs (build/scan500k, tools/make_scan_bams.py)."""
import os as _os; _os.environ.setdefault("LANCET_PHASE_TIMES", "1")      # (the engine accounts per-phase ticks only on request)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import abi, engine, host, workload
D = os.path.join(ROOT, "build", "scan500k")
H = host.NativeHost(os.path.join(D, "tumor.bam"), os.path.join(D, "normal.bam"), os.path.join(D, "ref.fa"))
o = host.default_opts(active_region=0)
hd = H.tile("chr22:186000-189500", o)
b, idx = H.batch(0, len(hd), o)
nr = np.diff(b.read_begin.astype(np.int64))
print("windows", b.n_windows, "reads per window", nr.tolist())
eng = engine.Engine(abi.default_params())
eng.upload(b)
for it in range(2):
    t = time.time(); eng.run(); dt = time.time() - t
    print("run", it, "wall", round(dt, 3), "kernel ms", eng.kernel_times(), "rerun", eng.rerun_count())
v, st = eng.results()
ph = eng.phase_times()
names = ["other", "ref repeat scan", "insert+verify", "node ids/hash", "pass2+csr+mate replay", "per-node minqv/lowcov", "materialize survivors", "order replay", "first lowcov+cc", "per-comp passes", "repeats in paths", "bfs/eka", "align fill", "align traceback", "transcript walk", "first compress"]
big = np.argsort(-nr)[:3]
for w in big:
    print("window", b.hdr[w], "reads", int(nr[w]), "builds", st[w]["n_builds"], "k", st[w]["final_k"], "nodes", st[w]["max_nodes"], "total ms", round(1000 * float(ph[w].sum()), 1))
    print("   ", [(names[i], round(1000 * float(ph[w][i]), 1)) for i in range(16) if ph[w][i] > 0.002])
