"""Engine statistics of an end-to-end region batch built by the native host side (debugging aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import host, engine, abi
D = sys.argv[1] if len(sys.argv) > 1 else "build/scan500k"
o = host.default_opts(active_region=0)
H = host.NativeHost(f"{D}/tumor.bam", f"{D}/normal.bam", f"{D}/ref.fa")
hdrs = H.tile(sys.argv[2] if len(sys.argv) > 2 else "chr22:1000-499000", o)
b, idx = H.batch(0, len(hdrs), o)
eng = engine.Engine(abi.default_params())
for it in range(3):
    t = time.time(); v, st = eng.process(b); dt = time.time() - t
    print(f"run {it}: process {dt:.3f} s, kernels {eng.timing_ms()} ms, reruns {eng.rerun_count()}, variants {len(v)}")
nb = np.array([s["n_builds"] for s in st]); fk = np.array([s["final_k"] for s in st]); mn = np.array([s["max_nodes"] for s in st])
print("reads/window", b.n_reads / b.n_windows, "builds mean/max", nb.mean(), nb.max(), "final_k hist", {int(k): int((fk == k).sum()) for k in np.unique(fk)})
print("max_nodes p50/p99/max", np.percentile(mn, [50, 99, 100]), "status", {int(k): int((np.array([s['status'] for s in st]) == k).sum()) for k in (-1, 0, 1, 2)})
pt = eng.phase_times().sum(axis=0); print("phase shares %", np.round(100 * pt / pt.sum(), 1))
lens = np.diff(b.seq_off.astype(np.int64)); print("read len min/max", lens.min(), lens.max(), "names/window", len(set(b.name_rank[b.read_begin[0]:b.read_begin[1]])), "reads in w0", b.read_begin[1])
