"""Tuning/debug aid: the bench workload with and without graphs built ahead (LANCET_AHEAD_DEPTH): results must be identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet_amd import abi, engine, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
big = workload.make_scan_batch(n, 30, 30, seed=22)
out = {}
for depth in ("0", "3", "3"):
    os.environ["LANCET_AHEAD_DEPTH"] = depth
    eng = engine.Engine(abi.default_params(min_k=11, max_k=101))
    eng.upload(big); eng.run()
    v, st = eng.results()
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    cur = (v, [key(s) for s in st])
    print("depth", depth, "variants", len(v), "ahead", eng.ahead_counts(), "kernel ms", eng.kernel_times())
    if "0" in out:
        ref = out["0"]
        badw = [w for w in range(n) if ref[1][w] != cur[1][w]]
        print("  stats differ in", len(badw), "windows", badw[:10], [(ref[1][w], cur[1][w]) for w in badw[:3]])
        print("  variants equal:", ref[0] == cur[0])
        if ref[0] != cur[0]:
            a = set(map(str, ref[0])); b = set(map(str, cur[0]))
            print("  only without:", list(a - b)[:3]); print("  only with:", list(b - a)[:3])
    out.setdefault(depth, cur)
    eng.close()
