"""The two side configurations of bench.py at a glance: wall time per pass, kernel times, how many windows went where -- and the same
with a 64-entry tier-1 node table (LANCET_NODE_CAP1=64: practically every window through the several-wave kernel of the re-run tier),
to see what that kernel makes of ordinary heavy windows.  usage: side_probe.py [windows60 [windows4]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import abi, engine, workload
n60 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n4 = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cases = [("60x/60x", n60, 60.0, 60.0, {}), ("config 4", n4, 100.0, 40.0, dict(str_fraction=0.3, lowcomplex_fraction=0.05))]
for name, n, ct, cn, kw in cases:
    if n <= 0:
        continue
    b = workload.make_scan_batch(n, ct, cn, seed=22, **kw)
    for cap in (None, "64"):
        if cap: os.environ["LANCET_NODE_CAP1"] = cap
        else: os.environ.pop("LANCET_NODE_CAP1", None)
        eng = engine.Engine(abi.default_params(), device=0)
        eng.upload(b); eng.run()
        t = time.perf_counter(); eng.run(); eng.run(); dt = (time.perf_counter() - t) / 2
        v, st = eng.results()
        print(name, "node cap", cap, "windows", n, "wall ms %.1f" % (dt * 1e3), "kernel ms", [round(x, 2) for x in eng.kernel_times()], "timing", [round(x, 2) for x in eng.timing_ms()],
              "in LDS", eng.prebuilt_count(), "re-run tier", eng.rerun_count(), "records", len(v), "bad", sum(1 for s in st if s["status"] < 0), flush=True)
        eng.close()
