import sys, os
sys.path.insert(0, "/root/repo")
from lancet_amd import abi, engine, workload
n = int(sys.argv[1])
big = workload.make_scan_batch(n, 60, 60, seed=22)
eng = engine.Engine(abi.default_params())
eng.upload(big); eng.run(); eng.run()
v, st = eng.results()
print("n", n, "env", {k: v_ for k, v_ in os.environ.items() if k.startswith("LANCET_")}, "kernel ms", eng.kernel_times(), "prebuilt", eng.prebuilt_count(), "ahead", eng.ahead_counts(), "variants", len(v), "bad", sum(1 for s in st if s["status"] < 0))
