"""A coverage pile-up inside an ordinary batch (what tests/test_engine_gpu.py::test_coverage_pile_up... builds), run a few times: under
`rocprofv3 --kernel-trace --stats` this shows the several-wave kernel of the re-run tier (window_kernel_fat) next to the others."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lancet_amd import abi, engine, workload
import test_engine_gpu as T
plain = workload.make_scan_batch(2048, 30, 30, seed=5)
pile = workload.make_scan_batch(4, 1500, 1500, seed=6)
both = T._concat(plain, pile)
eng = engine.Engine(abi.default_params())
eng.upload(both)
for it in range(3):
    t = time.time(); eng.run(); dt = time.time() - t
    v, st = eng.results()
    print("run", it, "wall %.3f s" % dt, "kernel ms", eng.kernel_times(), "windows in the re-run tier", eng.rerun_count(), "reads of the pile-ups", np.diff(both.read_begin)[-4:].tolist(), "bad", sum(1 for s in st if s["status"] < 0))
