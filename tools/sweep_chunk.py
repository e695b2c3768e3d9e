"""Windows [a, b) of one draw of tests/test_engine_gpu_sweep.py on the device (debugging aid). usage: sweep_chunk.py <seed> <a> <b>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lancet_amd import abi, engine, workload
import test_engine_gpu_sweep as sw
seed, a, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
over, wl = sw.draw(seed)
p = abi.default_params(**over)
batch = workload.sub_batch(workload.make_scan_batch(384, seed=700 + seed, **wl), a, b)
os.environ.setdefault("LANCET_PRE_WIDE", "1")
eng = engine.Engine(p, device=0)
variants, stats = eng.process(batch)
print("chunk", a, b, "ok: records", len(variants), "in LDS", eng.prebuilt_count(), "final k", [s["final_k"] for s in stats], flush=True)
