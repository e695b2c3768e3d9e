"""One draw of tests/test_engine_gpu_sweep.py on the device, alone (a debugging aid: LANCET_DEBUG=1 names the kernel that was running).
usage: sweep_one.py <seed> [windows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lancet_amd import abi, engine, workload
import test_engine_gpu_sweep as sw
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 384
over, wl = sw.draw(seed)
print(over, wl, flush=True)
p = abi.default_params(**over)
batch = workload.make_scan_batch(n, seed=700 + seed, **wl)
eng = engine.Engine(p, device=0)
variants, stats = eng.process(batch)
import ctypes as C
out = (C.c_uint32 * (8 * batch.n_windows))()
eng.L.lancet_debug_pre_headers(eng.h, out)
for w in range(batch.n_windows):
    if (out[8 * w] >> 8) == 98:
        print("DIAG window", w, "slot", out[8 * w + 1], "e", hex(out[8 * w + 3]), "nslots", out[8 * w + 2], "N", out[8 * w + 4], "K", out[8 * w + 6], flush=True)
print("ran: records", len(variants), "in LDS", eng.prebuilt_count(), "re-run", eng.rerun_count(), "svc", eng.svc_counts(), flush=True)
ov, ostats = sw.oracle_parallel(batch, p)
key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
print("equal:", variants == ov, [key(s) for s in stats] == [key(s) for s in ostats], flush=True)
eng.close()
