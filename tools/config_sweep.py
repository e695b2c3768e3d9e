"""BASELINE.md configs 2 and 4 on one GPU: windows/s, k-attempt histogram, tier-2 (work-space overflow) re-runs.

    python tools/config_sweep.py [n_windows]

config 2: chr22-scan proxy at 30x/30x and 60x/60x; config 4: 100x tumor / 40x normal over STR-rich sequence
(30 % STR blocks, 5 % two-letter low complexity).  Same engine call as bench.py; inputs resident before timing."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lancet_amd import abi, engine, workload

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
CASES = [("config2 30x/30x", dict(cov_t=30, cov_n=30)), ("config2 60x/60x", dict(cov_t=60, cov_n=60)),
         ("config4 100x/40x STR-rich", dict(cov_t=100, cov_n=40, str_fraction=0.30, lowcomplex_fraction=0.05))]
out = []
for name, kw in CASES:
    batch = workload.make_scan_batch(nw, seed=22, **kw)
    eng = engine.Engine(abi.default_params())
    eng.upload(batch)
    eng.run()                                               # warm-up
    t = time.perf_counter(); eng.run(); dt = time.perf_counter() - t
    v, st = eng.results()
    fk = np.array([s["final_k"] for s in st]); nb = np.array([s["n_builds"] for s in st]); status = np.array([s["status"] for s in st])
    hist = {int(k): int((fk == k).sum()) for k in np.unique(fk)}
    rec = {"case": name, "windows": nw, "reads_per_window": round(batch.n_reads / nw, 1), "windows_per_s": round(nw / dt, 1),
           "kernel_ms": round(eng.timing_ms()[1], 2), "Mkmers_per_s": round(sum(s["n_kmers"] for s in st) / dt / 1e6, 1),
           "variants": len(v), "builds_per_window": round(float(nb.mean()), 3), "max_builds": int(nb.max()),
           "tier2_reruns": int(eng.rerun_count()), "failed_windows": int((status < 0).sum()),
           "k_exhausted": int((status == 2).sum()),
           "max_nodes_p50_p99_max": [int(x) for x in np.percentile([s["max_nodes"] for s in st], [50, 99, 100])],
           "final_k_hist": hist}
    print(json.dumps(rec)); sys.stdout.flush()
    out.append(rec)
    eng.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "config_sweep.json"), "w"), indent=1)
