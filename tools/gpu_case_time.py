"""Times one golden case on the GPU through the C-ABI and checks records / stats / trace / VCF against the oracle and the reference golden."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from lancet_amd import engine
from oracle import oracle

for case in sys.argv[1:]:
    meta, batch, kept, (mk, xk) = gu.case_batch(case)
    p = gu.params(meta)
    eng = engine.Engine(p, device=0, trace_words=1 << 17)
    t = time.time(); variants, stats = eng.process(batch); dt = time.time() - t
    ov, ostats, _ = oracle.run(batch, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    ok = variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
    tr = gu.digest_trace(eng.trace_text()) == gu.golden_trace(case)
    print(f"{case}: windows {batch.n_windows} wall {dt:.2f}s kernel {eng.timing_ms()[1]:.1f} ms rerun {eng.rerun_count()} records_ok {ok} trace_ok {tr} bad {[s['status'] for s in stats if s['status'] < 0]}", flush=True)
    eng.close()
