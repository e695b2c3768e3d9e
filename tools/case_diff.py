"""Debug aid: run one golden case on the GPU and print the records that differ from the oracle's."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from lancet_amd import engine
from oracle import oracle
case = sys.argv[1]
meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
p = gu.params(meta)
eng = engine.Engine(p, device=0, trace_words=1 << 17)
variants, stats = eng.process(batch)
ov, ostats, _ = oracle.run(batch, p)
print("n", len(variants), len(ov), "rerun", eng.rerun_count())
for a, b in zip(variants, ov):
    if a != b:
        print("DIFF window", a["window"], "seq", a["seq"])
        for k in a:
            if a[k] != b.get(k): print("   ", k, a[k], "!=", b.get(k))
