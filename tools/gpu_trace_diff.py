"""Debug helper: run a golden case on the GPU with tracing and print the first differences against the
reference trace digest."""
import difflib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from lancet_amd import abi, engine
case = sys.argv[1] if len(sys.argv) > 1 else "cfg1_k25"
meta, batch, kept, (mk, xk) = gu.case_batch(case)
eng = engine.Engine(abi.default_params(min_k=mk, max_k=xk), trace_words=1 << 17)
for it in range(2):
    v, st = eng.process(batch)
    d = gu.digest_trace(eng.trace_text()); g = gu.golden_trace(case)
    print("run", it, "variants", len(v), "status", [s["status"] for s in st][:10], "trace", "OK" if d == g else "DIFF")
    if d != g:
        for l in list(difflib.unified_diff(g.splitlines(), d.splitlines(), lineterm="", n=1))[:40]:
            print("   ", l[:200])
