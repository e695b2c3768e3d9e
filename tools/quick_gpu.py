"""Quick GPU sanity/timing: replicate a golden case N times into one batch and time the engine."""
import os as _os; _os.environ.setdefault("LANCET_PHASE_TIMES", "1")      # (the engine accounts per-phase ticks only on request)
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
from lancet_amd import abi, engine, frontend

def replicate(batch, times):
    import copy
    b = batch
    R = b.n_reads
    def tile(a): return np.concatenate([a] * times)
    ref_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(b.ref_off.astype(np.int64)), times))]).astype(np.uint32)
    read_begin = np.concatenate([[0], np.cumsum(np.tile(np.diff(b.read_begin.astype(np.int64)), times))]).astype(np.uint32)
    seq_off = np.concatenate([[0], np.cumsum(np.tile(np.diff(b.seq_off.astype(np.int64)), times))]).astype(np.uint32)
    return frontend.WindowBatch(n_windows=b.n_windows * times, hdr=b.hdr * times, chrom=b.chrom * times, chr_id=tile(b.chr_id),
        ref_start=tile(b.ref_start), ref_off=ref_off, ref_bases=tile(b.ref_bases), read_begin=read_begin, seq_off=seq_off,
        seq=tile(b.seq), qual=tile(b.qual), label=tile(b.label), strand=tile(b.strand), mate=tile(b.mate), mapped=tile(b.mapped),
        name_rank=tile(b.name_rank))

case = sys.argv[1] if len(sys.argv) > 1 else "tile30"
times = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if case in ("bench", "bench60", "bench4", "bench5"):   # the bench.py workload (bench60: at 60x/60x; bench4: config 4; bench5: linked reads), `times` windows
    from lancet_amd import workload
    cov = 60 if case == "bench60" else 30
    if case == "bench4": big, mk, xk = workload.make_scan_batch(times, 100.0, 40.0, seed=22, str_fraction=0.30, lowcomplex_fraction=0.05), 11, 101
    elif case == "bench5": big, mk, xk = workload.make_scan_batch(times, 30.0, 30.0, seed=22, linked=True), 11, 101
    else: big, mk, xk = workload.make_scan_batch(times, cov, cov, seed=22), 11, 101
else:
    meta, batch, kept, (mk, xk) = gu.case_batch(case)
    big = replicate(batch, times)
p = abi.default_params(min_k=mk, max_k=xk, lr_mode=1) if case == "bench5" else abi.default_params(min_k=mk, max_k=xk)
eng = engine.Engine(p)
t = time.time(); eng.upload(big); print("upload s", time.time() - t, "windows", big.n_windows, "slots/bytes", eng.geometry())
for it in range(3):
    t = time.time(); eng.run(); dt = time.time() - t
    v, st = eng.results()
    nk = sum(s["n_kmers"] for s in st)
    print(f"run {it}: wall {dt:.3f}s kernel {eng.timing_ms()[1]:.1f} ms  windows/s {big.n_windows / dt:.0f}  Mkmers/s {nk / dt / 1e6:.1f} variants {len(v)} bad {sum(1 for s in st if s['status'] < 0)}")

names = ["other", "ref repeat scan", "build: k-mer insert+verify", "build: node ids/hash", "build: pass2+csr+mate replay", "build: per-node minqv/lowcov pred", "materialize survivors", "libstdc++ order replay", "first lowcov+cc", "per-comp graph passes", "repeats in paths", "bfs+path string+hamming", "align fill", "align traceback", "transcript walk", "first compress"]
pt = eng.phase_times().sum(axis=0)
tot = pt.sum()
for i, n in enumerate(names):
    print(f"  phase {i:2d} {n:36s} {pt[i]:9.3f} s  {100 * pt[i] / tot:5.1f} %")
print("  total slot-seconds", tot, "per window ms", 1000 * tot / big.n_windows)

bn = ["setup", "reads -> LDS", "repeat scan + k", "occurrence space + pairs", "insert", "node ids", "slot -> id + counts", "std::hash", "class counts + mate replay", "candidates", "per-position counts", "survivors", "ref coverage", "trace masks", "edges + records", "tail"]
bp = eng.build_phase_times(); bt = sum(bp)
print("build service (posted, served, not buildable, taken back):", eng.svc_counts(), "graphs built ahead / taken from the pool:", eng.ahead_counts())
print("LDS build kernel: prebuilt", eng.prebuilt_count(), "of", big.n_windows, "kernel ms", eng.kernel_times())
for i, n in enumerate(bn):
    print(f"  build phase {i:2d} {n:32s} {bp[i]:9.3f} s  {100 * bp[i] / max(bt, 1e-12):5.1f} %")
print("  total workgroup-seconds", bt, "per window us", 1e6 * bt / max(1, big.n_windows))
ptw = eng.phase_times().sum(axis=1)
import numpy as _np
order_ = _np.argsort(-ptw)
print("slowest windows (ms, builds, final_k):", [(round(1000 * float(ptw[i]), 2), st[i]["n_builds"], st[i]["final_k"]) for i in order_[:12]])
print("window time percentiles ms: p50 %.2f p90 %.2f p99 %.2f max %.2f ; windows with > 1 build: %d" % (1000 * _np.percentile(ptw, 50), 1000 * _np.percentile(ptw, 90), 1000 * _np.percentile(ptw, 99), 1000 * ptw.max(), sum(1 for s_ in st if s_["n_builds"] > 1)))
if os.environ.get("QUICK_SLOW"):
    ns = int(os.environ["QUICK_SLOW"])
    ph = eng.phase_times()
    sl = ph[order_[:ns]].sum(axis=0)
    print(f"phase split of the {ns} slowest windows (total {sl.sum():.3f} s):")
    for i, n in enumerate(names):
        print(f"  slow phase {i:2d} {n:36s} {sl[i]:9.3f} s  {100 * sl[i] / sl.sum():5.1f} %")
    multi = [i for i, s_ in enumerate(st) if s_["n_builds"] > 1]
    ml = ph[multi].sum(axis=0)
    print(f"phase split of the {len(multi)} windows with > 1 build (total {ml.sum():.3f} s, {100 * ml.sum() / tot:.1f} % of all slot time):")
    for i, n in enumerate(names):
        print(f"  multi phase {i:2d} {n:36s} {ml[i]:9.3f} s  {100 * ml[i] / ml.sum():5.1f} %")
    import collections
    print("builds histogram:", sorted(collections.Counter(s_["n_builds"] for s_ in st).items()))
if os.environ.get("QUICK_TIMELINE"):      # (engine built with -DLANCET_PROF_TIMELINE: tools/variant.sh) the window kernel's launch over time
    ph = eng.phase_times()
    start, end, susp, res, nres, slot = ph[:, 2], ph[:, 3], ph[:, 4], ph[:, 5], ph[:, 6] * 1e8, (ph[:, 7] * 1e8).astype(int)
    t0 = start.min(); span = end.max() - t0
    ms = lambda x: round(1000 * float(x - t0), 2)
    print(f"timeline: span {1000 * span:.2f} ms; the last window was taken at {ms(start.max())} ms; ends: p50 {ms(_np.percentile(end, 50))} p90 {ms(_np.percentile(end, 90))} p99 {ms(_np.percentile(end, 99))} p99.9 {ms(_np.percentile(end, 99.9))}")
    parked = (nres > 0)
    print(f"  put aside for the build service: {int(parked.sum())} windows, {int(nres.sum())} times; last hand-over {ms(susp[parked].max()) if parked.any() else 0} ms, last resume {ms(res[parked].max()) if parked.any() else 0} ms; mean wait of the last round {1000 * float((res[parked] - susp[parked]).mean()) if parked.any() else 0:.2f} ms")
    if parked.any():                            # the waits (last round of each window) by the millisecond in which the window was put aside
        w_ = res[parked] - susp[parked]; at = susp[parked] - t0
        for b in range(int(1000 * span) + 1):
            m = (at >= b * 1e-3) & (at < (b + 1) * 1e-3)
            if m.any(): print(f"    put aside in ms {b:2d}: {int(m.sum()):4d} windows, wait mean {1000 * float(w_[m].mean()):.2f} max {1000 * float(w_[m].max()):.2f} ms")
    nslot = int(slot.max()) + 1
    last = _np.zeros(nslot); busy = _np.zeros(nslot)
    _np.maximum.at(last, slot, end - t0)
    print(f"  slots seen {len(set(slot.tolist()))}; a slot's last window ends at: mean {1000 * last[last > 0].mean():.2f} ms p10 {1000 * _np.percentile(last[last > 0], 10):.2f} p50 {1000 * _np.percentile(last[last > 0], 50):.2f} p90 {1000 * _np.percentile(last[last > 0], 90):.2f}")
    edges = _np.arange(0, span + 5e-4, 5e-4)
    act = [int(((start - t0 < b + 5e-4) & (end - t0 > b)).sum()) for b in edges]
    print("  windows between taken and finished, per 0.5 ms:", act)
    fin = [int(((end - t0 >= b) & (end - t0 < b + 5e-4)).sum()) for b in edges]
    print("  windows finished per 0.5 ms:", fin)
    tk = [int(((start - t0 >= b) & (start - t0 < b + 5e-4)).sum()) for b in edges]
    print("  windows taken per 0.5 ms:", tk)
    lastw = _np.argsort(-end)[:20]
    print("  the last to finish (taken, put aside, resumed, finished ms; times resumed; builds; k; busy ms):")
    for i in lastw: print("   ", ms(start[i]), ms(susp[i]) if nres[i] else "-", ms(res[i]) if nres[i] else "-", ms(end[i]), int(nres[i]), st[i]["n_builds"], st[i]["final_k"], round(1000 * float(ptw[i] - ph[i, 2:8].sum()), 2))
