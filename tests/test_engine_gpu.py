"""Parity tests proper: the HIP engine, called through the C-ABI, against (a) the oracle on the same inputs,
(b) the committed reference goldens (VCF + `-v` stage trace).  Bit-exact: everything is integer/index work
except four float coverages per node, which must round exactly like the reference (no tolerance)."""
import os

import numpy as np
import pytest

import golden_util as gu
from lancet_amd import abi, engine
from oracle import oracle

pytestmark = pytest.mark.gpu


def _names(batch):
    id2chr = {}
    for c, i in zip(batch.chrom, batch.chr_id):
        id2chr[int(i)] = c
    return [id2chr[i] for i in range(len(id2chr))]


# tools/fat_check.sh runs this suite with a 64-entry tier-1 node table (LANCET_NODE_CAP1=64: practically every window is assembled by the
# several-wave kernel of the re-run tier): the records are compared as always, the counts of what tier 1 itself served are not
TIER1 = "LANCET_NODE_CAP1" not in os.environ

# every reference-made golden, including `bushy` (--low-cov 0: one path search of 600 k partial paths, 12 s for its 17 windows on
# an MI355X, all through the worst-case tier) and `evenk` (--min-k 12: k-mers that are their own reverse complement)

GPU_CASES = list(gu.CASES)


@pytest.mark.parametrize("case", GPU_CASES)
def test_engine_matches_oracle_and_reference(case):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    lr = gu.case_lr(meta)                                                     # --linked-reads goldens
    p = gu.params(meta)
    eng = engine.Engine(p, device=0, trace_words=1 << 17)
    variants, stats = eng.process(batch)
    ov, ostats, _ = oracle.run(batch, p)
    assert all(s["status"] >= 0 for s in stats), [s for s in stats if s["status"] < 0][:3]
    assert variants == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in stats] == [key(s) for s in ostats]
    assert gu.digest_trace(eng.trace_text()) == gu.golden_trace(case)          # every stage == reference `-v`
    db = engine.VariantDB()
    vp, n, blob, _ = eng.raw_results()
    if lr:
        lp, bp, _ = eng.raw_results_lr()
        db.add_raw_lr(vp, lp, n, blob + b"\0", bp, batch.bx_names, _names(batch))
    else:
        db.add_raw(vp, n, blob + b"\0", _names(batch))
    assert db.vcf() == gu.golden_vcf(case)                                      # byte-identical VCF
    eng.close()


def test_graphs_built_ahead_after_the_reference_table_was_trimmed():
    """tests/golden/ahead_trim.npz: windows of the bench workload where the window kernel takes a graph the LDS build kernel built
    ahead, after the rejected k had trimmed Ref_t::seq: records, stats and trace equal the oracle's."""
    p = abi.default_params()
    used = 0
    for batch in gu.load_batches_npz("ahead_trim.npz"):
        eng = engine.Engine(p, device=0, trace_words=1 << 17)
        variants, stats = eng.process(batch)
        ov, ostats, otr = oracle.run(batch, p, verbose=True)
        key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
        assert variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
        assert gu.digest_trace(eng.trace_text()) == gu.digest_trace(otr)
        used += eng.ahead_counts()[1]
        eng.close()
    assert used >= 4


def test_deep_windows_through_the_1024_lane_build_configuration():
    """60x/60x windows (~360 reads, ~58 k bases: above the 512-lane configuration's LDS limits) are taken off the list the first
    build kernel leaves and built by the 1024-lane one; records and stats equal the oracle's, also with that kernel switched
    off (LANCET_NO_LARGE_BUILD: general path)."""
    import os
    from lancet_amd import workload
    batch = workload.make_scan_batch(160, 60, 60, seed=22)
    p = abi.default_params()
    ov, ostats, _ = oracle.run(batch, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    for off in (False, True):
        if off:
            os.environ["LANCET_NO_LARGE_BUILD"] = "1"
        try:
            eng = engine.Engine(p, device=0)
            variants, stats = eng.process(batch)
            built = eng.prebuilt_count()
            eng.close()
        finally:
            os.environ.pop("LANCET_NO_LARGE_BUILD", None)
        assert variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
        assert (built == 0) if off else (built == 160), built        # (all: the 1024-lane configuration's replay list holds the 6417 occurrences of windows 94-96)


def test_deep_str_windows_build_in_lds_at_k_above_31():
    """BASELINE config 4 on the device: 100x / 40x windows over STR-rich sequence build at k = 33..101 with 8-12 k distinct k-mers -- in
    LDS, by the 1024-lane configuration (multi-word k-mers, 16 384-slot table, wide hand-off areas), the libstdc++ table order of those
    tables (bucket counts 10 273 and 20 753) included; equal to the oracle, twice."""
    from lancet_amd import workload
    batch = workload.make_scan_batch(192, 100.0, 40.0, seed=22, str_fraction=0.30, lowcomplex_fraction=0.05)
    p = abi.default_params()
    ov, ostats, _ = oracle.run(batch, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    builds = sum(1 for s in ostats if s["n_builds"] > 0)
    assert builds > 60 and max(s["final_k"] for s in ostats) > 95
    eng = engine.Engine(p, device=0)
    for _ in range(2):
        variants, stats = eng.process(batch)
        assert variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
        assert not TIER1 or (eng.prebuilt_count() >= (9 * builds) // 10 and eng.rerun_count() == 0), (eng.prebuilt_count(), builds, eng.rerun_count())
        if TIER1:     # the table order, cleanDead and the components of these 8-12 k-node tables came from the build kernel too (16-bit bucket minima)
            hd = [h for h in eng.pre_headers() if h["built"]]
            assert len(hd) >= (9 * builds) // 10 and all(h["order"] for h in hd) and max(h["nodes"] for h in hd) > 10273 and min(h["nodes"] for h in hd) > 4096
    eng.close()


def test_very_deep_windows_use_17_bit_offsets_in_lds():
    """100x/100x windows at a low error rate: ~640 reads, ~100 k bases with every read padded to 16 -- LDS offsets above 65 535 in the
    1024-lane build configuration (15-bit fingerprint over a 17-bit offset in the k-mer table); all built in LDS, equal to the oracle."""
    from lancet_amd import workload
    batch = workload.make_scan_batch(96, 100, 100, seed=5, error_rate=0.001)
    p = abi.default_params()
    ov, ostats, _ = oracle.run(batch, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    eng = engine.Engine(p, device=0)
    for _ in range(2):
        variants, stats = eng.process(batch)
        assert variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
        assert not TIER1 or (eng.prebuilt_count() == 96 and eng.rerun_count() == 0)      # (windows 80-83: a mate-overlap replay of 14 k occurrences, in two ranges of nodes)
    eng.close()


def test_engine_is_deterministic_and_order_independent():
    """Same windows in a different batch order / slot assignment give the same per-window records."""
    meta, batch, kept, (min_k, max_k) = gu.case_batch("tile30")
    p = abi.default_params(min_k=min_k, max_k=max_k)
    eng = engine.Engine(p)
    a, sa = eng.process(batch)
    b, sb = eng.process(batch)
    assert a == b and sa == sb
    eng.close()


def test_empty_and_readless_windows():
    from lancet_amd import frontend
    w = [frontend.Window("c:1-600", "c", 1, 601, "ACGT" * 150), frontend.Window("c:101-700", "c", 101, 701, "TTGCA" * 120)]
    batch = frontend.build_batch(w, [[], []])
    eng = engine.Engine()
    v, st = eng.process(batch)
    assert v == [] and [s["status"] for s in st] == [1, 1]       # LANCET_W_NO_READS (Microassembler.cc:83)
    ov, ost, _ = oracle.run(batch)
    assert [s["status"] for s in ost] == [1, 1]
    empty = frontend.build_batch([], [])
    v, st = eng.process(empty)
    assert v == [] and st == []
    eng.close()


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def test_device_alignment_equals_reference_align_cc():
    """Device Gotoh (anti-diagonal wavefront + traceback) vs the oracle, which test_oracle_golden pins against
    the reference's own align.cc (oracle/_ref)."""
    rng = np.random.default_rng(11)
    eng = engine.Engine()
    n_band = 0
    for it in range(40):
        n = int(rng.integers(30, 600))
        s = _rand_seq(rng, n)
        t = list(s)
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, max(1, len(t))))
            r = rng.random()
            if r < 0.34 and t:
                t[p] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.67:
                t[p:p] = list(_rand_seq(rng, int(rng.integers(1, 60))))
            elif t:
                del t[p:p + int(rng.integers(1, 60))]
        t = "".join(t) or "A"
        want = oracle.align(s, t)
        assert eng.debug_align(s, t) == want, (s, t)                     # band of 128 diagonals, full matrix when it cannot certify itself
        assert eng.debug_align(s, t, mode=1) == want, (s, t)             # the full matrix alone
        band = eng.debug_align(s, t, mode=2)                             # the band alone: either refuses or is right
        assert band is None or band == want, (s, t)
        n_band += band is not None
    assert n_band >= 20
    eng.close()


def test_overflowed_windows_are_rerun_with_worst_case_workspace(monkeypatch):
    """Tier 1 (small tables) overflows on purpose; the engine must re-run those windows with the worst-case work
    space and still return exactly the oracle's records (never approximate, never silently drop)."""
    monkeypatch.setenv("LANCET_NODE_CAP1", "1024")
    meta, batch, kept, (min_k, max_k) = gu.case_batch("tile60")
    p = abi.default_params(min_k=min_k, max_k=max_k)
    eng = engine.Engine(p)
    variants, stats = eng.process(batch)
    assert eng.rerun_count() > 0
    ov, ostats, _ = oracle.run(batch, p)
    assert variants == ov
    assert [s["status"] for s in stats] == [s["status"] for s in ostats]
    eng.close()


def test_linked_reads_tier2_rerun_and_determinism(monkeypatch):
    """--linked-reads through the worst-case work space (tier 1 forced to overflow) gives the same records as the
    oracle; twice the same batch gives the same answer."""
    meta, batch, kept, (min_k, max_k) = gu.case_batch("lr30")
    p = abi.default_params(min_k=min_k, max_k=max_k, lr_mode=1)
    ov, ostats, _ = oracle.run(batch, p)
    monkeypatch.setenv("LANCET_NODE_CAP1", "256")
    eng = engine.Engine(p)
    a, sa = eng.process(batch)
    assert eng.rerun_count() > 0
    b, sb = eng.process(batch)
    eng.close()
    assert a == ov and a == b and sa == sb


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_engine_matches_oracle_on_random_cycle_prone_windows(seed):
    """The stress inputs of test_emu_kernels.py on the real engine: tandem duplications / STR-rich reference / dense
    variants, k up to the 90s (multi-word keys, rolling-insert path), hundreds of hasCycle and near-repeat decisions."""
    from lancet_amd import frontend, synth
    data = synth.make_tumor_normal(ref_len=4200, cov_t=34, cov_n=28, ref_seed=70 + seed, tumor_seed=170 + seed, normal_seed=270 + seed,
                                   dup_prob=1.0 if seed % 2 == 0 else 0.3, str_fraction=0.25 if seed >= 2 else 0.05,
                                   lowcomplex_fraction=0.05, somatic_every=350, germline_every=260, read_len=100,
                                   insert_mean=230.0, insert_sd=40.0)
    windows = frontend.tile_region(data["ref"], data["rname"], "chr22:500-3600")
    batch, kept = frontend.batch_from_sam(windows, synth.pairs_to_sorted_reads(data["tumor"]), synth.pairs_to_sorted_reads(data["normal"]))
    p = abi.default_params()
    eng = engine.Engine(p, device=0, trace_words=1 << 17)
    v, st = eng.process(batch)
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    assert v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(eng.trace_text()) == gu.digest_trace(otr)
    eng.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_engine_matches_oracle_on_random_linked_read_windows(seed):
    """--linked-reads stress on the real engine (same inputs as test_emu_kernels.py): haplotype counts, barcode sets,
    stats and the -v trace against the oracle."""
    from lancet_amd import frontend, synth
    data = synth.make_tumor_normal(ref_len=3600, cov_t=36, cov_n=30, ref_seed=90 + seed, tumor_seed=190 + seed, normal_seed=290 + seed,
                                   linked=True, dup_prob=0.5 if seed == 1 else 0.0, str_fraction=0.1, somatic_every=400,
                                   germline_every=300, read_len=120, insert_mean=200.0 + 40 * seed, insert_sd=40.0)
    windows = frontend.tile_region(data["ref"], data["rname"], "chr22:500-3000")
    batch, kept = frontend.batch_from_sam(windows, synth.pairs_to_sorted_reads(data["tumor"]), synth.pairs_to_sorted_reads(data["normal"]),
                                          linked=True)
    p = abi.default_params(lr_mode=1)
    eng = engine.Engine(p, device=0, trace_words=1 << 17)
    v, st = eng.process(batch)
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    assert len(ov) > 0 and v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(eng.trace_text()) == gu.digest_trace(otr)
    eng.close()


def _concat(a, b):
    """two WindowBatches one after the other (test helper)"""
    from lancet_amd import frontend
    cat = np.concatenate
    off = lambda x, y: cat([x, (y[1:].astype(np.int64) + int(x[-1])).astype(np.uint32)])
    return frontend.WindowBatch(
        n_windows=a.n_windows + b.n_windows, hdr=a.hdr + b.hdr, chrom=a.chrom + b.chrom, chr_id=cat([a.chr_id, b.chr_id]),
        ref_start=cat([a.ref_start, b.ref_start]), ref_off=off(a.ref_off, b.ref_off), ref_bases=cat([a.ref_bases, b.ref_bases]),
        read_begin=off(a.read_begin, b.read_begin), seq_off=off(a.seq_off, b.seq_off), seq=cat([a.seq, b.seq]), qual=cat([a.qual, b.qual]),
        label=cat([a.label, b.label]), strand=cat([a.strand, b.strand]), mate=cat([a.mate, b.mate]), mapped=cat([a.mapped, b.mapped]),
        name_rank=cat([a.name_rank, b.name_rank]))


def test_coverage_pile_up_window_does_not_size_the_whole_batch():
    """One window at 25x the coverage of the others: the tier-1 work space is laid out for the typical window (the
    pile-up overflows at once and is assembled in the worst-case tier), slot size stays that of the plain batch, and
    every window still equals the oracle."""
    from lancet_amd import workload
    plain = workload.make_scan_batch(48, 20, 20, seed=5, read_len=100)
    pile = workload.make_scan_batch(2, 500, 500, seed=6, read_len=100)
    both = _concat(plain, pile)
    p = abi.default_params()
    eng = engine.Engine(p)
    eng.upload(plain); _, bytes_plain = eng.geometry()
    v, st = eng.process(both)
    _, bytes_both = eng.geometry()
    assert eng.rerun_count() == 2 and all(s["status"] >= 0 for s in st)
    assert bytes_both < 3 * bytes_plain
    ov, ost, _ = oracle.run(both, p)
    assert v == ov and [s["final_k"] for s in st] == [s["final_k"] for s in ost]
    eng.close()


def test_a_few_deep_windows_among_ordinary_ones():
    """Three 60x/60x windows (beyond the 512-lane build configuration) among 64 ordinary ones -- a local coverage spike, the usual case in a
    real scan: the 512-lane kernel lists them, the 1024-lane one builds them, the service stays in the 512-lane configuration (the deep
    windows build their later graphs themselves).  Twice on one engine (the second batch's hand-off areas are the first one's), equal
    to the oracle.  (Sending such a handful to the re-run tier instead of launching the 1024-lane kernel for them was tried: 0.4 ms less
    kernel time per batch, but the tier's set-up copies in lancet_engine_submit wait for the other engine's kernels -- a net loss.)"""
    from lancet_amd import workload
    plain = workload.make_scan_batch(64, 30, 30, seed=7)
    deep = workload.make_scan_batch(3, 60, 60, seed=23)
    both = _concat(plain, deep)
    p = abi.default_params()
    ov, ost, _ = oracle.run(both, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    eng = engine.Engine(p)
    for _ in range(2):
        v, st = eng.process(both)
        assert v == ov and [key(s) for s in st] == [key(s) for s in ost]
        assert not TIER1 or eng.prebuilt_count() + eng.rerun_count() == 67
    eng.close()


def test_pile_up_started_early_does_not_read_the_previous_batch_hand_off():
    """Two different batches through ONE engine; in the second a coverage pile-up sits at a window index whose hand-off area
    still holds a graph of the first batch (PB_BUILT, a k that also suits the pile-up's window).  The pile-up is launched in the
    re-run tier next to the build kernel, i.e. before its own hand-off area is rewritten: it must not look at it (every graph
    by the general build).  Both batches equal the oracle, also when run twice more in alternation."""
    from lancet_amd import workload
    p = abi.default_params()
    first = workload.make_scan_batch(50, 20, 20, seed=5, read_len=100)
    plain = workload.make_scan_batch(48, 20, 20, seed=15, read_len=100)
    pile = workload.make_scan_batch(2, 500, 500, seed=6, read_len=100)
    second = _concat(_concat(workload.sub_batch(plain, 0, 20), pile), workload.sub_batch(plain, 20, 48))     # pile-ups at indices 20, 21
    o1, s1, _ = oracle.run(first, p)
    o2, s2, _ = oracle.run(second, p)
    eng = engine.Engine(p)
    for rnd in range(3):
        v, st = eng.process(first)
        assert v == o1 and [s["final_k"] for s in st] == [s["final_k"] for s in s1], rnd
        v, st = eng.process(second)
        assert (not TIER1 or eng.rerun_count() == 2) and all(s["status"] >= 0 for s in st)
        assert v == o2 and [(s["final_k"], s["n_builds"]) for s in st] == [(s["final_k"], s["n_builds"]) for s in s2], rnd
    eng.close()


def test_build_service_is_scheduling_only(monkeypatch):
    """Cycle-prone windows (tandem duplications: k climbs) with the build service on, with more service workgroups, with nothing
    built ahead (every later graph on request), with the service off, and with the windows taken in another order (engine.hip
    order_class): identical records and per-window statistics, equal to the oracle's; the service did serve requests."""
    from lancet_amd import workload
    p = abi.default_params()
    b = workload.make_scan_batch(600, 30, 30, seed=11)
    ov, ost, _ = oracle.run(b, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    served = []
    # (the last two: the order the window kernel takes the windows in -- two classes as in round 4, batch order -- instead of longest expected first)
    for env in ({}, {"LANCET_SVC_WGS": "64"}, {"LANCET_AHEAD_DEPTH": "0", "LANCET_SVC_DEPTH": "0"}, {"LANCET_NO_SVC": "1"}, {"LANCET_ORDER": "two"}, {"LANCET_NO_HEAVY_FIRST": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        eng = engine.Engine(p)
        for _ in range(2):
            v, st = eng.process(b)
            assert v == ov and [key(s) for s in st] == [key(s) for s in ost], env
        served.append(eng.svc_counts())
        eng.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    if TIER1:
        assert served[0][0] >= 1 and served[0][1] + served[0][3] == served[0][0] and served[3] == (0, 0, 0, 0), served
        extra = sum(s["n_builds"] - 1 for s in ost)
        assert served[2][0] == extra, (served, extra)


def test_two_engines_taking_turns_is_scheduling_only(monkeypatch):
    """Two engines on one GPU, submitted in turn (lancet_engine_submit_after): the second batch's build kernel starts under the end of
    the first batch's window kernel (engine.hip gate_kernel; its first generation of build workgroups leaves after a few windows, scratch
    slots by a bitmap).  Six batches -- enough windows to fill every window slot, a small one, an empty one -- through the pair with the gate
    on (default) and off: the records and statistics of one engine taking the batches one by one, which equal the oracle's on the first."""
    from lancet_amd import workload
    p = abi.default_params()
    batches = [workload.make_scan_batch(n, 30, 30, seed=sd) for n, sd in ((6000, 31), (5000, 32), (300, 33), (6000, 34), (4500, 35))]
    batches.insert(3, workload.sub_batch(batches[2], 0, 0))
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    one = engine.Engine(p)
    want = []
    for b in batches:
        v, st = one.process(b)
        want.append((v, [key(s) for s in st]))
    one.close()
    ov, ost, _ = oracle.run(batches[2], p)
    assert want[2] == (ov, [key(s) for s in ost])
    for env in ({}, {"LANCET_GATE": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        pair = [engine.Engine(p), engine.Engine(p)]
        for rep in range(2):
            got = [None] * len(batches)
            pair[0].upload(batches[0]); pair[0].submit()
            for i in range(1, len(batches) + 1):
                cur, prev = pair[i & 1], pair[(i - 1) & 1]
                if i < len(batches):
                    cur.upload(batches[i]); cur.submit(after=prev)
                prev.wait()
                v, st = prev.results()
                got[i - 1] = (v, [key(s) for s in st])
            assert got == want, (env, rep, [a == b for a, b in zip(got, want)])
        for e_ in pair:
            e_.close()
        for k_ in env:
            monkeypatch.delenv(k_)


def test_host_and_device_trim_and_pack_agree(monkeypatch):
    """Graph_t::trim + packing runs on host threads at upload (pinned staging, one DMA); LANCET_PREP=device keeps the device kernel
    (prep_kernel).  Same records either way, on reads with low-quality ends, N bases and junk reads (golden `filters`-like
    batch: tests/read_variety) and on a scan batch."""
    from lancet_amd import workload
    p = abi.default_params(min_qual_trim=33 + 12, min_qual_call=33 + 20)
    odd = workload.make_scan_batch(64, 25, 25, seed=4, error_rate=0.01)
    rng = np.random.default_rng(8)
    for r in range(odd.n_reads):                                            # low-quality ends, N / IUPAC inside, whole reads of junk
        a, e = int(odd.seq_off[r]), int(odd.seq_off[r + 1])
        u = rng.random()
        if u < 0.15: odd.qual[a:a + int(rng.integers(1, 12))] = 33 + 2
        elif u < 0.30: odd.qual[e - int(rng.integers(1, 12)):e] = 33 + 5
        elif u < 0.35: odd.seq[a + int(rng.integers(0, e - a))] = ord("N")
        elif u < 0.40: odd.seq[a + int(rng.integers(0, e - a))] = ord("R")          # (an IUPAC code: junk like N)
        elif u < 0.43: odd.qual[a:e] = 33
        elif u < 0.46: odd.seq[a] = ord("N"); odd.seq[e - 1] = ord("N")
        elif u < 0.60: odd.qual[a + int(rng.integers(0, e - a))] = 33 + 15     # between the two thresholds: kept, not counted
    batches = [workload.make_scan_batch(96, 25, 25, seed=3, error_rate=0.01), odd]
    for b in batches:
        out = []
        for mode in ("host", "device"):
            monkeypatch.setenv("LANCET_PREP", mode)
            eng = engine.Engine(p)
            v, st = eng.process(b)
            out.append((v, [(s["status"], s["final_k"], s["n_kmers"], s["max_nodes"]) for s in st]))
            eng.close()
        assert out[0] == out[1]
        ov, ost, _ = oracle.run(b, p)
        assert out[0][0] == ov


def test_window_size_1000_inside_ordinary_batches():
    """`--window-size 1000` (reference src/Lancet.cc:662,732: any -w): windows of 600, 1000 and 1024 (= LC_MAXW) bases in one batch --
    work space and hand-off areas laid out for the longest, the full-matrix alignment forced for every path on a second engine
    (LANCET_NO_BAND is not a knob: the band certifies itself or not; the device alignment test covers 1024-base strings) -- equal
    to the oracle in records, stats and trace."""
    from lancet_amd import workload
    p = abi.default_params()
    big = workload.concat_batches([workload.make_scan_batch(40, 30, 30, seed=7), workload.make_scan_batch(48, 30, 30, seed=8, window=1000),
                                   workload.make_scan_batch(16, 30, 30, seed=9, window=1024)])
    ov, ostats, otr = oracle.run(big, p, verbose=True)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    eng = engine.Engine(p, device=0, trace_words=1 << 17)
    for _ in range(2):
        variants, stats = eng.process(big)
        assert variants == ov and [key(s) for s in stats] == [key(s) for s in ostats]
        assert gu.digest_trace(eng.trace_text()) == gu.digest_trace(otr)
    assert all(s["status"] >= 0 for s in stats) and len(ov) > 20
    eng.close()
    # the device alignment on strings of the longest window (full matrix: 16 rows per lane; band; band with fall-back)
    import random
    rnd = random.Random(5)
    S = "".join(rnd.choice("ACGT") for _ in range(1024))
    T = S[:300] + "ACGTTGCA" + S[300:700] + S[720:]
    eng = engine.Engine(p, device=0)
    for mode in (0, 1):
        assert eng.debug_align(S, T, mode=mode) == oracle.align(S, T)
    eng.close()


_KEY = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])


def test_windows_of_more_than_65535_reads():
    """The reference assembles up to MAX_AVG_COV = 10 000x per sample (src/Microassembler.cc:491-496): ~96 000 reads of 150 bases in a 600-base
    window at 11 500x / 11 500x (13 M k-mer occurrences; pairs with overlapping mates by the hundred, so the mate-name vectors of the
    overlap replay hold read ids above 16 bits), and ~95 000 reads of 50 bases, among ordinary windows -- they run in the re-run tier
    next to the rest of the batch and equal the oracle; twice on one engine."""
    from lancet_amd import workload
    plain = workload.make_scan_batch(24, 30, 30, seed=7)
    big150 = workload.make_scan_batch(1, 11500, 11500, seed=5, read_len=150, error_rate=0.0005)
    big50 = workload.make_scan_batch(1, 3100, 3100, seed=8, read_len=50, error_rate=0.0005)
    # (reads of different lengths in one batch: seq_off is per read)
    both = _concat(_concat(plain, big150), big50)
    assert int(both.read_begin[25] - both.read_begin[24]) > 90000 and int(both.read_begin[26] - both.read_begin[25]) > 90000
    p = abi.default_params()
    ov, ost, _ = oracle.run(both, p)
    eng = engine.Engine(p)
    for _ in range(2):
        v, st = eng.process(both)
        assert v == ov and [_KEY(s) for s in st] == [_KEY(s) for s in ost]
        assert not TIER1 or eng.rerun_count() == 2
    eng.close()


def test_linked_read_window_of_more_than_65535_reads():
    """--linked-reads on a window of ~95 000 reads: the barcode replay and getBXsetAt walk csr runs whose read ids need 17 bits."""
    from lancet_amd import workload
    big = workload.make_scan_batch(1, 3100, 3100, seed=8, read_len=50, error_rate=0.0005, linked=True)
    assert int(big.read_begin[1]) > 90000
    p = abi.default_params(); p.lr_mode = 1
    ov, ost, _ = oracle.run(big, p)
    eng = engine.Engine(p)
    v, st = eng.process(big)
    assert v == ov and [_KEY(s) for s in st] == [_KEY(s) for s in ost] and len(ov) > 0
    eng.close()


def test_upload_refuses_labels_the_reference_does_not_have():
    """label / strand / mate outside the reference's values (src/Ref.hh:36-37, src/ReadInfo.hh:30-31) would be read as 'not normal' / 'not
    reverse' by the kernels and as neither sample by anything that tests for equality: refused at upload, LANCET_E_ARG."""
    import copy
    from lancet_amd import workload
    b = workload.make_scan_batch(4, 20, 20, seed=4, read_len=100)
    eng = engine.Engine(abi.default_params())
    for field, bad in (("label", 0), ("strand", 0), ("mate", 3)):
        bb = copy.copy(b)
        a = getattr(b, field).copy(); a[5] = bad
        setattr(bb, field, a)
        with pytest.raises(engine.EngineError, match="label must be"):
            eng.upload(bb)
    v, st = eng.process(b)                                        # (the engine is usable after a refused upload)
    ov, ost, _ = oracle.run(b, abi.default_params())
    assert v == ov
    eng.close()


def test_a_window_beyond_the_engine_limits_fails_alone():
    """One window of 1100 bp (LC_MAXW is 1024) inside an ordinary batch is reported LANCET_W_OVERFLOW on its own: the batch is not refused and
    every other window equals the oracle -- among them one with 70 000 reads, which overflows the one-wave kernel's 16-bit read ids and
    is assembled by the re-run tier (64-bit csr words and mate-name records there: layout.h cs_t)."""
    from lancet_amd import frontend, workload
    p = abi.default_params()
    b = workload.make_scan_batch(24, 20, 20, seed=9, read_len=100)
    # window 5: 500 more reference bases
    ref = bytearray(b.ref_bases.tobytes()); cut = int(b.ref_off[6])
    ref[cut:cut] = (b"ACGTA" * 100)
    ref_off = b.ref_off.astype(np.int64).copy(); ref_off[6:] += 500
    # window 11: its reads 400 times over (70 k reads, names ranked densely)
    r0, r1 = int(b.read_begin[11]), int(b.read_begin[12]); n = r1 - r0; times = 70000 // n + 1
    lens = np.diff(b.seq_off.astype(np.int64))
    def rep(a, per_read=True):
        mid = np.tile(a[r0:r1], times)
        return np.concatenate([a[:r0], mid, a[r1:]])
    s0, s1 = int(b.seq_off[r0]), int(b.seq_off[r1])
    seq = np.concatenate([b.seq[:s0], np.tile(b.seq[s0:s1], times), b.seq[s1:]]); qual = np.concatenate([b.qual[:s0], np.tile(b.qual[s0:s1], times), b.qual[s1:]])
    lens2 = np.concatenate([lens[:r0], np.tile(lens[r0:r1], times), lens[r1:]])
    seq_off = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint32)
    name = np.concatenate([b.name_rank[:r0], (np.arange(n * times) // 1).astype(np.uint32), b.name_rank[r1:]])
    read_begin = b.read_begin.astype(np.int64).copy(); read_begin[12:] += n * (times - 1)
    big = frontend.WindowBatch(n_windows=b.n_windows, hdr=b.hdr, chrom=b.chrom, chr_id=b.chr_id, ref_start=b.ref_start, ref_off=ref_off.astype(np.uint32),
                               ref_bases=np.frombuffer(bytes(ref), dtype=np.uint8), read_begin=read_begin.astype(np.uint32), seq_off=seq_off, seq=seq, qual=qual,
                               label=rep(b.label), strand=rep(b.strand), mate=rep(b.mate), mapped=rep(b.mapped), name_rank=name)
    eng = engine.Engine(p)
    v, st = eng.process(big)
    assert st[5]["status"] < 0
    ov, ost, _ = oracle.run(big, p)
    keep = [w for w in range(b.n_windows) if w != 5]
    assert all(st[w]["status"] >= 0 for w in keep)
    assert [x for x in v if x["window"] in keep] == [x for x in ov if x["window"] in keep]
    assert [_KEY(st[w]) for w in keep] == [_KEY(ost[w]) for w in keep]
    assert int(big.read_begin[12] - big.read_begin[11]) > 65535 and st[11]["n_kmers"] == ost[11]["n_kmers"] > 5_000_000
    eng.close()
