"""The kernel source (lancet_amd/csrc/kernels.h) compiled for the host with the wave emulator
(tests/emu, test infrastructure) must already agree with the oracle and the reference goldens: this is how the
graph logic is debugged on a GPU-less box.  The GPU parity tests are in test_engine_gpu.py."""
import os
import sys

import pytest

import golden_util as gu
from lancet_amd import abi
from oracle import oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402


@pytest.mark.parametrize("case", gu.CASES)
def test_emulated_kernels_match_oracle_and_reference_trace(case):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    p = gu.params(meta)
    v, st, tr = emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, _ = oracle.run(batch, p)
    assert v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(tr) == gu.golden_trace(case)


def test_repeat_scan_bitparallel_matches_bytewise():
    """The LDS/bit-parallel repeat scan (isRepeat / isAlmostRepeat operands, reference src/util.cc:295-360) against
    the byte-wise restatement on random, repetitive and N-containing strings, several mismatch budgets."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_repeat_scan
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    f.restype = None
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 3, 15, 16, 17, 31, 33, 100, 600, 601, 1396):
        cases.append(rng.integers(0, 4, size=n).astype(np.uint8))
    unit = rng.integers(0, 4, size=7).astype(np.uint8)
    cases.append(np.tile(unit, 60)[:400])
    s = rng.integers(0, 4, size=600).astype(np.uint8); s[300:340] = s[100:140]; s[310] ^= 1; cases.append(s)
    s = rng.integers(0, 4, size=600).astype(np.uint8); s[50:90] = 4; cases.append(s)          # run of N
    s = np.zeros(640, dtype=np.uint8); cases.append(s)
    for s in cases:
        for mm in (0, 1, 2, 3, 7, 9):
            out = []
            for bp in (0, 1):
                e, m = ctypes.c_int(-1), ctypes.c_int(-1)
                f(s.ctypes.data, len(s), mm, bp, ctypes.byref(e), ctypes.byref(m))
                out.append((e.value, m.value))
            assert out[0] == out[1], (len(s), mm, out)


def test_repeat_scan_of_a_path_looks_only_around_what_differs_from_the_reference():
    """kernels.h repeats_in_graph_paths (round 5): isAlmostRepeat of a path that differs from the reference between the anchors is decided
    from the windows that overlap the difference (common prefix and suffix taken off), because the window reference passed the same test
    for this k.  On references that pass it, random edits -- substitutions, insertions (random sequence, copies of what is next to them:
    tandem duplications, copies from elsewhere: dispersed repeats), deletions, several edits far apart -- must be decided exactly like
    the full byte-wise scan decides them (reference src/util.cc:317-360)."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_repeat_scan
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    f.restype = None
    g = L.lancet_emu_repeat_scan_range
    g.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    g.restype = None
    rng = np.random.default_rng(55)

    def full_m(s, mm):
        e, m = ctypes.c_int(-1), ctypes.c_int(-1)
        f(s.ctypes.data, len(s), mm, 0, ctypes.byref(e), ctypes.byref(m))
        return m.value

    n_true = n_cases = 0
    for trial in range(400):
        n = int(rng.choice([120, 300, 599, 640]))
        K = int(rng.choice([11, 13, 15, 21, 31, 45]))
        mm = int(rng.choice([0, 1, 2, 2, 3]))
        R = rng.integers(0, 4, size=n).astype(np.uint8)
        if rng.random() < 0.3:                                       # low-complexity stretches: references near the limit
            a = int(rng.integers(0, n - 40)); R[a:a + 40] = np.tile(rng.integers(0, 4, size=int(rng.integers(1, 5))).astype(np.uint8), 40)[:40]
        if n - K <= 0 or full_m(R, mm) >= K + 1:
            continue                                                 # (the loop over k never builds at such a k)
        P = R.copy()
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(1, len(P) - 1)); kind = rng.random()
            if kind < 0.3:
                P[pos] = (P[pos] + 1 + rng.integers(0, 3)) % 4
            elif kind < 0.5:
                ln = int(rng.integers(1, 40)); P = np.concatenate([P[:pos], rng.integers(0, 4, size=ln).astype(np.uint8), P[pos:]])
            elif kind < 0.7:                                         # tandem duplication
                ln = int(rng.integers(1, min(60, pos) + 1)); P = np.concatenate([P[:pos], P[pos - ln:pos], P[pos:]])
            elif kind < 0.8:                                         # a copy from elsewhere
                ln = int(rng.integers(8, 50)); src = int(rng.integers(0, max(1, len(P) - ln))); P = np.concatenate([P[:pos], P[src:src + ln], P[pos:]])
            else:
                ln = int(rng.integers(1, min(40, len(P) - pos - 1) + 1)); P = np.concatenate([P[:pos], P[pos + ln:]])
        P = np.ascontiguousarray(P)
        if len(P) - K <= 0:
            continue
        ml = min(len(P), len(R))
        neq = np.nonzero(P[:ml] != R[:ml])[0]; a = int(neq[0]) if len(neq) else ml
        neq = np.nonzero(P[::-1][:ml] != R[::-1][:ml])[0]; sf = int(neq[0]) if len(neq) else ml
        if len(P) == len(R) and a == ml:
            continue
        if a + sf > ml:
            sf = ml - a
        b = len(P) - sf
        want = full_m(P, mm) >= K + 1
        m = ctypes.c_int(-1)
        g(P.ctypes.data, len(P), mm, K + 1, a - (K + 1), b + (K + 1), ctypes.byref(m))
        assert (m.value >= K + 1) == want, (trial, n, K, mm, a, b, len(P), m.value, full_m(P, mm))
        n_cases += 1; n_true += int(want)
    assert n_cases > 200 and n_true > 20, (n_cases, n_true)


def test_repeat_scan_long_matches_only_decides_like_the_full_scan():
    """repeat_scan_min looks only at match runs long enough to matter for `E >= k` (k >= lminE) and `M >= k + 1` (k + 1 >= lminM):
    every such decision must equal the byte-wise restatement's (reference src/util.cc:295-360)."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_repeat_scan
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    f.restype = None
    g = L.lancet_emu_repeat_scan_min
    g.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    g.restype = None
    rng = np.random.default_rng(9)
    cases = []
    for n in (2, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 599, 600, 608, 640, 1200):
        for _ in range(3):
            cases.append(rng.integers(0, 4, size=n).astype(np.uint8))
    for n in (96, 600):
        for per in (1, 2, 3, 5, 11, 37):
            s = np.tile(rng.integers(0, 4, size=per).astype(np.uint8), n // per + 1)[:n].copy()
            for _ in range(int(rng.integers(0, 6))):
                s[int(rng.integers(0, n))] ^= 1
            cases.append(s)
    s = rng.integers(0, 4, size=600).astype(np.uint8); s[300:364] = s[100:164]; s[310] ^= 1; s[340] ^= 2; cases.append(s)
    s = rng.integers(0, 4, size=600).astype(np.uint8); s[568:600] = s[10:42]; cases.append(s)                 # match up to the last base
    s = rng.integers(0, 4, size=592).astype(np.uint8); s[560:592] = s[0:32]; cases.append(s)
    s = rng.integers(0, 4, size=600).astype(np.uint8); s[50:90] = 4; cases.append(s)
    for s in cases:
        for mm in (0, 1, 2, 3, 7):
            e0, m0 = ctypes.c_int(-1), ctypes.c_int(-1)
            f(s.ctypes.data, len(s), mm, 0, ctypes.byref(e0), ctypes.byref(m0))
            # forms: 4 bits per base; 2 bits per base staged from the bytes (back to 4 when the string holds an N); 2 bits per base
            # from a packed copy (the LDS build kernel's call: strings without N only)
            forms = (0, 1) if (s > 3).any() or len(s) > 640 else (0, 1, 2)
            for k0 in (3, 7, 11, 12, 13, 25, 61):
                for form in forms:
                    e, m = ctypes.c_int(-1), ctypes.c_int(-1)
                    g(s.ctypes.data, len(s), mm, k0, k0 + 1, ctypes.byref(e), ctypes.byref(m), form)
                    assert max(e.value, k0 - 1) == max(e0.value, k0 - 1), (len(s), mm, k0, form, e.value, e0.value)
                    assert max(m.value, k0) == max(m0.value, k0), (len(s), mm, k0, form, m.value, m0.value)
                    g(s.ctypes.data, len(s), mm, 0x7FFF, k0 + 1, ctypes.byref(e), ctypes.byref(m), form)     # the path scan: only M is asked
                    assert max(m.value, k0) == max(m0.value, k0), (len(s), mm, k0, form, m.value, m0.value)


def test_repeat_scan_planes_on_planted_near_repeats():
    """The bit-plane form of repeat_scan_min (round 6: 64 positions per word, a long run sized from the 64 positions either side of it):
    planted copies with 0..4 substitutions at random places -- runs longer than a word, mismatches on word boundaries, copies that end at
    the last base -- and low-complexity stretches, against the byte-wise restatement (reference src/util.cc:295-360)."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_repeat_scan
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    f.restype = None
    g = L.lancet_emu_repeat_scan_min
    g.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    g.restype = None
    rng = np.random.default_rng(606)
    for trial in range(160):
        n = int(rng.choice([70, 128, 129, 191, 192, 257, 600, 640, 1000, 1400]))
        s = rng.integers(0, 4, size=n).astype(np.uint8)
        for _ in range(int(rng.integers(1, 4))):
            ln = int(rng.integers(8, min(200, n // 2)))
            src = int(rng.integers(0, n - ln)); dst = int(rng.integers(0, n - ln))
            if rng.random() < 0.3:
                dst = n - ln                                          # up to the last base
            seg = s[src:src + ln].copy()
            for _ in range(int(rng.integers(0, 5))):
                seg[int(rng.integers(0, ln))] ^= int(rng.integers(1, 4))
            s[dst:dst + ln] = seg
        if rng.random() < 0.3:
            a = int(rng.integers(0, n - 60)); s[a:a + 60] = np.tile(rng.integers(0, 4, size=int(rng.integers(1, 4))).astype(np.uint8), 60)[:60]
        for mm in (0, 1, 2, 3, 5):
            e0, m0 = ctypes.c_int(-1), ctypes.c_int(-1)
            f(s.ctypes.data, n, mm, 0, ctypes.byref(e0), ctypes.byref(m0))
            for k0 in (5, 11, 13, 31):
                for form in ((1, 2) if n <= 640 else (1,)):
                    e, m = ctypes.c_int(-1), ctypes.c_int(-1)
                    g(s.ctypes.data, n, mm, k0, k0 + 1, ctypes.byref(e), ctypes.byref(m), form)
                    assert max(e.value, k0 - 1) == max(e0.value, k0 - 1), (trial, n, mm, k0, form, e.value, e0.value)
                    assert max(m.value, k0) == max(m0.value, k0), (trial, n, mm, k0, form, m.value, m0.value)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_emulated_kernels_match_oracle_on_random_cycle_prone_windows(seed):
    """Not reference goldens but oracle-pinned stress: tandem duplications, STR-rich reference, dense variants -- many
    hasCycle / near-repeat / k-bump decisions per window (the oracle takes them the reference's way, on the k-mer graph,
    the kernels on the compacted graph) -- everything must still agree: records, stats and the whole stage trace."""
    from lancet_amd import frontend, synth
    data = synth.make_tumor_normal(ref_len=4200, cov_t=34, cov_n=28, ref_seed=70 + seed, tumor_seed=170 + seed, normal_seed=270 + seed,
                                   dup_prob=1.0 if seed % 2 == 0 else 0.3, str_fraction=0.25 if seed >= 2 else 0.05,
                                   lowcomplex_fraction=0.05, somatic_every=350, germline_every=260, read_len=100,
                                   insert_mean=230.0, insert_sd=40.0)
    windows = frontend.tile_region(data["ref"], data["rname"], "chr22:500-3600")
    batch, kept = frontend.batch_from_sam(windows, synth.pairs_to_sorted_reads(data["tumor"]), synth.pairs_to_sorted_reads(data["normal"]))
    p = abi.default_params()
    v, st, tr = emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    assert batch.n_windows >= 5 and v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(tr) == gu.digest_trace(otr)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_emulated_kernels_match_oracle_on_random_linked_read_windows(seed):
    """--linked-reads stress (oracle-pinned): random barcodes / haplotypes, overlapping mates, dense variants; records
    (haplotype counts, barcode sets), stats and the stage trace (HPref / HPalt per transcript) must agree."""
    from lancet_amd import frontend, synth
    data = synth.make_tumor_normal(ref_len=3600, cov_t=36, cov_n=30, ref_seed=90 + seed, tumor_seed=190 + seed, normal_seed=290 + seed,
                                   linked=True, dup_prob=0.5 if seed == 1 else 0.0, str_fraction=0.1, somatic_every=400,
                                   germline_every=300, read_len=120, insert_mean=200.0 + 40 * seed, insert_sd=40.0)
    windows = frontend.tile_region(data["ref"], data["rname"], "chr22:500-3000")
    batch, kept = frontend.batch_from_sam(windows, synth.pairs_to_sorted_reads(data["tumor"]), synth.pairs_to_sorted_reads(data["normal"]),
                                          linked=True)
    p = abi.default_params(lr_mode=1)
    v, st, tr = emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    assert batch.n_windows >= 5 and len(ov) > 0 and v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(tr) == gu.digest_trace(otr)


@pytest.mark.parametrize("route", ["small", "large", "fat"])
def test_linked_read_windows_through_the_lds_build_and_the_replay_over_its_runs(route, monkeypatch):
    """--linked-reads windows whose first graph comes from the LDS build kernel (build_lds_impl.h: the tracked nodes' occurrences are handed
    over as csr runs; kernels.h load_prebuilt_lr replays barcodes / haplotypes over them, recounts the ten per-position counters of the
    survivors, recomputes Ref_t's tables from the barcode counts and builds the table getBXsetAt looks k-mers up in) -- the windows of
    bench.py's config 5, both build configurations and the re-run tier's source: records (haplotype counts, barcode sets), stats and the
    whole stage trace equal the oracle's, and equal the general build of every window (LANCET_LR_PREBUILD=0, the route until round 5)."""
    from lancet_amd import workload
    b = workload.make_scan_batch(10, 30, 30, seed=5 if route != "large" else 6, linked=True, somatic_every=700, germline_every=400)
    p = abi.default_params(lr_mode=1)
    if route == "large":
        monkeypatch.setenv("LANCET_EMU_FORCE_LARGE", "1")
    emu.FAT[0] = route == "fat"
    try:
        v, st, tr = emu.run(b, p, evt_cap=1 << 18)
        built = emu.LAST_PREBUILT[0]
        ov, ost, otr = oracle.run(b, p, verbose=True)
        assert built == b.n_windows and len(ov) > 3 and v == ov
        assert any(any(len(x) for x in r["bx"]) for r in ov) and any(any(r["hp"]) for r in ov)
        key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
        assert [key(s) for s in st] == [key(s) for s in ost]
        assert gu.digest_trace(tr) == gu.digest_trace(otr)
        monkeypatch.setenv("LANCET_LR_PREBUILD", "0")
        v2, st2, tr2 = emu.run(b, p, evt_cap=1 << 18)
        assert emu.LAST_PREBUILT[0] == 0 and v2 == ov and gu.digest_trace(tr2) == gu.digest_trace(otr)
    finally:
        emu.FAT[0] = False


@pytest.mark.parametrize("seed", [0, 2])
def test_linked_read_windows_that_climb_through_many_k_take_graphs_built_ahead_and_by_the_service(seed):
    """--linked-reads with tandem duplications everywhere: windows that reject a dozen k.  Their later graphs come from the build kernel too
    (built ahead, or by the build service on request) -- with Ref_t::seq trimmed by the k attempts before, which load_prebuilt_lr's mer-table
    bits and coverage tables have to follow -- and everything still equals the oracle: records, stats, the whole trace."""
    from lancet_amd import frontend, synth
    data = synth.make_tumor_normal(ref_len=4200, cov_t=34, cov_n=28, ref_seed=70 + seed, tumor_seed=170 + seed, normal_seed=270 + seed, linked=True,
                                   dup_prob=1.0, str_fraction=0.05, somatic_every=350, germline_every=260, read_len=100, insert_mean=330.0, insert_sd=30.0)
    windows = frontend.tile_region(data["ref"], data["rname"], "chr22:500-3600")
    batch, kept = frontend.batch_from_sam(windows, synth.pairs_to_sorted_reads(data["tumor"]), synth.pairs_to_sorted_reads(data["normal"]), linked=True)
    p = abi.default_params(lr_mode=1)
    v, st, tr = emu.run(batch, p, evt_cap=1 << 18)
    taken, served = emu.LAST_AHEAD[1], emu.LAST_SVC[1]
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    assert emu.LAST_PREBUILT[0] >= 10 and taken >= 40 and served >= 40 and max(s["n_builds"] for s in ost) >= 9
    assert len(ov) > 0 and v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(tr) == gu.digest_trace(otr)


@pytest.mark.parametrize("name", ["dups", "nref"])
def test_table_doubling_path_gives_the_same_graphs(name, monkeypatch):
    """Each build sizes its k-mer table from an estimate and doubles it when it fills up; started at 64 slots, every
    build of every window goes through several doublings and must end with the same records, stats and trace
    (dups: k up to the 90s through the rolling-insert + verify passes; nref: N k-mers)."""
    meta, batch, kept, (min_k, max_k) = gu.case_batch(name)
    p = abi.default_params(min_k=min_k, max_k=max_k)
    base = emu.run(batch, p, evt_cap=1 << 17)
    monkeypatch.setenv("LANCET_TABLE_START", "64")
    grown = emu.run(batch, p, evt_cap=1 << 17)
    assert grown[0] == base[0] and grown[1] == base[1] and gu.digest_trace(grown[2]) == gu.digest_trace(base[2])


def test_emulated_kernels_on_a_coverage_pile_up():
    """Nodes with more occurrences than the LDS staging area of the per-position pass (rounds with carried counts, whose
    buffer shares its space with the candidate-list cache)."""
    from lancet_amd import workload
    pile = workload.make_scan_batch(2, 300, 300, seed=6, read_len=100)
    p = abi.default_params()
    v, st, _ = emu.run(pile, p)
    ov, ost, _ = oracle.run(pile, p)
    assert v == ov and [s["final_k"] for s in st] == [s["final_k"] for s in ost] and len(ov) > 0


def test_find_tandems_local_equals_the_oracle():
    """findTandems (reference src/util.cc:574-758) evaluated from the neighbourhood of the position only must report what the
    oracle's whole-string scan reports: answer, length (last report wins) and the concatenated motifs, on random, STR-rich and periodic
    strings, at every kind of position (string ends included) and under several option sets."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_find_tandems
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    f.restype = ctypes.c_int
    rng = np.random.default_rng(17)

    def run(codes, pos, opt, local):
        ln, ml = ctypes.c_int(0), ctypes.c_int(0)
        motif = np.zeros(64, dtype=np.uint8)
        a = f(codes.ctypes.data, len(codes), pos, opt[0], opt[1], opt[2], opt[3], local, ctypes.byref(ln), motif.ctypes.data, ctypes.byref(ml))
        return bool(a), ln.value, "".join("ACGT"[x] for x in motif[:ml.value])

    strings = []
    for n in (1, 2, 5, 9, 30, 120, 600):
        strings.append(rng.integers(0, 4, size=n).astype(np.uint8))
    for _ in range(12):                                   # STR blocks of unit 1..5 in random sequence, some interrupted
        parts = []
        while sum(len(p) for p in parts) < 300:
            if rng.random() < 0.5:
                parts.append(rng.integers(0, 4, size=int(rng.integers(1, 25))).astype(np.uint8))
            else:
                u = rng.integers(0, 4, size=int(rng.integers(1, 6))).astype(np.uint8)
                blk = np.tile(u, int(rng.integers(2, 15)))
                if rng.random() < 0.3 and len(blk) > 4:
                    blk[int(rng.integers(0, len(blk)))] ^= 1
                parts.append(blk)
        strings.append(np.concatenate(parts))
    strings.append(np.tile(np.array([0, 1], dtype=np.uint8), 60))
    strings.append(np.zeros(80, dtype=np.uint8))
    opts = [(4, 3, 7, 1), (3, 2, 5, 2), (6, 1, 2, 0), (1, 3, 7, 4), (8, 2, 4, 1)]
    checked = 0
    for s in strings:
        n = len(s)
        text = "".join("ACGT"[x] for x in s)
        positions = sorted(set([0, 1, n - 1, n, n // 2] + [int(x) for x in rng.integers(0, n + 1, size=min(n + 1, 40))]))
        for opt in opts:
            for pos in positions:
                loc = run(s, pos, opt, 1)
                a, ln, mo = oracle.find_tandems(text, pos, *opt)
                assert (a, ln if a else 0, mo[:60]) == (loc[0], loc[1] if loc[0] else 0, loc[2]), (text, pos, opt, (a, ln, mo), loc)
                checked += 1
    assert checked > 2000


def test_banded_alignment_equals_full_matrix_and_oracle():
    """global_align_aff (reference src/align.cc:235-364) over a band of 128 diagonals, when the band certifies itself, must give the
    aligned strings of the full matrix (and of the oracle, which test_oracle_golden pins on the reference's own align.cc); pairs the
    band cannot hold must be refused (the full matrix then runs)."""
    import ctypes
    import numpy as np
    L = emu.lib()
    f = L.lancet_emu_align
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    f.restype = ctypes.c_int
    rng = np.random.default_rng(23)

    def rs(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))

    def run(s, t, mode):
        cap = len(s) + len(t) + 8
        a, b = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
        r = f(s.encode(), t.encode(), a, b, cap, mode)
        if r == -1:
            return "undefined"          # the traceback left the matrix: undefined behaviour in the reference (reported as an overflow by the engine)
        return None if r == -2 else (a.value.decode(), b.value.decode())

    n_band = n_refused = 0
    for it in range(160):
        n = int(rng.integers(2, 640))
        s = rs(n)
        t = list(s)
        for _ in range(int(rng.integers(0, 7))):
            p = int(rng.integers(0, max(1, len(t))))
            r = rng.random()
            if r < 0.3 and t:
                t[p] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.6:
                t[p:p] = list(rs(int(rng.integers(1, 90))))
            elif r < 0.9 and t:
                del t[p:p + int(rng.integers(1, 90))]
            elif t:                                           # tandem duplication / STR: many equally good alignments
                u = t[max(0, p - 6):p] or ["A"]
                t[p:p] = u * int(rng.integers(1, 6))
        t = "".join(t) or "A"
        if it % 9 == 0:
            t = rs(int(rng.integers(2, 640)))                  # unrelated strings: the band must not certify a wrong answer
        full = run(s, t, 1)
        band = run(s, t, 2)
        if full != "undefined":
            assert full == oracle.align(s, t), (s, t)
        if band is None:
            n_refused += 1
        else:
            n_band += 1
            assert band == full, (s, t, band, full)
        assert run(s, t, 0) == full
    assert n_band > 60 and n_refused > 5, (n_band, n_refused)


def _same_as_oracle(batch, p, trace=True):
    v, st, tr = emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, otr = oracle.run(batch, p, verbose=True)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert v == ov
    assert [key(s) for s in st] == [key(s) for s in ost]
    if trace:
        assert gu.digest_trace(tr) == gu.digest_trace(otr)
    return st


def test_lds_build_kernel_and_graphs_built_ahead_on_the_bench_workload(monkeypatch):
    """The two-kernel path on the kind of windows bench.py measures (30x/30x scan): every window's first graph comes from the
    LDS build kernel, the windows whose k is going to be rejected get the next graphs built ahead and the window kernel takes
    them; a window whose next graph nobody built ahead is suspended and gets it from the build service.  Records, stats and the
    stage trace equal the oracle's.  Then the same windows with nothing built ahead (LANCET_AHEAD_DEPTH=0: every later graph on
    request), with a service nobody runs (the slots take their requests back: general build), without the service, and with no
    LDS build at all (LANCET_NO_PREBUILD): same results -- what is built where is scheduling only."""
    from lancet_amd import workload
    b = workload.make_scan_batch(400, 30, 30, seed=5)
    p = abi.default_params()
    st = _same_as_oracle(b, p)
    extra = sum(s["n_builds"] - 1 for s in st)
    assert emu.LAST_PREBUILT[0] == 400
    assert extra >= 10 and emu.LAST_AHEAD[1] >= extra // 2 and emu.LAST_AHEAD[0] <= 3 * extra, (extra, emu.LAST_AHEAD)
    base = emu.run(b, p, evt_cap=1 << 17)
    monkeypatch.setenv("LANCET_STOP_PHASE", "130")          # (knob: 16-bit instead of 8-bit per-position counters in the build kernel)
    wide = emu.run(b, p, evt_cap=1 << 17)
    monkeypatch.delenv("LANCET_STOP_PHASE")
    assert wide[0] == base[0] and wide[1] == base[1] and gu.digest_trace(wide[2]) == gu.digest_trace(base[2])
    assert emu.LAST_SVC[0] >= 1 and emu.LAST_SVC[1] == emu.LAST_SVC[0] and emu.LAST_SVC[2] == 0, emu.LAST_SVC
    monkeypatch.setenv("LANCET_AHEAD_DEPTH", "0")
    monkeypatch.setenv("LANCET_SVC_DEPTH", "0")
    on_request = emu.run(b, p, evt_cap=1 << 17)
    assert emu.LAST_SVC[0] == extra and emu.LAST_SVC[1] == extra and emu.LAST_AHEAD == [0, extra], (extra, emu.LAST_SVC, emu.LAST_AHEAD)
    monkeypatch.setenv("LANCET_SVC_DEAD", "1")
    taken_back = emu.run(b, p, evt_cap=1 << 17)
    assert emu.LAST_SVC[0] == extra and emu.LAST_SVC[1] == 0 and emu.LAST_SVC[2] == extra, (extra, emu.LAST_SVC)
    monkeypatch.delenv("LANCET_SVC_DEAD")
    monkeypatch.setenv("LANCET_NO_SVC", "1")
    plain = emu.run(b, p, evt_cap=1 << 17)
    assert emu.LAST_AHEAD == [0, 0] and emu.LAST_SVC == [0, 0, 0] and emu.LAST_PREBUILT[0] == 400
    monkeypatch.setenv("LANCET_NO_PREBUILD", "1")
    general = emu.run(b, p, evt_cap=1 << 17)
    assert emu.LAST_PREBUILT[0] == 0
    for other in (on_request, taken_back, plain, general):
        assert other[0] == base[0] and other[1] == base[1] and gu.digest_trace(other[2]) == gu.digest_trace(base[2])


def test_graph_built_ahead_is_adjusted_to_the_trimmed_reference_table():
    """tests/golden/ahead_trim.npz (tools/make_ahead_fixture.py): windows where a graph built ahead is taken after the rejected
    k had trimmed Ref_t::seq; the table of reference k-mers is re-indexed over the trimmed seq at every k (reference
    src/Ref.cc:40-64), which changes which nodes are reference nodes and the coverage computeCoverage reads for the
    variants at the window's edges."""
    p = abi.default_params()
    used = 0
    for b in gu.load_batches_npz("ahead_trim.npz"):
        _same_as_oracle(b, p)
        used += emu.LAST_AHEAD[1]
    assert used >= 4


def test_deep_windows_through_the_1024_lane_build_configuration(monkeypatch):
    """60x/60x windows are above the 512-lane configuration's limits (40 960 bases in LDS): the first build kernel lists them and
    the 1024-lane configuration (131 040 bases under 17-bit offsets, one workgroup per CU) builds them, graphs built ahead
    included -- all of them: windows 94-96 of this batch need the mate-overlap replay of 6417 occurrences (the 512-lane
    configuration's list holds 4096).  Then 100x/100x windows (640 reads, 100 k bases: offsets above 65 535), and 30x/30x windows
    forced through the 1024-lane configuration (every size-dependent piece of it on inputs the other one also takes)."""
    from lancet_amd import workload
    p = abi.default_params()
    b = workload.make_scan_batch(100, 60, 60, seed=22)
    _same_as_oracle(b, p)
    assert emu.LAST_BIGLIST[0] == 100 and emu.LAST_PREBUILT[0] == 100
    b100 = workload.make_scan_batch(24, 100, 100, seed=5, error_rate=0.001)
    assert min(b100.seq_off[b100.read_begin[w + 1]] - b100.seq_off[b100.read_begin[w]] for w in range(24)) > 70000
    _same_as_oracle(b100, p)
    assert emu.LAST_BIGLIST[0] == 24 and emu.LAST_PREBUILT[0] == 24
    monkeypatch.setenv("LANCET_EMU_FORCE_LARGE", "1")
    b30 = workload.make_scan_batch(60, 30, 30, seed=4)
    _same_as_oracle(b30, p)
    assert emu.LAST_PREBUILT[0] == 60


def test_deep_str_windows_build_in_lds_at_k_above_31(monkeypatch):
    """BASELINE config 4 (100x tumor / 40x normal over STR-rich sequence): the windows that find a k build at k = 33..101 with 8-12 k
    distinct k-mers.  The 1024-lane configuration cuts k-mers of up to four 64-bit words out of the LDS reads (CanonicalMer_t::set,
    reference src/Mer.hh:57-71, on multi-word keys), holds 14 336 nodes under a 16 384-slot table and hands candidate keys over in
    `kw` words (wide hand-off areas): every window that builds is built there, records / stats / trace equal the oracle's.  With the
    narrow areas forced (one-word keys) the same windows take the general build: same results."""
    from lancet_amd import workload
    p = abi.default_params()
    seen = set()
    for seed, n in ((22, 60), (202, 64)):
        b = workload.make_scan_batch(n, 100.0, 40.0, seed=seed, str_fraction=0.30, lowcomplex_fraction=0.05)
        st = _same_as_oracle(b, p)
        builds = sum(1 for s in st if s["n_builds"] > 0)
        assert builds >= 20 and emu.LAST_PREBUILT[0] == builds, (builds, emu.LAST_PREBUILT)
        seen |= {(2 * s["final_k"] + 63) // 64 for s in st if s["status"] == 0}
        assert max(s["max_nodes"] for s in st) > 8192
    assert seen >= {2, 3, 4}, seen                                # two-, three- and four-word k-mers all occurred
    monkeypatch.setenv("LANCET_PRE_WIDE", "0")
    b = workload.make_scan_batch(24, 100.0, 40.0, seed=22, str_fraction=0.30, lowcomplex_fraction=0.05)
    _same_as_oracle(workload.sub_batch(b, 12, 24), p)
    assert emu.LAST_PREBUILT[0] == 0


def test_windows_of_1000_bases_among_ordinary_ones():
    """`--window-size 1000` (the reference takes any -w: src/Lancet.cc:662,732): the work space and the hand-off areas are laid out for
    the longest window of the batch (EngineCaps::max_w, up to LC_MAXW = 1024), the full-matrix alignment keeps 16 rows per lane for
    them.  Windows of 600, 1000 and 1024 bases in one batch: records, stats and trace equal the oracle's, all built in LDS."""
    from lancet_amd import workload
    p = abi.default_params()
    big = workload.concat_batches([workload.make_scan_batch(10, 30, 30, seed=7), workload.make_scan_batch(12, 30, 30, seed=8, window=1000),
                                   workload.make_scan_batch(4, 30, 30, seed=9, window=1024)])
    st = _same_as_oracle(big, p)
    assert all(s["status"] >= 0 for s in st) and emu.LAST_PREBUILT[0] == big.n_windows


def test_mate_overlap_replay_in_ranges_of_nodes(monkeypatch):
    """hasOverlappingMate's exact replay in the LDS build kernel sorts the occurrences of the marked nodes in an LDS list; when they do
    not fit, the nodes are taken in ranges.  170-base reads at a 400 +- 40 insert: a tenth of the pairs overlap (~200 occurrences in
    32 windows are overlapping-mate occurrences, all windows built in LDS).  With the list cut to 512 entries (LANCET_STOP_PHASE=131)
    every window goes through several ranges: same records, stats and trace.  (The `ovl` golden, whose mates overlap almost always,
    flags more occurrences than the build kernel keeps and runs on the general build.)"""
    from lancet_amd import workload
    b = workload.make_scan_batch(32, 30, 30, seed=8, read_len=170)
    p = abi.default_params()
    _same_as_oracle(b, p)
    assert emu.LAST_PREBUILT[0] == 32
    base = emu.run(b, p, evt_cap=1 << 17)
    monkeypatch.setenv("LANCET_STOP_PHASE", "131")
    cut = emu.run(b, p, evt_cap=1 << 17)
    assert emu.LAST_PREBUILT[0] == 32 and cut[0] == base[0] and cut[1] == base[1] and gu.digest_trace(cut[2]) == gu.digest_trace(base[2])


# ---- the source of the several-wave re-run tier (window_fat.hip = kernels.h with LANCET_FAT): its own code paths -- per-position
#      counts split over lane groups, the mate-overlap prefilter six reads at a time, the larger staging area -- emulated lane after
#      lane with every graph built by the general build (LANCET_NO_PREBUILD).  The races a real workgroup could add are the GPU
#      suite's business (tests/test_engine_gpu*.py under LANCET_NODE_CAP1=64); this pins the logic.
@pytest.fixture
def fat_emu(monkeypatch):
    monkeypatch.setenv("LANCET_NO_PREBUILD", "1")
    emu.FAT[0] = True
    yield emu
    emu.FAT[0] = False


_KEY = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])


@pytest.mark.parametrize("case", gu.CASES)
def test_fat_source_matches_oracle_and_reference_trace(case, fat_emu):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    p = gu.params(meta)
    v, st, tr = fat_emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, _ = oracle.run(batch, p)
    assert v == ov
    assert [_KEY(s) for s in st] == [_KEY(s) for s in ost]
    assert gu.digest_trace(tr) == gu.golden_trace(case)


def test_fat_source_on_coverage_pile_ups(fat_emu):
    """Candidates with more occurrences than even the larger staging area (1536): rounds with the counts carried in LDS across
    them, groups of one or two candidates split over lane groups; overlapping mates by the hundred for the grouped prefilter."""
    from lancet_amd import workload
    p = abi.default_params()
    for pile in (workload.make_scan_batch(1, 1300, 1300, seed=6, read_len=100), workload.make_scan_batch(2, 300, 300, seed=6, read_len=100),
                 workload.make_scan_batch(3, 150, 90, seed=9, read_len=250)):
        v, st, _ = fat_emu.run(pile, p)
        ov, ost, _ = oracle.run(pile, p)
        assert v == ov and [_KEY(s) for s in st] == [_KEY(s) for s in ost]


def test_linked_read_replay_by_the_wave_sorts_runs_that_arrive_out_of_order(monkeypatch):
    """lr_replay_batches takes a node's csr run as the csr pass left it -- in ARRIVAL order of the lanes' atomics, which on the device is
    not always the visiting order; the emulated lanes run one after the other and always deliver sorted runs.  With LANCET_STOP_PHASE=141
    every staged run is read backwards and must come out sorted by rank: same records, same trace."""
    meta, batch, kept, _ = gu.case_batch("lr30")
    p = gu.params(meta)
    monkeypatch.setenv("LANCET_NO_PREBUILD", "1")
    base = emu.run(batch, p, evt_cap=1 << 17)
    monkeypatch.setenv("LANCET_STOP_PHASE", "141")
    back = emu.run(batch, p, evt_cap=1 << 17)
    assert back[0] == base[0] and [_KEY(s) for s in back[1]] == [_KEY(s) for s in base[1]] and gu.digest_trace(back[2]) == gu.digest_trace(base[2])
    assert gu.digest_trace(base[2]) == gu.golden_trace("lr30") and len(base[0]) > 0


def test_a_read_longer_than_the_position_field_fails_its_window_alone():
    """k-mer positions are 10 bits in the occurrence words of both build paths: a window that holds a read of 1100 + bases is reported
    LANCET_W_OVERFLOW (never assembled with wrapped positions), the windows beside it equal the oracle."""
    import numpy as np
    from lancet_amd import frontend, workload
    b = workload.make_scan_batch(3, 20, 20, seed=4, read_len=100)
    rng = np.random.default_rng(1)
    L = 1150
    r1 = int(b.read_begin[2])                                    # the new read becomes the last one of window 1
    s1 = int(b.seq_off[r1])
    ins = lambda a, v: np.concatenate([a[:r1], np.array([v], dtype=a.dtype), a[r1:]])
    seq = np.concatenate([b.seq[:s1], np.frombuffer(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)), dtype=np.uint8), b.seq[s1:]])
    qual = np.concatenate([b.qual[:s1], np.full(L, ord("I"), dtype=np.uint8), b.qual[s1:]])
    lens = np.diff(b.seq_off.astype(np.int64)); lens = np.concatenate([lens[:r1], [L], lens[r1:]])
    read_begin = b.read_begin.astype(np.int64).copy(); read_begin[2:] += 1
    big = frontend.WindowBatch(n_windows=3, hdr=b.hdr, chrom=b.chrom, chr_id=b.chr_id, ref_start=b.ref_start, ref_off=b.ref_off, ref_bases=b.ref_bases,
                               read_begin=read_begin.astype(np.uint32), seq_off=np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32), seq=seq, qual=qual,
                               label=ins(b.label, b.label[0]), strand=ins(b.strand, b.strand[0]), mate=ins(b.mate, 0), mapped=ins(b.mapped, 1),
                               name_rank=ins(b.name_rank, int(b.read_begin[2] - b.read_begin[1])))
    p = abi.default_params()
    ov, ost, _ = oracle.run(b, p)
    for fat in (False, True):
        emu.FAT[0] = fat
        try:
            v, st, _ = emu.run(big, p)
        finally:
            emu.FAT[0] = False
        assert st[1]["status"] < 0 and st[0]["status"] >= 0 and st[2]["status"] >= 0
        assert [x for x in v if x["window"] != 1] == [x for x in ov if x["window"] != 1]


def test_a_kmer_with_more_occurrences_than_a_16_bit_count_is_counted_as_the_reference_counts_it():
    """800 reads of 150 A's piled on one window: the k-mer A^k occurs ~110 000 times, more than the 16-bit occurrence counts of the 1024-lane
    build kernel (which must turn the window away instead of counting modulo 65 536 in ITS counters) and more than the reference's own
    unsigned short per-position counters -- which wrap (src/Ref.hh:43-52, Node_t::updateCovDistr src/Node.cc:470-497).  The general build
    counts them modulo 65 536 as the reference does (round 5; it reported the window LANCET_W_OVERFLOW before): equal to the oracle."""
    import numpy as np
    from lancet_amd import frontend, workload
    b = workload.make_scan_batch(3, 10, 10, seed=4, read_len=100)
    L, M = 150, 800
    r1 = int(b.read_begin[2]); s1 = int(b.seq_off[r1])
    insn = lambda a, v: np.concatenate([a[:r1], np.full(M, v, dtype=a.dtype), a[r1:]])
    seq = np.concatenate([b.seq[:s1], np.full(L * M, ord("A"), dtype=np.uint8), b.seq[s1:]])
    qual = np.concatenate([b.qual[:s1], np.full(L * M, ord("I"), dtype=np.uint8), b.qual[s1:]])
    lens = np.diff(b.seq_off.astype(np.int64)); lens = np.concatenate([lens[:r1], np.full(M, L), lens[r1:]])
    read_begin = b.read_begin.astype(np.int64).copy(); read_begin[2:] += M
    n1 = int(b.read_begin[2] - b.read_begin[1])
    name = np.concatenate([b.name_rank[:r1], (n1 + np.arange(M)).astype(np.uint32), b.name_rank[r1:]])
    big = frontend.WindowBatch(n_windows=3, hdr=b.hdr, chrom=b.chrom, chr_id=b.chr_id, ref_start=b.ref_start, ref_off=b.ref_off, ref_bases=b.ref_bases,
                               read_begin=read_begin.astype(np.uint32), seq_off=np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32), seq=seq, qual=qual,
                               label=insn(b.label, b.label[0]), strand=insn(b.strand, 0), mate=insn(b.mate, 0), mapped=insn(b.mapped, 1), name_rank=name)
    p = abi.default_params()
    ov, ost, _ = oracle.run(big, p)
    v, st, _ = emu.run(big, p)
    assert emu.LAST_PREBUILT[0] == 2                              # (windows 0 and 2: the build kernel; window 1 turned away)
    assert v == ov and [(x["status"], x["final_k"], x["n_builds"], x["n_kmers"], x["max_nodes"]) for x in st] == [(x["status"], x["final_k"], x["n_builds"], x["n_kmers"], x["max_nodes"]) for x in ost]
    assert st[1]["status"] >= 0
    emu.FAT[0] = True                                            # (the re-run tier's source: its split counts add up in 32 bits and are cut to 16 at the end)
    try:
        v2, st2, _ = emu.run(big, p)
    finally:
        emu.FAT[0] = False
    assert v2 == ov and st2[1]["status"] >= 0


@pytest.mark.parametrize("linked", [False, True])
def test_fat_source_on_a_window_of_more_than_65535_reads(linked, fat_emu):
    """~95 000 reads of 50 bases in one window (3100x / 3100x; the reference goes up to MAX_AVG_COV = 10 000x per sample): the re-run
    tier's csr words, mate-name records and work items keep the read in 32 / 21 bits (layout.h cs_t) -- the one-wave source refuses it."""
    from lancet_amd import workload
    big = workload.make_scan_batch(1, 3100, 3100, seed=8, read_len=50, error_rate=0.0005, linked=linked)
    assert int(big.read_begin[1]) > 90000
    p = abi.default_params(); p.lr_mode = 1 if linked else 0
    ov, ost, _ = oracle.run(big, p)
    v, st, _ = fat_emu.run(big, p)
    assert v == ov and [_KEY(s) for s in st] == [_KEY(s) for s in ost] and len(ov) > 0
    if not linked:
        emu.FAT[0] = False
        _, st1, _ = emu.run(big, p)
        assert st1[0]["status"] < 0


@pytest.mark.parametrize("seed", [0, 1])
def test_fat_source_on_random_linked_read_windows(seed, fat_emu):
    test_emulated_kernels_match_oracle_on_random_linked_read_windows(seed)
