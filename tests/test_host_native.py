"""The native host side (include/lancet_host.h: BAM / FASTA input, tiling, window filters, read selection, batch
assembly) against the Python front end it replaces (lancet_amd/frontend.py + bamio.py, themselves pinned on
reference-made fixtures by test_cli.py / test_frontend*.py): identical batches array for array.  CPU only."""
import io
import os
import subprocess

import numpy as np
import pytest

import bam_writer
import read_variety
import golden_util as gu
from lancet_amd import bamio, build, frontend, host, synth

G = gu.GOLDEN
FIELDS = ("chr_id", "ref_start", "ref_off", "ref_bases", "read_begin", "seq_off", "seq", "qual", "label", "strand", "mate", "mapped", "name_rank")


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build()


def _python_batch(tumor_bam, normal_bam, fasta, region, o, fp=None):
    _, tumor = bamio.read_bam(tumor_bam)
    _, normal = bamio.read_bam(normal_bam)
    chrom = region.split(":")[0]
    tumor = [r for r in tumor if r.rname == chrom]
    normal = [r for r in normal if r.rname == chrom]
    contigs = bamio.read_fasta(fasta)
    wins = frontend.tile_region(contigs[chrom], chrom, region, padding=o.padding, window_size=o.window_size)
    fp = fp or frontend.ReadFilterParams(min_map_qual=o.min_map_qual, max_delta_as_xs=o.max_delta_as_xs,
                                         primary_alignment_only=bool(o.primary_alignment_only), xa_filter=bool(o.xa_filter),
                                         max_avg_cov=o.max_avg_cov)
    pb, kept = frontend.batch_from_sam(wins, tumor, normal, fp, max_k=o.max_k, linked=bool(o.linked), active_region=bool(o.active_region),
                                       min_evidence=o.min_evidence, min_qual_call=o.min_qual_call)
    return [w.hdr for w in frontend.windows_in_processing_order(wins)], pb


def _same(b, pb, linked):
    assert b.n_windows == pb.n_windows and b.hdr == pb.hdr
    for f in FIELDS:
        assert np.array_equal(getattr(b, f), getattr(pb, f)), f
    if linked:
        assert np.array_equal(b.bx_rank, pb.bx_rank) and np.array_equal(b.hp, pb.hp) and b.bx_names == pb.bx_names


@pytest.mark.parametrize("case,region,kw", [
    ("ar_small", "chr22:900-3000", {}),
    ("ar_small", "chr22:900-3000", {"active_region": 0}),
    ("ar_small", "chr22:1200-1900", {"padding": 0, "window_size": 400, "max_k": 31}),
    ("lr_small", "chr22:800-2700", {"linked": 1}),
    ("lr_small", "chr22:800-2700", {"linked": 1, "active_region": 0, "padding": 100}),
    ("lr_small", "chr22", {"active_region": 0}),
])
def test_native_host_batches_equal_the_python_front_end_on_reference_made_bams(case, region, kw):
    paths = [os.path.join(G, f"{case}.tumor.bam"), os.path.join(G, f"{case}.normal.bam"), os.path.join(G, f"{case}.fa")]
    o = host.default_opts(**kw)
    H = host.NativeHost(*paths)
    hdrs = H.tile(region, o)
    want_hdrs, pb = _python_batch(*paths, region, o)
    assert hdrs == want_hdrs
    b, idx = H.batch(0, len(hdrs), o)
    _same(b, pb, bool(o.linked))
    assert [hdrs[i] for i in idx] == b.hdr
    assert (H.sample(False), H.sample(True)) == ("NORMAL", "TUMOR")
    # the same windows in two chunks (how lancet_gpu walks a region): the kept windows concatenate to the whole
    mid = len(hdrs) // 2
    b1, i1 = H.batch(0, mid, o)
    b2, i2 = H.batch(mid, len(hdrs), o)
    assert i1 + i2 == idx and b1.n_reads + b2.n_reads == b.n_reads
    H.close()


_decorate = read_variety.decorate


@pytest.mark.parametrize("seed,linked", [(0, False), (1, True), (2, False)])
def test_native_host_on_randomized_bams_with_filtered_reads(tmp_path, seed, linked):
    """Synthetic tumor/normal with XT/XA/XS tags, duplicates, secondary alignments, low MAPQ, soft clips and unmapped
    flags sprinkled in, written as BAM by the test-side writer: every filter of extractReads / isActiveRegion is hit."""
    rng = np.random.default_rng(400 + seed)
    data = synth.make_tumor_normal(ref_len=6000, cov_t=25, cov_n=20, ref_seed=30 + seed, tumor_seed=130 + seed, normal_seed=230 + seed,
                                   somatic_every=500, germline_every=400, str_fraction=0.05 if seed == 2 else 0.0)
    tumor = _decorate(synth.pairs_to_sorted_reads(data["tumor"]), rng, linked)
    normal = _decorate(synth.pairs_to_sorted_reads(data["normal"]), rng, linked)
    refs = [("chrA", 1000), (data["rname"], len(data["ref"]))]
    tb, nb, fa = str(tmp_path / "t.bam"), str(tmp_path / "n.bam"), str(tmp_path / "r.fa")
    bam_writer.write_bam(tb, refs, tumor, sample="TT")
    bam_writer.write_bam(nb, refs, normal, sample="NN")
    synth.write_fasta(fa, "chrA", "ACGT" * 250)
    with open(fa, "a") as fh:
        fh.write(f">{data['rname']} some description\n")
        s = data["ref"][:3000].lower() + "R" + data["ref"][3001:]          # soft-masked half + an IUPAC code
        for i in range(0, len(s), 70):
            fh.write(s[i:i + 70] + "\n")
    # the writer and the Python reader agree (so the comparison below is about the native reader)
    _, back = bamio.read_bam(tb)
    assert [(r.qname, r.flag, r.pos, r.cigar, r.seq, r.qual) for r in back] == [(r.qname, r.flag, r.pos, r.cigar, r.seq, r.qual) for r in tumor]
    for kw in ({"linked": int(linked)}, {"linked": int(linked), "active_region": 0, "primary_alignment_only": 1, "xa_filter": 1, "min_map_qual": 10},
               {"linked": int(linked), "max_avg_cov": 20, "active_region": 0}):
        o = host.default_opts(**kw)
        region = f"{data['rname']}:700-5200"
        H = host.NativeHost(tb, nb, fa)
        hdrs = H.tile(region, o)
        want_hdrs, pb = _python_batch(tb, nb, fa, region, o)
        assert hdrs == want_hdrs and len(hdrs) > 30
        b, idx = H.batch(0, len(hdrs), o)
        _same(b, pb, linked)
        assert (H.sample(False), H.sample(True)) == ("NN", "TT")
        H.close()
    assert 0 < pb.n_windows < len(hdrs)                     # (--max-avg-cov 20 skipped some windows, kept others)


def test_native_host_reports_bad_inputs(tmp_path):
    with pytest.raises(Exception):
        host.NativeHost(str(tmp_path / "missing.bam"), str(tmp_path / "missing.bam"), os.path.join(G, "ar_small.fa"))
    H = host.NativeHost(os.path.join(G, "ar_small.tumor.bam"), os.path.join(G, "ar_small.normal.bam"), os.path.join(G, "ar_small.fa"))
    with pytest.raises(Exception):
        H.tile("chrZ:1-100", host.default_opts())
    notbam = tmp_path / "x.bam"
    notbam.write_bytes(b"hello")
    H2 = host.NativeHost(str(notbam), str(notbam), os.path.join(G, "ar_small.fa"))
    with pytest.raises(Exception):
        H2.tile("chr22:900-3000", host.default_opts())


def test_lancet_gpu_binary_has_no_cpu_path():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, "ar_small.tumor.bam"), "--normal", os.path.join(G, "ar_small.normal.bam"),
                        "--ref", os.path.join(G, "ar_small.fa"), "--reg", "chr22:900-3000"], capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout == "" and "cannot create the MI355X engine" in r.stderr


def _body(text):
    return "".join(l + "\n" for l in text.splitlines()
                   if not l.startswith("##fileDate") and not l.startswith("##cmdline") and not l.startswith("##reference"))


@pytest.mark.gpu
@pytest.mark.parametrize("case,region,extra", [("ar_small", "chr22:900-3000", ["--num-threads", "1"]),
                                               ("lr_small", "chr22:800-2700", ["--linked-reads"]),
                                               ("ar_small", "chr22:900-3000", ["--batch-windows", "7"]),
                                               ("lr_small", "chr22:800-2700", ["--linked-reads", "--batch-windows", "5"]),
                                               ("ar_small", "chr22:900-3000", ["--batch-windows", "3", "--devices", "0,0"]),
                                               ("lr_small", "chr22:800-2700", ["--linked-reads", "--batch-windows", "4", "--devices", "0,0,0"])])
def test_lancet_gpu_binary_vcf_is_byte_identical_to_the_reference(case, region, extra):
    """The native command-line program end to end (BAM -> host front end -> engine -> VariantDB -> VCF) on the
    reference-made fixtures, also with the region cut into several engine batches and with the batches going round several
    engines (here: on the same GPU)."""
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, f"{case}.tumor.bam"), "--normal", os.path.join(G, f"{case}.normal.bam"),
                        "--ref", os.path.join(G, f"{case}.fa"), "--reg", region, "--date-line", "Sun Sep 27 05:27:00 2026"] + extra,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "##fileDate=Sun Sep 27 05:27:00 2026\n##source=lancet 1.1.0" in r.stdout and "##cmdline=lancet --tumor" in r.stdout
    assert "--date-line" not in r.stdout
    assert _body(r.stdout) == gu.golden_vcf(case)


def test_native_host_multi_contig_inputs_and_whole_contig_region(tmp_path):
    """Two contigs in BAM and FASTA (reads of the other contig must not leak into the windows), region given as a bare
    contig name (loadRefs takes the whole contig, no padding), reads touching the contig's first and last bases."""
    rng = np.random.default_rng(77)
    a = synth.make_tumor_normal(ref_len=2600, cov_t=18, cov_n=14, ref_seed=41, tumor_seed=141, normal_seed=241, somatic_every=600, germline_every=500)
    b = synth.make_tumor_normal(ref_len=2200, cov_t=18, cov_n=14, ref_seed=42, tumor_seed=142, normal_seed=242, somatic_every=500, germline_every=400)
    def rename(reads, name):
        return [synth.SamRead(r.qname, r.flag, name, r.pos, r.mapq, r.cigar, r.seq, r.qual, r.tags) for r in reads]
    refs = [("chrA", len(a["ref"])), ("chrB", len(b["ref"]))]
    tum = rename(synth.pairs_to_sorted_reads(a["tumor"]), "chrA") + rename(synth.pairs_to_sorted_reads(b["tumor"]), "chrB")
    nor = rename(synth.pairs_to_sorted_reads(a["normal"]), "chrA") + rename(synth.pairs_to_sorted_reads(b["normal"]), "chrB")
    tb, nb, fa = str(tmp_path / "t.bam"), str(tmp_path / "n.bam"), str(tmp_path / "r.fa")
    bam_writer.write_bam(tb, refs, tum, sample="T2")
    bam_writer.write_bam(nb, refs, nor, sample="N2")
    with open(fa, "w") as fh:
        for name, seq in (("chrA", a["ref"]), ("chrB", b["ref"])):
            fh.write(f">{name}\n")
            for i in range(0, len(seq), 61):
                fh.write(seq[i:i + 61] + "\n")
    for region in ("chrB", "chrA:1-2600", "chrB:300-1900"):
        o = host.default_opts(active_region=0)
        H = host.NativeHost(tb, nb, fa)
        hdrs = H.tile(region, o)
        want_hdrs, pb = _python_batch(tb, nb, fa, region, o)
        assert hdrs == want_hdrs
        bt, idx = H.batch(0, len(hdrs), o)
        _same(bt, pb, False)
        assert bt.n_reads > 100 and all(h.startswith(region.split(":")[0] + ":") for h in bt.hdr)
        H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,region,extra", [("ar_small", "chr22:900-3000", ["--batch-windows", "7"]),
                                               ("lr_small", "chr22:800-2700", ["--linked-reads", "--devices", "0,0", "--batch-windows", "6"])])
def test_lancet_gpu_verbose_trace_equals_the_reference_trace(case, region, extra):
    """`lancet_gpu -v`: the per-window stage trace on stderr (native formatter, windows numbered across engine batches)
    against the trace of the reference's own -v run on the same BAMs."""
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, f"{case}.tumor.bam"), "--normal", os.path.join(G, f"{case}.normal.bam"),
                        "--ref", os.path.join(G, f"{case}.fa"), "--reg", region, "-v"] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert gu.digest_trace(r.stderr) == gu.digest_trace(gu.golden_trace(case))
    assert _body(r.stdout) == gu.golden_vcf(case)


FLT_CASES = {   # name: (region, host options, reference flags, linked)
    "flt_small": ("chr22:900-3300", dict(xa_filter=1, primary_alignment_only=1, min_map_qual=20),
                  ["--XA-tag-filter", "--primary-alignment-only", "--min-map-qual", "20"], False),
    "lrflt_small": ("chr22:800-2900", dict(primary_alignment_only=1, linked=1), ["--linked-reads", "--primary-alignment-only"], True),
}


@pytest.mark.parametrize("case", sorted(FLT_CASES))
def test_read_filters_against_a_reference_run_with_the_filters_engaged(case):
    """`flt_small` / `lrflt_small` (tools/make_filter_golden.py): the reference itself on reads with XT:A:R / XA tags, AS-XS
    ties, duplicates, secondary alignments, low MAPQ, soft clips and unmapped flags (and, linked, BX / HP tags on most but
    not all reads), run with its read-filter options and active regions on.  The native host side must assemble exactly
    the reference's windows with exactly its read counts (the reference's -v prints both per window); the oracle and the
    emulated kernels on that batch must reproduce the reference's VCF and stage trace."""
    import re
    import sys
    from oracle import oracle
    from lancet_amd import abi, engine
    region, opts, _, linked = FLT_CASES[case]
    paths = [os.path.join(G, f"{case}.tumor.bam"), os.path.join(G, f"{case}.normal.bam"), os.path.join(G, f"{case}.fa")]
    o = host.default_opts(**opts)
    H = host.NativeHost(*paths)
    hdrs = H.tile(region, o)
    b, idx = H.batch(0, len(hdrs), o)
    ref_trace = gu.golden_trace(case)
    want = [(m.group(1), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"== Processing \d+: (\S+) numsequences: (\d+) mapped: (\d+)", ref_trace)]
    nr = np.diff(b.read_begin.astype(np.int64))
    got = [(b.hdr[w], int(nr[w]), int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())) for w in range(b.n_windows)]
    assert len(want) > 15 and got == want and len(hdrs) >= len(want)
    assert any(m < n for _, n, m in want)                       # (unmapped-flagged reads are loaded and counted as such)
    want_hdrs, pb = _python_batch(*paths, region, o)
    _same(b, pb, linked)
    p = abi.default_params(lr_mode=int(linked))
    ov, ost, otr = oracle.run(b, p, verbose=True)
    db = engine.VariantDB()
    db.add_records(ov, ["chr22"], bx_names=b.bx_names if linked else None)
    assert db.vcf(sample_normal="NORMAL", sample_tumor="TUMOR") == gu.golden_vcf(case)
    assert gu.digest_trace(otr) == gu.digest_trace(ref_trace)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu                                                   # the kernel source under the wave emulator, same batch
    ev, est, etr = emu.run(b, p, evt_cap=1 << 17)
    assert ev == ov and gu.digest_trace(etr) == gu.digest_trace(ref_trace)
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(FLT_CASES))
def test_lancet_gpu_with_the_read_filters_engaged_equals_the_reference(case):
    region, _, args, _ = FLT_CASES[case]
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, f"{case}.tumor.bam"), "--normal", os.path.join(G, f"{case}.normal.bam"),
                        "--ref", os.path.join(G, f"{case}.fa"), "--reg", region, "-v", "--batch-windows", "9"] + args,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _body(r.stdout) == gu.golden_vcf(case)
    assert gu.digest_trace(r.stderr) == gu.digest_trace(gu.golden_trace(case))


# reference option -> lancet_host_opts field as the command-line programs map them (lancet_main.cc / cli.py); --max-as-xs-diff
# maps to nothing: the reference's main() parses it and never hands it to the assemblers (src/Lancet.cc:865-918 vs :496)
_OPT_MAP = {"--max-avg-cov": ("max_avg_cov", int), "--min-map-qual": ("min_map_qual", int), "--window-size": ("window_size", int),
            "--padding": ("padding", int), "--min-alt-count-tumor": ("min_evidence", int), "--min-base-qual": ("min_qual_call", lambda v: int(v) + 33),
            "--max-as-xs-diff": (None, int), "--quality-range": (None, str)}
_OPT_FLAGS = {"--active-region-off": ("active_region", 0), "--XA-tag-filter": ("xa_filter", 1), "--primary-alignment-only": ("primary_alignment_only", 1)}


def test_window_and_read_selection_under_the_reference_s_options():
    """tests/golden/flt_small.options.txt (tools/make_option_goldens.py): for fifteen option sets, the windows the reference
    assembled and its read counts per window, from its own -v.  The native host side, given the same options, must select
    the same windows with the same reads -- tiling (--window-size, --padding), coverage cut-off, MAPQ, active-region
    thresholds (--min-alt-count-tumor, --min-base-qual), XA / primary-alignment filters; and --max-as-xs-diff changes
    nothing, as in the reference."""
    import json
    spec = json.load(open(os.path.join(G, "flt_small.options.txt")))
    paths = [os.path.join(G, "flt_small.tumor.bam"), os.path.join(G, "flt_small.normal.bam"), os.path.join(G, "flt_small.fa")]
    assert len(spec["option_sets"]) >= 15
    for optstr, want in spec["option_sets"].items():
        toks, kw, i = optstr.split(), {}, 0
        while i < len(toks):
            if toks[i] in _OPT_FLAGS:
                kw[_OPT_FLAGS[toks[i]][0]] = _OPT_FLAGS[toks[i]][1]; i += 1
            else:
                field, conv = _OPT_MAP[toks[i]]
                if field:
                    kw[field] = conv(toks[i + 1])
                i += 2
        o = host.default_opts(**kw)
        H = host.NativeHost(*paths)
        hdrs = H.tile(spec["region"], o)
        b, idx = H.batch(0, len(hdrs), o)
        nr = np.diff(b.read_begin.astype(np.int64))
        got = [f"{b.hdr[w]} {int(nr[w])} {int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())}" for w in range(b.n_windows)]
        assert got == want, optstr
        H.close()
    sets = spec["option_sets"]
    assert sets["--max-as-xs-diff 2"] == sets[""] == sets["--max-as-xs-diff 9"] and sets["--min-map-qual 5"] != sets[""]


def test_reads_of_a_window_without_mapped_reads_stay_in_the_graph_as_in_the_reference():
    """`leak_small` (tools/make_filter_golden.py): every read starting in [1500, 2250) carries the unmapped flag, so three
    consecutive windows hold no mapped read.  The reference's processGraph returns there before g.clear()
    (src/Microassembler.cc:83): the windows print nothing but use up a number, and their reads are still in
    the graph when the next window is loaded (631 reads in `chr22:1850-2450`).  Both host sides reproduce it; oracle and
    emulated kernels on that batch give the reference's VCF and trace."""
    import re
    import sys
    from oracle import oracle
    from lancet_amd import abi, engine
    paths = [os.path.join(G, "leak_small.tumor.bam"), os.path.join(G, "leak_small.normal.bam"), os.path.join(G, "leak_small.fa")]
    o = host.default_opts(active_region=0)
    H = host.NativeHost(*paths)
    hdrs = H.tile("chr22:1000-3000", o)
    b, idx = H.batch(0, len(hdrs), o)
    ref_trace = gu.golden_trace("leak_small")
    want = [(int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))) for m in
            re.finditer(r"== Processing (\d+): (\S+) numsequences: (\d+) mapped: (\d+)", ref_trace)]
    nr = np.diff(b.read_begin.astype(np.int64))
    got = [(w + 1, b.hdr[w], int(nr[w]), int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())) for w in range(b.n_windows)]
    assert [g for g in got if g[3] > 0] == want and len(got) == len(want) + 3
    assert [g[0] for g in got if g[3] == 0] == [6, 7, 8] and (9, "chr22:1850-2450", 631, 20) in want
    # the same scan in chunks of four windows: the reads are carried from one call to the next
    parts = [H.batch(lo, min(lo + 4, len(hdrs)), o)[0] for lo in range(0, len(hdrs), 4)]
    assert sum(x.n_reads for x in parts) == b.n_reads and [int(n) for x in parts for n in np.diff(x.read_begin.astype(np.int64))] == [g[2] for g in got]
    want_hdrs, pb = _python_batch(*paths, "chr22:1000-3000", o)
    _same(b, pb, False)
    p = abi.default_params()
    ov, ost, otr = oracle.run(b, p, verbose=True)
    db = engine.VariantDB()
    db.add_records(ov, ["chr22"])
    assert db.vcf(sample_normal="NORMAL", sample_tumor="TUMOR") == gu.golden_vcf("leak_small")
    assert gu.digest_trace(otr) == gu.digest_trace(ref_trace)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu
    ev, est, etr = emu.run(b, p, evt_cap=1 << 18)
    assert ev == ov and gu.digest_trace(etr) == gu.digest_trace(ref_trace) and [s["status"] for s in est][5:8] == [1, 1, 1]
    H.close()


@pytest.mark.gpu
def test_lancet_gpu_reproduces_the_reference_s_read_leak():
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, "leak_small.tumor.bam"), "--normal", os.path.join(G, "leak_small.normal.bam"),
                        "--ref", os.path.join(G, "leak_small.fa"), "--reg", "chr22:1000-3000", "--active-region-off", "-v", "--batch-windows", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _body(r.stdout) == gu.golden_vcf("leak_small")
    assert gu.digest_trace(r.stderr) == gu.digest_trace(gu.golden_trace("leak_small"))
