"""The native host side (include/lancet_host.h: BAM / FASTA input, tiling, window filters, read selection, batch
assembly) against the Python front end it replaces (lancet_amd/frontend.py + bamio.py, themselves pinned on
reference-made fixtures by test_cli.py / test_frontend*.py): identical batches array for array.  CPU only."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import bam_writer
import read_variety
import golden_util as gu
from lancet_amd import bamio, build, frontend, host, synth

G = gu.GOLDEN
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def test_read_group_filter_against_a_reference_run_with_rg_file():
    """`rg_small` (tools/make_filter_golden.py): the reference with --rg-file naming two of three read groups, on reads that carry
    RG:Z:rgA / rgB / rgC or no RG tag, active regions on (loadRG and its two uses, reference src/Microassembler.cc:29-48, :296-302,
    :611-616).  The native host side selects exactly the reference's windows and read counts; the oracle on that batch reproduces
    the reference's VCF and stage trace; without the file the selection differs (the filter is not a no-op on this input)."""
    import re
    from oracle import oracle
    from lancet_amd import abi, engine
    case, region = "rg_small", "chr22:900-3100"
    paths = [os.path.join(G, f"{case}.tumor.bam"), os.path.join(G, f"{case}.normal.bam"), os.path.join(G, f"{case}.fa")]
    o = host.default_opts()
    ref_trace = gu.golden_trace(case)
    want = [(m.group(1), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"== Processing \d+: (\S+) numsequences: (\d+) mapped: (\d+)", ref_trace)]
    got = {}
    for rg in (os.path.join(G, f"{case}.rg.txt"), None):
        H = host.NativeHost(*paths)
        H.set_rg_file(rg)
        hdrs = H.tile(region, o)
        b, idx = H.batch(0, len(hdrs), o)
        nr = np.diff(b.read_begin.astype(np.int64))
        got[rg] = [(b.hdr[w], int(nr[w]), int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())) for w in range(b.n_windows)]
        if rg:
            assert len(want) > 15 and got[rg] == want
            p = abi.default_params()
            ov, ost, otr = oracle.run(b, p, verbose=True)
            db = engine.VariantDB()
            db.add_records(ov, ["chr22"])
            assert db.vcf(sample_normal="NORMAL", sample_tumor="TUMOR") == gu.golden_vcf(case)
            assert gu.digest_trace(otr) == gu.digest_trace(ref_trace)
        H.close()
    assert got[None] != want and sum(n for _, n, _ in got[None]) > 1.3 * sum(n for _, n, _ in want)


@pytest.mark.gpu
def test_lancet_gpu_with_rg_file_equals_the_reference():
    case = "rg_small"
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, f"{case}.tumor.bam"), "--normal", os.path.join(G, f"{case}.normal.bam"),
                        "--ref", os.path.join(G, f"{case}.fa"), "--reg", "chr22:900-3100", "-v", "--rg-file", os.path.join(G, f"{case}.rg.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _body(r.stdout) == gu.golden_vcf(case)
    assert gu.digest_trace(r.stderr) == gu.digest_trace(gu.golden_trace(case))


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


def test_batch_calls_in_any_order_carry_the_read_leak_of_the_windows_before_them(tmp_path, monkeypatch):
    """An N-process run deals the window table out (lancet_gpu --ranks): a batch call that does not start where the previous one ended
    has to work out again what the windows before it left in the graph (lancet_host.h; reference src/Microassembler.cc:83).  On the
    leak fixture -- windows 6, 7, 8 hold no mapped read, window 9 inherits their reads: 631 in all -- every window asked for on its own, last
    window first, equals that window of the sequential scan; so do runs of three windows starting inside the leaking run, with and without
    the per-call loading of lazy mode (an indexed copy of the fixture), and with lancet_host_load_range in front."""
    from lancet_amd import bamio
    fa = os.path.join(G, "leak_small.fa")
    o = host.default_opts(active_region=0)

    def baseline(paths):
        H = host.NativeHost(*paths)
        n = len(H.tile("chr22:1000-3000", o))
        seq = [H.batch(w, w + 1, o)[0] for w in range(n)]
        H.close()
        assert int(np.diff(seq[8].read_begin)[0]) == 631 and [b.n_windows for b in seq[5:8]] == [1, 1, 1]
        return n, seq

    def same(a, b):
        assert a.n_windows == b.n_windows
        for f in FIELDS:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f

    def check(paths, n, seq):
        H2 = host.NativeHost(*paths)
        assert len(H2.tile("chr22:1000-3000", o)) == n
        for w in reversed(range(n)):
            same(H2.batch(w, w + 1, o)[0], seq[w])
        for lo in (6, 7, 8, 5):                                   # runs that start inside / just before the leaking run
            b3 = H2.batch(lo, min(n, lo + 3), o)[0]
            assert [int(x) for x in np.diff(b3.read_begin)] == [int(np.diff(seq[w].read_begin)[0]) for w in range(lo, min(n, lo + 3)) if seq[w].n_windows]
        H2.L.lancet_host_load_range.argtypes = [host.C.c_void_p, host.C.c_int, host.C.c_int, host.C.POINTER(host.LancetHostOpts)]
        assert H2.L.lancet_host_load_range(H2.h, 7, n, host.C.byref(o)) == 0
        for w in range(7, n):
            same(H2.batch(w, w + 1, o)[0], seq[w])
        H2.close()

    paths = [os.path.join(G, "leak_small.tumor.bam"), os.path.join(G, "leak_small.normal.bam"), fa]
    n, seq = baseline(paths)
    check(paths, n, seq)
    # lazy mode needs indexes: copies of the fixture written with the test-side BAM writer (+ .bai)
    lp = []
    for nm, sample in (("tumor", "TUMOR"), ("normal", "NORMAL")):
        hdr, reads = bamio.read_bam(os.path.join(G, f"leak_small.{nm}.bam"))
        out = str(tmp_path / (nm + ".bam"))
        bam_writer.write_bam(out, hdr["refs"], reads, sample=sample, index=True)
        lp.append(out)
    monkeypatch.setenv("LANCET_HOST_LAZY", "0")
    n2, seq2 = baseline(lp + [fa])
    monkeypatch.setenv("LANCET_HOST_LAZY", "1")
    check(lp + [fa], n2, seq2)


@pytest.mark.gpu
def test_lancet_gpu_reproduces_the_reference_s_read_leak():
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, "leak_small.tumor.bam"), "--normal", os.path.join(G, "leak_small.normal.bam"),
                        "--ref", os.path.join(G, "leak_small.fa"), "--reg", "chr22:1000-3000", "--active-region-off", "-v", "--batch-windows", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _body(r.stdout) == gu.golden_vcf("leak_small")
    assert gu.digest_trace(r.stderr) == gu.digest_trace(gu.golden_trace("leak_small"))


# ---------------------------------------------------------------------------------------------------------------------
# input at scale: .bai / streaming region reads, several contigs, --bed

def _bed2_paths():
    return [os.path.join(G, "bed2.tumor.bam"), os.path.join(G, "bed2.normal.bam"), os.path.join(G, "bed2.fa")]


def _trace_windows(case):
    import re
    out = []
    for line in open(os.path.join(G, f"{case}.trace.txt")):
        m = re.match(r"== Processing (\d+): (\S+) numsequences: (\d+) mapped: (\d+)", line)
        if m:
            out.append((m.group(2), int(m.group(3))))
    return out


def _batches_equal(a, b, linked=False):
    assert a.n_windows == b.n_windows and a.hdr == b.hdr and a.chrom == b.chrom
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_bed_plus_region_tiling_and_selection_equal_the_reference_run(tmp_path, capfd):
    """tests/golden/bed2.*: the reference itself on a BED file (three lines on two contigs, two of them overlapping once padded)
    together with --reg.  Its -v trace names every window it assembled, in its processing order, with the number of reads:
    the native host must tile (double padding of BED intervals, one table ordered by header, duplicates once), filter (active
    regions on) and select the same.  Read through the bamtools-made .bai (no linear index), through a .bai with a linear
    index, and streamed without an index: identical batches; and equal to the Python front end."""
    paths = _bed2_paths()
    case = __import__("json").load(open(os.path.join(G, "bed2.case.txt")))
    bed = os.path.join(G, "bed2.bed")
    want = _trace_windows("bed2")
    assert len(want) == 32 and {h.split(":")[0] for h, _ in want} == {"chr21", "chr22"}
    o = host.default_opts()
    os.environ["LANCET_HOST_TIMING"] = "1"
    try:
        H = host.NativeHost(*paths)
        hdrs = H.tile_regions([case["region"]], o, bed=bed)
        err = capfd.readouterr().err
        assert err.count("indexed (.bai)") == 2
        assert H.chroms() == ["chr22", "chr21"] and (H.first_has_md(True), H.first_has_md(False)) == (True, True)
        assert hdrs == sorted(set(hdrs)) and len(hdrs) >= len(want)
        b, idx = H.batch(0, len(hdrs), o)
        H.close()
        assert [(h, int(b.read_begin[i + 1] - b.read_begin[i])) for i, h in enumerate(b.hdr)] == want
        assert [b.chrom[i] for i in range(b.n_windows)] == [h.split(":")[0] for h in b.hdr]
        assert [H2 for H2 in np.asarray(b.chr_id)] == [["chr22", "chr21"].index(c) for c in b.chrom]
        # no index: streamed
        import shutil
        for f in ("bed2.tumor.bam", "bed2.normal.bam", "bed2.fa"):
            shutil.copy(os.path.join(G, f), tmp_path / f)
        sp = [str(tmp_path / "bed2.tumor.bam"), str(tmp_path / "bed2.normal.bam"), str(tmp_path / "bed2.fa")]
        H = host.NativeHost(*sp)
        assert H.tile_regions([case["region"]], o, bed=bed) == hdrs
        assert capfd.readouterr().err.count("no .bai: streamed") == 2
        b2, _ = H.batch(0, len(hdrs), o)
        H.close()
        _batches_equal(b, b2)
        # the same BAM content rewritten by the test-side writer with a samtools-style .bai (linear index present)
        for smp in ("tumor", "normal"):
            _, reads = bamio.read_bam(os.path.join(G, f"bed2.{smp}.bam"))
            bam_writer.write_bam(str(tmp_path / f"lin.{smp}.bam"), [("chr21", 3600), ("chr22", 4000)], reads, sample=smp.upper(), index=True, block=9000)
        H = host.NativeHost(str(tmp_path / "lin.tumor.bam"), str(tmp_path / "lin.normal.bam"), sp[2])
        assert H.tile_regions([case["region"]], o, bed=bed) == hdrs
        assert capfd.readouterr().err.count("indexed (.bai)") == 2
        b3, _ = H.batch(0, len(hdrs), o)
        H.close()
        _batches_equal(b, b3)
    finally:
        del os.environ["LANCET_HOST_TIMING"]
    # Python front end on the same inputs, contigs side by side
    contigs = bamio.read_fasta(paths[2])
    by = {}
    for smp, pth in (("tumor", paths[0]), ("normal", paths[1])):
        _, reads = bamio.read_bam(pth)
        by[smp] = {c: [r for r in reads if r.rname == c] for c in ("chr21", "chr22")}
    wins = []
    for line in open(bed):
        if line.startswith("#"):
            continue
        c, s, e = line.rstrip("\n").split("\t")[:3]
        reg = f"{c}:{max(1, int(s) - o.padding)}-{int(e) + o.padding}"
        wins += frontend.tile_region(contigs[c], c, reg, padding=o.padding, window_size=o.window_size)
    c = case["region"].split(":")[0]
    wins += frontend.tile_region(contigs[c], c, case["region"], padding=o.padding, window_size=o.window_size)
    seen, uniq = set(), []
    for w in wins:
        if w.hdr not in seen:
            seen.add(w.hdr); uniq.append(w)
    fp = frontend.ReadFilterParams(min_map_qual=o.min_map_qual, max_delta_as_xs=o.max_delta_as_xs, max_avg_cov=o.max_avg_cov)
    pb, _ = frontend.batch_from_sam(uniq, by["tumor"], by["normal"], fp, max_k=o.max_k, active_region=True, min_evidence=o.min_evidence,
                                    min_qual_call=o.min_qual_call)
    assert pb.hdr == b.hdr
    for f in FIELDS:
        if f != "chr_id":                                   # (the two number the contigs in their own order of first appearance)
            assert np.array_equal(getattr(b, f), getattr(pb, f)), f


def test_indexed_reads_of_far_apart_stretches_seek_instead_of_streaming(tmp_path, capfd):
    """Two contigs of 60 kb at low coverage, three small stretches far apart (one on the second contig): with a .bai the reader
    seeks to each stretch and inflates a fraction of the file; the batches equal those of the streamed read."""
    rng = np.random.default_rng(5)
    datas = [synth.make_tumor_normal(ref_len=60000, cov_t=5, cov_n=4, ref_seed=501 + i, tumor_seed=1501 + i, normal_seed=2501 + i,
                                     somatic_every=900, germline_every=700) for i in range(2)]
    names = ["ctgA", "ctgB"]
    ren = lambda rs, name: [synth.SamRead(r.qname + name, r.flag, name, r.pos, r.mapq, r.cigar, r.seq, r.qual, r.tags) for r in rs]
    refs = [(n, len(d["ref"])) for n, d in zip(names, datas)]
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as fh:
        for n, d in zip(names, datas):
            fh.write(f">{n}\n")
            for i in range(0, len(d["ref"]), 80):
                fh.write(d["ref"][i:i + 80] + "\n")
    for smp in ("tumor", "normal"):
        reads = sum((ren(synth.pairs_to_sorted_reads(d[smp]), n) for n, d in zip(names, datas)), [])
        for tag, kw in (("idx", dict(index=True)), ("lin0", dict(index=True, linear=False)), ("raw", dict())):
            bam_writer.write_bam(str(tmp_path / f"{tag}.{smp}.bam"), refs, reads, sample=smp, block=20000, **kw)
    regions = ["ctgA:20100-21000", "ctgA:52000-52900", "ctgB:40500-41400"]
    o = host.default_opts(active_region=0)
    os.environ["LANCET_HOST_TIMING"] = "1"
    os.environ["LANCET_HOST_SLAB_KB"] = "64"            # (so that this small file spans many slabs)
    try:
        got = {}
        for tag in ("idx", "lin0", "raw"):
            H = host.NativeHost(str(tmp_path / f"{tag}.tumor.bam"), str(tmp_path / f"{tag}.normal.bam"), fa)
            hdrs = H.tile_regions(regions, o)
            err = capfd.readouterr().err
            import re
            mb = [int(x) for x in re.findall(r"(\d+) MB inflated", err)]
            seeks = [int(x) for x in re.findall(r"(\d+) seeks", err)]
            b, _ = H.batch(0, len(hdrs), o)
            got[tag] = (b, seeks, mb)
            H.close()
    finally:
        del os.environ["LANCET_HOST_TIMING"], os.environ["LANCET_HOST_SLAB_KB"]
    _batches_equal(got["idx"][0], got["raw"][0]); _batches_equal(got["lin0"][0], got["raw"][0])
    assert got["raw"][0].n_reads > 100 and got["raw"][0].n_windows >= 20
    assert all(s >= 2 for s in got["idx"][1]) and all(s >= 2 for s in got["lin0"][1]) and got["raw"][1] == [0, 0]


def test_native_bam_reader_rejects_damaged_files_instead_of_reading_past_them(tmp_path):
    """Truncated file, a header length pointing past the end, a record with field lengths past its end, unsorted records:
    an error message, not a crash (every length is checked against what is there)."""
    import struct
    import zlib
    fa = os.path.join(G, "ar_small.fa")
    good = open(os.path.join(G, "ar_small.tumor.bam"), "rb").read()
    o = host.default_opts()

    def tile_fails(raw, what):
        p = tmp_path / "x.bam"
        p.write_bytes(raw)
        H = host.NativeHost(str(p), str(p), fa)
        with pytest.raises(Exception) as ei:
            H.tile("chr22:900-3000", o)
        assert what in str(ei.value), str(ei.value)
        H.close()
    tile_fails(good[:len(good) // 2], "BGZF")                                        # cut inside a block
    inflated = bamio.bgzf_decompress(good)
    blocks = lambda body: b"".join(bam_writer._bgzf_block(body[i:i + 60000]) for i in range(0, len(body), 60000)) + bam_writer._bgzf_block(b"")
    tile_fails(blocks(inflated[:len(inflated) // 2 + 7]), "truncated alignment record")   # cut inside a record
    bad = bytearray(inflated); bad[4:8] = struct.pack("<i", 0x7FFFFF00)
    tile_fails(blocks(bytes(bad)), "truncated BAM header")
    # first record: l_seq far larger than the record
    l_text = struct.unpack_from("<i", inflated, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", inflated, p)[0]; p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", inflated, p)[0]; p += 8 + ln
    bad = bytearray(inflated); bad[p + 4 + 16:p + 4 + 20] = struct.pack("<i", 1 << 24)
    # (the damaged record must be one the region selects: move it into the region)
    bad[p + 4 + 4:p + 4 + 8] = struct.pack("<i", 1000)
    tile_fails(blocks(bytes(bad)), "fields past its end")
    # records out of coordinate order
    _, reads = bamio.read_bam(os.path.join(G, "ar_small.tumor.bam"))
    sw = list(reads); sw[40], sw[300] = sw[300], sw[40]
    bam_writer.write_bam(str(tmp_path / "u.bam"), [("chr22", 4000)], sw)
    tile_fails(open(tmp_path / "u.bam", "rb").read(), "not coordinate sorted")


def test_first_alignment_without_md_is_reported_for_the_active_region_switch(tmp_path):
    """checkPresenceOfMDtag looks at the first alignment of each BAM; main() turns the active-region module off when neither
    has MD (reference src/Lancet.cc:817-825, src/util.cc:416-427)."""
    _, reads = bamio.read_bam(os.path.join(G, "ar_small.tumor.bam"))
    strip = [synth.SamRead(r.qname, r.flag, r.rname, r.pos, r.mapq, r.cigar, r.seq, r.qual, {k: v for k, v in r.tags.items() if k != "MD"}) for r in reads]
    bam_writer.write_bam(str(tmp_path / "nomd.bam"), [("chr22", 4000)], strip)
    H = host.NativeHost(str(tmp_path / "nomd.bam"), os.path.join(G, "ar_small.normal.bam"), os.path.join(G, "ar_small.fa"))
    H.tile("chr22:900-3000", host.default_opts())
    assert (H.first_has_md(True), H.first_has_md(False)) == (False, True)
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("indexed,with_reg", [(True, True), (False, True), (True, False)])
def test_lancet_gpu_bed_and_region_on_two_contigs_is_byte_identical_to_the_reference(tmp_path, indexed, with_reg):
    """`lancet_gpu --bed regions.bed [--reg chr21:...]` on the two-contig fixture (through the .bai, and streamed without one;
    with the BED file alone, i.e. without --reg) against the VCF of the reference's own run with the same arguments."""
    import shutil
    case = __import__("json").load(open(os.path.join(G, "bed2.case.txt")))
    src = {f: os.path.join(G, f) for f in ("bed2.tumor.bam", "bed2.normal.bam", "bed2.fa", "bed2.bed")}
    if not indexed:
        for f in list(src):
            shutil.copy(src[f], tmp_path / f); src[f] = str(tmp_path / f)
    r = subprocess.run([build.BIN, "--tumor", src["bed2.tumor.bam"], "--normal", src["bed2.normal.bam"], "--ref", src["bed2.fa"],
                        "--bed", src["bed2.bed"], "--batch-windows", "9", "--devices", "0,0"] + (["--reg", case["region"]] if with_reg else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _body(r.stdout) == gu.golden_vcf("bed2" if with_reg else "bed2_bedonly")
    assert "chr21\t" in r.stdout and "chr22\t" in r.stdout


@pytest.mark.gpu
def test_lancet_gpu_finishes_and_lists_windows_that_exceeded_the_work_space(tmp_path):
    """A window that overflows the engine's tables contributes nothing and is named on stderr; the run finishes with the
    other windows' variants and exit code 3.  --strict: no VCF at all."""
    args = [build.BIN, "--tumor", os.path.join(G, "ar_small.tumor.bam"), "--normal", os.path.join(G, "ar_small.normal.bam"),
            "--ref", os.path.join(G, "ar_small.fa"), "--reg", "chr22:900-3000"]
    env = dict(os.environ, LANCET_NODE_CAP1="1500", LANCET_MAX_NODES="1500")   # node tables (both tiers) below what a 600-bp window at k = 11 needs
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    assert r.returncode == 3 and "exceeded the engine's work space" in r.stderr and "lancet_gpu:   chr22:" in r.stderr
    assert r.stdout.startswith("##fileformat=VCF")
    r = subprocess.run(args + ["--strict"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and r.stdout == "" and "--strict" in r.stderr


def test_window_repeat_filter_equals_the_set_of_kmers():
    """isRepeat (reference src/util.cc:295-315): a window whose k-mers at offsets [0, len - k) are not all distinct.  The native one uses a
    rolling hash with confirmation on the characters; against a set of substrings on random windows with and without planted repeats,
    several k, short and degenerate inputs."""
    import ctypes as C
    from lancet_amd import engine
    L = engine.lib()
    L.lancet_host_debug_is_repeat.restype = C.c_int
    L.lancet_host_debug_is_repeat.argtypes = [C.c_char_p, C.c_int]
    rng = np.random.default_rng(11)
    def want(s, k):
        n = len(s) - k
        return n > 1 and len({s[i:i + k] for i in range(n)}) < n
    seen = [0, 0]
    for trial in range(600):
        n = int(rng.choice([5, 40, 120, 600, 601, 850]))
        s = "".join("ACGTN"[int(x)] for x in rng.choice(5, size=n, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        k = int(rng.choice([3, 11, 31, 101, 127]))
        if rng.random() < 0.5 and n > 2 * k + 4:            # plant a copy of a k-mer (sometimes at the very end, where the last offset does not count)
            a = int(rng.integers(0, n - k)); b = int(rng.integers(0, n - k + 1))
            s = s[:b] + s[a:a + k] + s[b + k:]
        if rng.random() < 0.1:
            s = s[0] * len(s)                               # one letter: every k-mer the same
        w = want(s, k)
        seen[int(w)] += 1
        assert bool(L.lancet_host_debug_is_repeat(s.encode(), k)) == w, (trial, n, k)
    assert seen[0] > 100 and seen[1] > 100
    assert not L.lancet_host_debug_is_repeat(b"", 11) and not L.lancet_host_debug_is_repeat(b"ACGT", 11)


def test_lazy_per_batch_loading_gives_the_same_batches(tmp_path, monkeypatch, capfd):
    """Lazy mode (LANCET_HOST_LAZY=1, needs a .bai): the tiling only builds the window table, every batch call loads the alignments its
    windows can select.  Batches of 5 windows at a time, concatenated, equal the one batch of the eager mode -- on the two-contig BED
    golden (bamtools-made .bai) and on the read-leak golden rewritten with an index, where reads a window leaves in the graph have to
    survive the reload between two batches."""
    from lancet_amd import workload
    o = host.default_opts()
    # bed2: BED + region on two contigs
    paths = _bed2_paths()
    case = __import__("json").load(open(os.path.join(G, "bed2.case.txt")))
    bed = os.path.join(G, "bed2.bed")
    def all_batches(paths, tile, step, o=o):
        H = host.NativeHost(*paths)
        hdrs = tile(H)
        parts, kept = [], []
        for lo in range(0, len(hdrs), step):
            b, idx = H.batch(lo, min(len(hdrs), lo + step), o)
            if b.n_windows:
                parts.append(b)
            kept += list(idx)
        H.close()
        return hdrs, parts, kept
    tile_bed = lambda H: H.tile_regions([case["region"]], o, bed=bed)
    monkeypatch.setenv("LANCET_HOST_LAZY", "0")
    hdrs, eager, kept = all_batches(paths, tile_bed, 10 ** 6)
    monkeypatch.setenv("LANCET_HOST_LAZY", "1")
    monkeypatch.setenv("LANCET_HOST_TIMING", "1")
    hdrs2, lazy, kept2 = all_batches(paths, tile_bed, 5)
    monkeypatch.delenv("LANCET_HOST_TIMING")
    err = capfd.readouterr().err
    assert err.count("indexed (.bai)") >= 2 * (len(hdrs) // 5) and "0 alignments kept" in err      # (the tiling itself loaded none)
    assert hdrs2 == hdrs and kept2 == kept and len(lazy) > 3
    _batches_equal(eager[0], workload.concat_batches(lazy))
    # leak_small: three windows without a mapped read leave their reads to the next one
    lp = []
    for smp in ("tumor", "normal"):
        hdr, reads = bamio.read_bam(os.path.join(G, f"leak_small.{smp}.bam"))
        out = str(tmp_path / f"leak.{smp}.bam")
        bam_writer.write_bam(out, [("chr22", 4000)], reads, sample=smp.upper(), index=True, block=9000)
        lp.append(out)
    lp.append(os.path.join(G, "leak_small.fa"))
    o2 = host.default_opts(active_region=0)
    tile_reg = lambda H: H.tile("chr22:1000-3000", o2)
    monkeypatch.setenv("LANCET_HOST_LAZY", "0")
    h1, e1, k1 = all_batches(lp, tile_reg, 10 ** 6, o2)
    counts = [int(e1[0].read_begin[i + 1] - e1[0].read_begin[i]) for i in range(e1[0].n_windows)]
    assert 631 in counts                                     # (chr22:1850-2450 with the reads three read-less windows left behind)
    monkeypatch.setenv("LANCET_HOST_LAZY", "1")
    for step in (1, 2, 3):
        h2, l2, k2 = all_batches(lp, tile_reg, step, o2)
        assert h2 == h1 and k2 == k1
        _batches_equal(e1[0], workload.concat_batches(l2))


@pytest.mark.parametrize("shared_form", [True, False])
def test_packed_batch_equals_packing_the_ascii_batch(shared_form, monkeypatch):
    """lancet_host_batch_packed (the host threads trim and pack while they assemble the batch) against lancet_pack_read -- the routine
    lancet_engine_upload runs on the ASCII arrays -- applied read by read to the batch of lancet_host_batch: same window arrays, same trimmed
    lengths and flags, same packed words at the same offsets; with qualities that make the trim bite (--min-base-qual / trim parameters away
    from their defaults) and on the read-filter golden."""
    monkeypatch.setenv("LANCET_HOST_SHARED", "1" if shared_form else "0")
    import ctypes as C
    from lancet_amd import abi, engine
    L = engine.lib()
    L.lancet_pack_read.restype = None
    L.lancet_pack_read.argtypes = [C.POINTER(abi.LancetParams), C.c_char_p, C.c_char_p, C.c_int, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for case, over in (("ar_small", {}), ("tile30", dict(min_qual_trim=33 + 25, min_qual_call=33 + 30)), ("leak_small", dict(min_qual_trim=33 + 12))):
        paths = [os.path.join(G, f"{case}.tumor.bam"), os.path.join(G, f"{case}.normal.bam"), os.path.join(G, f"{case}.fa")]
        if not os.path.exists(paths[0]):
            continue
        p = abi.default_params(**over)
        o = host.default_opts(active_region=0)
        H = host.NativeHost(*paths)
        hdrs = H.tile("chr22:900-3100", o)
        b, idx = H.batch(0, len(hdrs), o)
        b2, idx2, pk = H.batch(0, len(hdrs), o, pack_params=p)
        H.close()
        assert idx2 == idx and b2.hdr == b.hdr and b2.seq.size == 0 and b2.qual.size == 0
        for f in FIELDS:
            if f not in ("seq", "qual"):
                assert np.array_equal(getattr(b, f), getattr(b2, f)), f
        R = b.n_reads
        lens = np.diff(b.seq_off.astype(np.int64))
        # round 5: the reads are stored once per batch -- `read_index` says which distinct read a read of a window is (an alignment lies in
        # several of the overlapping windows); LANCET_HOST_SHARED=0 (second pass of this test) keeps one copy per window
        ridx = pk["read_index"].astype(np.int64) if "read_index" in pk else np.arange(R)
        if shared_form:
            assert "read_index" in pk and pk["n_distinct"] < R and len(pk["rinfo"]) == pk["n_distinct"] + 1, case
        else:
            assert "read_index" not in pk
        assert np.array_equal(np.diff(pk["base_woff"].astype(np.int64))[ridx], (lens + 15) // 16) and np.array_equal(np.diff(pk["good_woff"].astype(np.int64))[ridx], (lens + 31) // 32)
        seq, qual = b.seq.tobytes(), b.qual.tobytes()
        trimmed = 0
        for r in range(0, R, max(1, R // 3000)):                       # (a few thousand reads per case)
            n = int(lens[r]); so = int(b.seq_off[r])
            ri = C.c_uint32(); wb = (C.c_uint32 * ((n + 15) // 16 + 1))(); wg = (C.c_uint32 * ((n + 31) // 32 + 1))()
            L.lancet_pack_read(C.byref(p), seq[so:so + n], qual[so:so + n], n, int(b.label[r]), int(b.strand[r]), int(b.mate[r]), int(b.mapped[r]), C.byref(ri), wb, wg)
            u = int(ridx[r])
            assert ri.value == int(pk["rinfo"][u]), (case, r)
            trimmed += int((ri.value & 0xFFFF) < n)
            a0 = int(pk["base_woff"][u]); g0 = int(pk["good_woff"][u])
            assert list(wb)[: (n + 15) // 16] == pk["bases"][a0:a0 + (n + 15) // 16].tolist(), (case, r)
            assert list(wg)[: (n + 31) // 32] == pk["good"][g0:g0 + (n + 31) // 32].tolist(), (case, r)
        if over:
            assert trimmed > 0, case


@pytest.mark.gpu
def test_packed_upload_gives_the_records_of_the_ascii_upload():
    """lancet_host_batch_packed -> lancet_engine_upload_packed against lancet_host_batch -> lancet_engine_upload on the same windows (and
    the oracle): same records and statistics, with trim parameters that bite; a packed batch whose offsets do not fit the read lengths is
    refused, and so is one packed with other quality thresholds than the engine's."""
    from oracle import oracle
    from lancet_amd import abi, engine
    for case, over in (("ar_small", {}), ("leak_small", dict(min_qual_trim=33 + 25, min_qual_call=33 + 30))):
        paths = [os.path.join(G, f"{case}.tumor.bam"), os.path.join(G, f"{case}.normal.bam"), os.path.join(G, f"{case}.fa")]
        p = abi.default_params(**over)
        o = host.default_opts(active_region=0, min_qual_call=p.min_qual_call)
        H = host.NativeHost(*paths)
        hdrs = H.tile("chr22:900-3100", o)
        b, idx = H.batch(0, len(hdrs), o)
        b2, idx2, pk = H.batch(0, len(hdrs), o, pack_params=p)
        H.close()
        ov, ost, _ = oracle.run(b, p)
        eng = engine.Engine(p)
        v1, s1 = eng.process(b)
        eng.upload_packed(b2, pk); eng.run()
        v2, s2 = eng.results()
        assert v1 == ov and v2 == ov and s2 == s1
        bad = dict(pk); bad["base_woff"] = pk["base_woff"].copy(); bad["base_woff"][1:] += 1
        with pytest.raises(engine.EngineError):
            eng.upload_packed(b2, bad)
        # a producer that does not say which thresholds it packed with is refused (the check below could not see a mismatch otherwise) ...
        with pytest.raises(engine.EngineError, match="min_qual_trim"):
            eng.upload_packed(b2, {k: v for k, v in pk.items() if k != "min_qual_trim"})
        # ... and so is a caller built against another layout of lancet_packed_reads (its size travels in the struct's first field)
        import ctypes as C
        keep = {k: np.ascontiguousarray(pk[k], dtype=np.uint32) for k in ("rinfo", "base_woff", "good_woff", "bases", "good")}
        old = abi.LancetPackedReads(40, 0, *[keep[k].ctypes.data_as(C.POINTER(C.c_uint32)) for k in ("rinfo", "base_woff", "good_woff", "bases", "good")])
        old.min_qual_trim, old.min_qual_call = int(pk["min_qual_trim"]), int(pk["min_qual_call"])
        cb = abi.batch_to_c(b2)
        eng.L.lancet_engine_upload_packed.restype = C.c_int
        eng.L.lancet_engine_upload_packed.argtypes = [C.c_void_p, C.POINTER(abi.LancetWindowBatch), C.POINTER(abi.LancetPackedReads)]
        assert eng.L.lancet_engine_upload_packed(eng.h, C.byref(cb), C.byref(old)) == -1       # LANCET_E_ARG
        assert b"struct_size" in eng.L.lancet_engine_last_error(eng.h)
        eng.close()
        # reads packed for other thresholds are refused, not assembled (the trim and the quality mask depend on them)
        other = engine.Engine(abi.default_params(min_qual_trim=p.min_qual_trim + 3))
        with pytest.raises(engine.EngineError):
            other.upload_packed(b2, pk)
        other.close()


def test_pack_read_against_a_plain_restatement():
    """lancet_pack_read (host_pack.h: what lancet_engine_upload does to every read, several characters per machine word) against
    Graph_t::trim (reference src/Graph.cc:355-384) + the packing spelled out character by character: random reads with N and other
    letters, lower case, low qualities at the ends and inside, quality bytes above 127 (negative characters), lengths around the word
    sizes, several thresholds."""
    import ctypes as C
    from lancet_amd import abi, engine
    L = engine.lib()
    L.lancet_pack_read.restype = None
    L.lancet_pack_read.argtypes = [C.POINTER(abi.LancetParams), C.c_char_p, C.c_char_p, C.c_int, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(3)
    code = {c: i for i, c in enumerate(b"ACGT")}
    code.update({c: i for i, c in enumerate(b"acgt")})
    n_trim = n_junk = 0
    for trial in range(1500):
        n = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 150, 151, 250]))
        qtrim, qcall = int(rng.choice([33 + 10, 33 + 2, 33 + 25])), int(rng.choice([33 + 17, 33 + 30, 34, 127]))
        p = abi.default_params(min_qual_trim=qtrim, min_qual_call=qcall)
        seq = bytearray(rng.choice(list(b"ACGT"), size=n).astype(np.uint8).tobytes())
        qual = bytearray(rng.integers(33 + 20, 33 + 42, size=n, dtype=np.uint8).tobytes())
        for i in range(n):
            u = rng.random()
            if u < 0.01: seq[i] = rng.choice(list(b"NRYn"))
            elif u < 0.05: seq[i] = seq[i] | 0x20
            if rng.random() < 0.06: qual[i] = int(rng.integers(33, 33 + 12))
            if rng.random() < 0.01: qual[i] = int(rng.integers(128, 256))
        k = int(rng.integers(0, 4))
        for i in range(min(k, n)): qual[i] = 33 + 1
        for i in range(min(int(rng.integers(0, 4)), n)): qual[n - 1 - i] = 33 + 1
        sq = lambda b: b - 256 if b >= 128 else b                       # a C `char` on this ABI
        good_end = lambda i: seq[i] in code and not (sq(qual[i]) < qtrim)
        fg = 0
        while fg < n and not good_end(fg): fg += 1
        lg = n - 1
        while lg >= fg and not good_end(lg): lg -= 1
        junk = fg >= n or lg < fg or any(seq[i] not in code for i in range(fg, lg + 1))
        tlen = 0 if junk else lg - fg + 1
        t5 = 0 if junk else fg
        wb = [0] * ((n + 15) // 16); wg = [0] * ((n + 31) // 32)
        for j in range(tlen):
            wb[j // 16] |= code[seq[t5 + j]] << (2 * (j % 16))
            wg[j // 32] |= (1 if sq(qual[t5 + j]) >= qcall else 0) << (j % 32)
        label, strand, mate, mapped = int(rng.choice([4, 5])), int(rng.choice([1, 2])), int(rng.integers(0, 3)), int(rng.integers(0, 2))      # TMR | NML, FWD | REV
        ri = C.c_uint32(); ob = (C.c_uint32 * (len(wb) + 1))(); og = (C.c_uint32 * (len(wg) + 1))()
        L.lancet_pack_read(C.byref(p), bytes(seq), bytes(qual), n, label, strand, mate, mapped, C.byref(ri), ob, og)
        want_ri = tlen | ((1 if label == 5 else 0) << 16) | ((1 if strand == 2 else 0) << 17) | (mate << 18) | (mapped << 20)
        assert ri.value == want_ri, (trial, n, ri.value, want_ri)
        assert list(ob)[:len(wb)] == wb and list(og)[:len(wg)] == wg, (trial, n, tlen)
        n_trim += int(0 < tlen < n); n_junk += int(junk and n > 0)
    assert n_trim > 300 and n_junk > 100


@pytest.mark.gpu
def test_scan_of_a_synthetic_contig_equals_the_oracle_window_by_window_and_in_the_vcf(tmp_path):
    """tools/e2e_parity.py on a 100 kb tumor / normal pair made here (981 windows; the 5 Mb contig of BASELINE config 2 goes through
    the same script on the box: profiles/r6_e2e_parity_5mb.json): `lancet_gpu`'s VCF, the engine's records of the native host side's
    batches and the oracle's records of the same batches (+ oracle/vcf_oracle.py) must agree record by record and byte by byte."""
    import json
    d = str(tmp_path / "scan100k")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_scan_bams.py"), d, "100000", "30", "30", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_parity.py"), d, "chr22:1000-99000", "--batch-windows", "400", "--active-region-off"],
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["records_identical"] and out["vcf_oracle_equals_lancet_gpu"] and out["vcf_engine_records_native_vdb_equals_lancet_gpu"] and out["vcf_header_lines_identical"]
    assert out["windows_assembled"] > 900 and out["vcf_lines"] > 0 and out["records_compared"] >= out["vcf_lines"]


@pytest.mark.gpu
@pytest.mark.parametrize("case,region,extra,files", [
    ("ar_small", "chr22:900-3000", ["--ranks", "1", "--batch-windows", "4"], False),                       # RCCL itself: one rank, the whole pack -> gather -> merge route
    ("lr_small", "chr22:800-2700", ["--linked-reads", "--ranks", "1", "--batch-windows", "5"], False),
    ("ar_small", "chr22:900-3000", ["--ranks", "2", "--devices", "0,0", "--batch-windows", "3"], True),   # two processes on the one GPU: payloads over the test transport
    ("lr_small", "chr22:800-2700", ["--linked-reads", "--ranks", "3", "--devices", "0,0,0", "--batch-windows", "2"], True),
    # the reference's read leak across windows (windows 6-8 without a mapped read, window 9 inherits their reads): rank boundaries inside that run
    ("leak_small", "chr22:1000-3000", ["--active-region-off", "--ranks", "2", "--devices", "0,0", "--batch-windows", "1"], True),
    ("leak_small", "chr22:1000-3000", ["--active-region-off", "--ranks", "3", "--devices", "0,0,0", "--batch-windows", "1"], True)])
def test_lancet_gpu_ranks_gather_the_records_to_rank_0_and_write_the_reference_vcf(case, region, extra, files):
    """`lancet_gpu --ranks N`: N processes (one engine each) take the batches in turn, pack + key + reduce their records
    (lancet_records_pack), gather them to rank 0 (lancet_comm_gather: RCCL; two ranks cannot share a GPU under RCCL, so the N > 1 cases
    on this one-GPU box carry the payloads through the test transport) and rank 0 replays them in window order (lancet_records_merge):
    the reference's single-process VCF byte for byte, ##cmdline without the launcher's --rank / --rendezvous."""
    env = dict(os.environ)
    if files:
        env["LANCET_COMM_TEST_FILES"] = "1"
    else:
        env.pop("LANCET_COMM_TEST_FILES", None)
    r = subprocess.run([build.BIN, "--tumor", os.path.join(G, f"{case}.tumor.bam"), "--normal", os.path.join(G, f"{case}.normal.bam"),
                        "--ref", os.path.join(G, f"{case}.fa"), "--reg", region, "--date-line", "Sun Sep 27 05:27:00 2026"] + extra,
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert _body(r.stdout) == gu.golden_vcf(case)
    assert "--rendezvous" not in r.stdout and "--rank " not in r.stdout and "--ranks" in r.stdout
    n = int(extra[extra.index("--ranks") + 1])
    assert sum(1 for l in r.stderr.splitlines() if l.startswith("[lancet_gpu] rank ")) == n
    assert ("(files)" in r.stderr) == files and ("(rccl)" in r.stderr) == (not files)
