#!/usr/bin/env python3
"""Generates tests/golden/* by running the REFERENCE ITSELF in this container.

What is run: the unmodified reference binary that the survey stage built from /root/reference with the
recipe in SURVEY.md §8(c); it lives at /tmp/oracle/src/lancet (sources there are byte-identical to
/root/reference/src -- this script checks that with `diff -rq` before trusting it).  SAM->BAM conversion and
BAM indexing use the htslib `test_view` and `bamtools index` programs from the same build.  None of this
travels to the GPU box; only the small data files written under tests/golden/ do:

  <case>.reads.npz   the simulated reads (SAM fields) + contig sequence (inputs)
  <case>.vcf         the reference's VCF (header without ##fileDate, + body)   (expected output)
  <case>.trace.txt   digest of the reference's `-v` stderr: per-window k attempts, rejection reasons,
                     anchors, per-stage node/edge counts, emitted transcripts    (expected stage outputs)
  <case>.json        the command line / parameters of the case

Usage:  python tools/make_golden.py [case ...]
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lancet_amd import synth  # noqa: E402

REF_BIN = "/tmp/oracle/src/lancet"
TEST_VIEW = "/tmp/oracle/htslib-1.15.1/test/test_view"
BAMTOOLS = "/tmp/oracle/bamtools-2.5.2/bin/bamtools"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (synth kwargs, region, extra reference flags)
    "cfg1_k25": (dict(ref_len=6000, cov_t=30, cov_n=30), "chr22:2200-2799",
                 ["--padding", "0", "--min-k", "25", "--max-k", "25"]),
    "tile30": (dict(ref_len=9000, cov_t=30, cov_n=30), "chr22:1500-6500", []),
    "tile60": (dict(ref_len=6000, cov_t=60, cov_n=60, tumor_seed=111, normal_seed=212), "chr22:1500-3500", []),
    "str100": (dict(ref_len=6000, cov_t=100, cov_n=40, str_fraction=0.30, lowcomplex_fraction=0.05,
                    ref_seed=44, tumor_seed=141, normal_seed=242), "chr22:1500-3000", []),
    "err_hi": (dict(ref_len=6000, cov_t=40, cov_n=30, error_rate=0.01, read_len=100, ref_seed=7,
                    tumor_seed=17, normal_seed=27, somatic_every=700, germline_every=500),
               "chr22:1000-4500", []),
    # short inserts: the two mates of most fragments overlap -> exercises hasOverlappingMate / unsorted binary_search
    "ovl": (dict(ref_len=6000, cov_t=40, cov_n=30, ref_seed=9, tumor_seed=19, normal_seed=29, insert_mean=210.0,
                 insert_sd=35.0, somatic_every=900, germline_every=600), "chr22:1200-3800", []),
    # N in the window reference: single N, short runs, a run longer than the small k values
    "nref": (dict(ref_len=6000, cov_t=30, cov_n=30, ref_seed=12, tumor_seed=112, normal_seed=212,
                  n_runs=((1510, 1), (1933, 3), (2405, 17), (2950, 2), (3391, 1))), "chr22:1200-3700", []),
    # --linked-reads: barcode sets per k-mer, haplotype counts; short inserts so mates overlap too
    "lr30": (dict(ref_len=6000, cov_t=30, cov_n=30, ref_seed=31, tumor_seed=131, normal_seed=231, linked=True,
                  insert_mean=260.0, insert_sd=40.0, somatic_every=800, germline_every=500), "chr22:1200-3600",
             ["--linked-reads"]),
    "lr_deep": (dict(ref_len=6000, cov_t=70, cov_n=45, ref_seed=33, tumor_seed=133, normal_seed=233, linked=True,
                     insert_mean=230.0, insert_sd=45.0, somatic_every=500, germline_every=350, error_rate=0.008,
                     str_fraction=0.15), "chr22:1000-4200", ["--linked-reads"]),
    # active-region prefilter ON (the reference's default; every other case runs with --active-region-off).
    # "--active-region-on" is a marker for this script, not a reference flag.  The BAM/FASTA inputs of this case
    # are committed too (tests/golden/ar_small.*): they are the fixture of the BAM reader and of the CLI test.
    "ar_small": (dict(ref_len=4000, cov_t=28, cov_n=24, ref_seed=51, tumor_seed=151, normal_seed=251,
                      somatic_every=900, germline_every=700), "chr22:900-3000", ["--active-region-on"]),
    # linked reads through the whole command line (BX:Z / HP:i decoded from BAM), active regions on; inputs committed
    "lr_small": (dict(ref_len=3600, cov_t=30, cov_n=26, ref_seed=61, tumor_seed=161, normal_seed=261, linked=True,
                      insert_mean=300.0, insert_sd=40.0, somatic_every=700, germline_every=600), "chr22:800-2700",
                 ["--linked-reads", "--active-region-on"]),
    # every graph / STR / quality knob of the command line away from its default (they all travel in lancet_params)
    "knobs": (dict(ref_len=7000, cov_t=45, cov_n=35, ref_seed=81, tumor_seed=181, normal_seed=281, error_rate=0.008, str_fraction=0.12,
                   somatic_every=600, germline_every=450, read_len=125, insert_mean=330.0, insert_sd=45.0), "chr22:1000-5600",
              ["--min-k", "13", "--max-k", "75", "--tip-len", "7", "--cov-thr", "8", "--cov-ratio", "0.03", "--low-cov", "2",
               "--max-indel-len", "120", "--max-mismatch", "1", "--max-unit-length", "3", "--min-report-unit", "2",
               "--min-report-len", "5", "--dist-from-str", "2", "--trim-lowqual", "12", "--min-base-qual", "20"]),
    # deep coverage: k-mers with far more occurrences than the engine's LDS staging area holds (per-position counts in rounds)
    "deep200": (dict(ref_len=3400, cov_t=220, cov_n=180, ref_seed=91, tumor_seed=191, normal_seed=291, error_rate=0.006,
                     somatic_every=500, germline_every=400), "chr22:1200-2100", []),
    # found by tools/fuzz_reference.py (seed 32): k climbs to 99 on 100-base reads, some of which are trimmed to exactly k bases
    # (a read needs more than k bases to contribute a k-mer); linked reads, duplications, STR-rich
    "len_eq_k": (dict(ref_len=4918, cov_t=45.0, cov_n=28.0, ref_seed=1032, tumor_seed=2032, normal_seed=3032, error_rate=0.003, read_len=100,
                      insert_mean=260.0, insert_sd=40.0, somatic_every=1200, germline_every=500, str_fraction=0.3, lowcomplex_fraction=0.05,
                      dup_prob=1.0, linked=True), "chr22:969-2650", ["--linked-reads"]),
    # found by tools/fuzz_reference.py (seed 102): --low-cov 0 keeps every sequencing-error k-mer, the path search enumerates tens
    # of thousands of partial paths per window (the worst-case tier's FIFO is sized from --dfs-limit); linked reads
    "bushy": (dict(ref_len=4284, cov_t=45.0, cov_n=15.0, ref_seed=1102, tumor_seed=2102, normal_seed=3102, error_rate=0.015, read_len=100,
                   insert_mean=260.0, insert_sd=20.0, somatic_every=1200, germline_every=500, dup_prob=1.0, linked=True), "chr22:1190-2791",
              ["--linked-reads", "--cov-thr", "3", "--low-cov", "0"]),
    # found by tools/fuzz_reference.py (seed 10083): --low-cov 0 on 250-base reads at 1.5 % errors, k climbing to 100 -- four windows
    # keep ~17 k of their 17.4 k nodes past the first low-coverage filter (the worst-case tier's survivor arrays were capped at 16 Ki)
    "allsurvive": (dict(ref_len=3863, cov_t=70.0, cov_n=15.0, ref_seed=11083, tumor_seed=12083, normal_seed=13083, error_rate=0.015, read_len=250,
                        insert_mean=190.0, insert_sd=20.0, somatic_every=300, germline_every=500, str_fraction=0.3, lowcomplex_fraction=0.0,
                        dup_prob=0.0, palindromes=((2113, 11), (1607, 12))), "chr22:750-2901", ["--cov-thr", "5", "--low-cov", "0", "--min-k", "20"]),
    # even k (--min-k 12: the loop visits 12, 14, ...): k-mers that are their own reverse complement (planted in the contig at
    # several lengths, so that the window's final k meets one), CanonicalMer_t::set ties -> R (reference src/Mer.hh:57-71)
    "evenk": (dict(ref_len=7000, cov_t=36, cov_n=30, ref_seed=121, tumor_seed=1121, normal_seed=2121, somatic_every=700, germline_every=500,
                   palindromes=((1700, 6), (1990, 7), (2300, 8), (2610, 6), (2900, 7), (3205, 9), (3500, 6), (3800, 8), (4100, 7), (4400, 6),
                                (2455, 12), (3350, 11))),
              "chr22:1300-5000", ["--min-k", "12"]),
    # duplications at a coverage whose windows fit the LDS build kernel (<= 512 reads): k climbs through several graphs per window,
    # most of them built ahead (build_lds.h, build_kernel_body) after Ref_t::seq was trimmed by the rejected k
    "dups20": (dict(ref_len=8000, cov_t=22, cov_n=18, ref_seed=205, tumor_seed=215, normal_seed=225, dup_prob=1.0,
                    somatic_every=600, germline_every=500, read_len=100), "chr22:1000-6500", []),
    "dups": (dict(ref_len=8000, cov_t=40, cov_n=40, ref_seed=5, tumor_seed=15, normal_seed=25, dup_prob=1.0,
                  somatic_every=600, germline_every=500, read_len=100), "chr22:1000-6500", []),
    # --window-size above the default (the reference takes any -w: src/Lancet.cc:662,732): 1000-base windows every 100 bases -- the
    # engine lays its work space out for the batch's longest window (round 4: up to LC_MAXW = 1024)
    "w1000": (dict(ref_len=9000, cov_t=30, cov_n=30, ref_seed=61, tumor_seed=161, normal_seed=261, somatic_every=700, germline_every=500),
              "chr22:1500-6500", ["--window-size", "1000"]),
}


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def check_reference_is_unmodified():
    r = subprocess.run(["diff", "-rq", "/tmp/oracle/src", "/root/reference/src"], capture_output=True, text=True)
    extra = [l for l in r.stdout.splitlines() if not re.search(r"Only in /tmp/oracle/src: (lancet|lancet_pg)$", l)]
    if extra:
        raise SystemExit("reference build tree differs from /root/reference/src:\n" + "\n".join(extra))


def digest_trace(stderr: str) -> str:
    """Keeps the lines that pin stage results; drops alignments dumps and progress chatter."""
    keep = []
    pat = re.compile(
        r"^(== Processing|Repeat in reference|Near-perfect repeat|reads: |  \d+: nodes:| nodes: |ref trim5|"
        r"Ambiguous match|No match to reference|Cycle found|compressing graph|  removing |removing low coverage|"
        r"remove tips round| removed|remove short links| Found |FINISHED|>p_| refcomp:| perfect:|"
        r"searching from|Missing source|WARNING: DFS_LIMIT)")
    for line in stderr.splitlines():
        if pat.match(line):
            keep.append(line.rstrip())
    return "\n".join(keep) + "\n"


def make_case(name: str):
    kwargs, region, flags = CASES[name]
    data = synth.make_tumor_normal(**kwargs)
    ref, rname = data["ref"], data["rname"]
    with tempfile.TemporaryDirectory(prefix="lancet_golden_") as td:
        fa = os.path.join(td, "ref.fa")
        synth.write_fasta(fa, rname, ref)
        bams = {}
        for sample, rg, pairs in (("TUMOR", "tumor", data["tumor"]), ("NORMAL", "normal", data["normal"])):
            sam = os.path.join(td, f"{rg}.sam")
            bam = os.path.join(td, f"{rg}.bam")
            synth.write_sam(sam, rname, len(ref), sample, rg, pairs)
            run([TEST_VIEW, "-b", "-p", bam, sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            run([BAMTOOLS, "index", "-in", bam])
            bams[rg] = bam
        ar_on = "--active-region-on" in flags
        cmd = [REF_BIN, "--tumor", bams["tumor"], "--normal", bams["normal"], "--ref", fa, "--reg", region,
               "--num-threads", "1"] + ([] if ar_on else ["--active-region-off"]) + ["-v"] + [f for f in flags if f != "--active-region-on"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit(f"reference failed on case {name}")
        if ar_on:      # keep the real inputs of this case as fixtures (data files, written by htslib's test_view)
            import shutil
            os.makedirs(GOLDEN, exist_ok=True)
            for rg in ("tumor", "normal"):
                shutil.copy(bams[rg], os.path.join(GOLDEN, f"{name}.{rg}.bam"))
            shutil.copy(fa, os.path.join(GOLDEN, f"{name}.fa"))
        vcf = "".join(l + "\n" for l in r.stdout.splitlines()
                      if not l.startswith("##fileDate") and not l.startswith("##cmdline")
                      and not l.startswith("##reference"))
        # `ctime()` ends with '\n' and the header has no newline after it: "##fileDate=...\n##source"
    os.makedirs(GOLDEN, exist_ok=True)
    with open(os.path.join(GOLDEN, f"{name}.vcf"), "w") as f:
        f.write(vcf)
    with open(os.path.join(GOLDEN, f"{name}.trace.txt"), "w") as f:
        f.write(digest_trace(r.stderr))
    reads = {}
    for rg in ("tumor", "normal"):
        rs = synth.pairs_to_sorted_reads(data[rg])
        reads[f"{rg}_qname"] = np.array([x.qname for x in rs])
        reads[f"{rg}_flag"] = np.array([x.flag for x in rs], dtype=np.int32)
        reads[f"{rg}_pos"] = np.array([x.pos for x in rs], dtype=np.int32)
        reads[f"{rg}_mapq"] = np.array([x.mapq for x in rs], dtype=np.int32)
        reads[f"{rg}_cigar"] = np.array([x.cigar for x in rs])
        reads[f"{rg}_seq"] = np.array([x.seq for x in rs])
        reads[f"{rg}_qual"] = np.array([x.qual for x in rs])
        reads[f"{rg}_as"] = np.array([x.tags["AS"] for x in rs], dtype=np.int32)
        reads[f"{rg}_xs"] = np.array([x.tags["XS"] for x in rs], dtype=np.int32)
        reads[f"{rg}_md"] = np.array([x.tags["MD"] for x in rs])
        if kwargs.get("linked"):
            reads[f"{rg}_bx"] = np.array([x.tags.get("BX", "") for x in rs])
            reads[f"{rg}_hp"] = np.array([x.tags.get("HP", -1) for x in rs], dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.reads.npz"), ref=np.array(ref), rname=np.array(rname), **reads)
    with open(os.path.join(GOLDEN, f"{name}.json"), "w") as f:
        json.dump({"synth": kwargs, "region": region, "flags": flags,
                   "reference_cmd": " ".join(os.path.basename(c) if c.startswith("/tmp") else c for c in cmd),
                   "n_vcf_records": sum(1 for l in vcf.splitlines() if not l.startswith("#"))}, f, indent=1)
    print(f"{name}: {sum(1 for l in vcf.splitlines() if not l.startswith('#'))} VCF records, "
          f"{len(reads['tumor_qname'])}+{len(reads['normal_qname'])} reads")


if __name__ == "__main__":
    check_reference_is_unmodified()
    for c in (sys.argv[1:] or list(CASES)):
        make_case(c)
